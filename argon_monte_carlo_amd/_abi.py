"""ctypes mirror of include/argonmc.h (structs only; no library is loaded here)."""
import ctypes as C

AMC_ABI_VERSION = 2

AMC_OK = 0
AMC_ERR_INVALID = -1
AMC_ERR_NO_DEVICE = -2
AMC_ERR_HIP = -3
AMC_ERR_CAPACITY = -4
AMC_ERR_FP = -5
AMC_ERR_STATE = -6

AMC_GEOM_CELL = 0
AMC_GEOM_CUBE = 1
AMC_GEOM_PORE = 2
AMC_GEOM_PORE_ENERGISED = 3

_D = C.c_double


class AmcParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("geometry", C.c_int32), ("n", C.c_int64),
        ("collision_range", _D), ("argon_mass", _D), ("argon_radius", _D),
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("reserved0", C.c_int32),
        ("dx", _D), ("dy", _D), ("dz", _D),
        ("overlap_x", _D), ("overlap_y", _D), ("overlap_z", _D),
        ("cube_x", _D), ("cube_y", _D), ("cube_z", _D),
        ("R_oa", _D), ("R_oa_c", _D), ("R_p", _D), ("R_p_c", _D), ("R_g", _D), ("R_g_c", _D),
        ("H", _D), ("h_oa", _D), ("z_cold", _D), ("z_gap_bottom", _D), ("z_gap_top", _D),
        ("oob_z_lo_fix", _D), ("oob_z_hi_fix", _D),
        ("R_oa_sq", _D), ("R_g_sq", _D), ("R_p_sq", _D),
        ("z_oob_hot_top", _D), ("z_oob_gap_top", _D),
        ("t_z3_cold", _D), ("t_z3_hot", _D), ("t_zgap_lo", _D), ("t_zgap_hi", _D),
        ("R_g_c_sq", _D), ("R_p_c_sq", _D),
        ("E_cold", _D), ("E_hot", _D), ("alpha_coated", _D), ("alpha_gap", _D), ("cos85", _D),
        ("hist_bins", C.c_int32), ("reserved1", C.c_int32), ("hist_lo", _D), ("hist_hi", _D),
        ("fine_cell", _D), ("device", C.c_int32), ("detect_mode", C.c_int32),
        ("max_candidates", C.c_int64), ("max_paths", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    @classmethod
    def from_dict(cls, d):
        p = cls()
        for k, _ in cls._fields_:
            if k in d:
                v = d[k]
                setattr(p, k, v.item() if hasattr(v, "item") else v)
        p.struct_size = C.sizeof(cls)
        return p


class AmcStepStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in (
        "n_pp", "n_wall", "n_oob_walls", "n_oob_pp", "n_paths", "n_candidates", "n_clusters", "n_rounds",
        "n_fp_errors", "flags")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class AmcPathRecord(C.Structure):
    _fields_ = [("step", C.c_int32), ("phase", C.c_int32), ("cell", C.c_int64), ("i", C.c_int32), ("j", C.c_int32),
                ("which", C.c_int32), ("reserved", C.c_int32), ("total", _D), ("px", _D), ("py", _D), ("pz", _D)]


class AmcIcConfig(C.Structure):
    """amc_ic_config (include/argonmc.h): synthetic initial conditions generated on the device."""
    _fields_ = [("struct_size", C.c_int32), ("n_regions", C.c_int32), ("seed", C.c_uint64), ("a_shape", C.c_double),
                ("first", C.c_int64 * 9), ("radius", C.c_double * 8), ("z_lo", C.c_double * 8), ("z_hi", C.c_double * 8)]


class AmcTempRng(C.Structure):
    """amc_temp_rng (include/argonmc.h): configuration of the opt-in device-side energised-wall sampling."""
    _fields_ = [("struct_size", C.c_int32), ("n_gl", C.c_int32), ("seed", C.c_uint64),
                ("t_cold", C.c_double), ("t_hot", C.c_double), ("gap_height", C.c_double), ("gap_bottom_height", C.c_double),
                ("t_debye_alumina", C.c_double), ("n_alumina", C.c_double), ("boltzman", C.c_double),
                ("gl_x", C.c_double * 32), ("gl_w", C.c_double * 32)]


# numpy dtype with the same layout as amc_path_record
def path_record_dtype():
    import numpy as np
    return np.dtype([("step", "<i4"), ("phase", "<i4"), ("cell", "<i8"), ("i", "<i4"), ("j", "<i4"),
                     ("which", "<i4"), ("reserved", "<i4"), ("total", "<f8"), ("px", "<f8"), ("py", "<f8"),
                     ("pz", "<f8")])


AMC_K_NAMES = ["drift_walls", "bin_count", "bin_scan", "bin_scatter", "detect", "resolve", "bounds", "validate",
               "resolve_more", "commit", "clusters_wide", "fixup"]
