"""Loader of libargonmc.so (the HIP/gfx950 C-ABI library, include/argonmc.h).

There is deliberately no fallback: if the library is missing or no MI355X is usable, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from ._abi import (AMC_ABI_VERSION, AmcIcConfig, AmcParams, AmcPathRecord, AmcStepStats, AmcTempRng)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libargonmc.so")
_LIB = None

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_ctx = C.c_void_p

# every symbol include/argonmc.h declares: (restype, argtypes)
SIGNATURES = {
    "amc_abi_version": (C.c_int, []),
    "amc_create": (C.c_int, [C.POINTER(_ctx), C.POINTER(AmcParams)]),
    "amc_destroy": (None, [_ctx]),
    "amc_last_error": (C.c_char_p, [_ctx]),
    "amc_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "amc_use_null_stream": (C.c_int, [_ctx]),
    "amc_synchronize": (C.c_int, [_ctx]),
    "amc_upload": (C.c_int, [_ctx] + [_dp] * 10 + [_u8p]),
    "amc_download": (C.c_int, [_ctx] + [_dp] * 10 + [_u8p]),
    "amc_download_prior": (C.c_int, [_ctx, _dp, _dp, _dp]),
    "amc_timestep": (C.c_int, [_ctx, C.c_double, C.POINTER(AmcStepStats)]),
    "amc_run": (C.c_int, [_ctx, C.c_double, C.c_int64, C.POINTER(AmcStepStats)]),
    "amc_stage_drift": (C.c_int, [_ctx, C.c_double]),
    "amc_stage_walls": (C.c_int, [_ctx, C.POINTER(AmcStepStats)]),
    "amc_stage_bounds": (C.c_int, [_ctx, _i64p]),
    "amc_stage_sweep": (C.c_int, [_ctx, C.POINTER(AmcStepStats)]),
    "amc_pairwise_cell": (C.c_int, [_ctx, C.c_int64] + [_dp] * 4 + [_u8p] + [_dp] * 6 + [_dp, C.c_size_t,
                                                                                       C.POINTER(C.c_size_t), _i64p]),
    "amc_temp_begin": (C.c_int, [_ctx, C.c_double]),
    "amc_temp_end": (C.c_int, [_ctx, C.POINTER(AmcStepStats)]),
    "amc_temp_cases_device": (C.c_int, [_ctx, C.POINTER(AmcTempRng)]),
    "amc_temp_device_results": (C.c_int, [_ctx, C.c_int, _i32p, _dp, _dp, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)]),
    "amc_temp_device_sums": (C.c_int, [_ctx, _dp, C.POINTER(C.c_int32)]),
    "amc_temp_device_draws": (C.c_int, [_ctx, C.c_int, _i32p, _dp, _dp, _dp, _dp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "amc_host_directions": (C.c_int, [C.POINTER(C.c_uint32), _i32p, C.POINTER(C.c_uint32), _i32p, _dp, _u8p, C.c_int64, C.c_double,
                                      C.c_double, C.c_int, C.c_void_p, _dp]),
    "amc_wall_hits": (C.c_int, [_ctx, C.c_int, _i32p, _dp, _dp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "amc_wall_apply": (C.c_int, [_ctx, C.c_int, _dp, _dp, C.c_size_t, _dp, _dp]),
    "amc_wall_park": (C.c_int, [_ctx, C.c_int, _dp, C.c_size_t]),
    "amc_wall_finish": (C.c_int, [_ctx, C.c_int, _dp, C.c_size_t, _dp, _dp]),
    "amc_wall_hits_again": (C.c_int, [_ctx]),
    "amc_drain_paths": (C.c_int, [_ctx, C.POINTER(AmcPathRecord), C.c_size_t, C.POINTER(C.c_size_t)]),
    "amc_paths_pending": (C.c_int, [_ctx, C.POINTER(C.c_size_t)]),
    "amc_histograms": (C.c_int, [_ctx, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "amc_reset_outputs": (C.c_int, [_ctx]),
    "amc_init_synthetic": (C.c_int, [_ctx, C.POINTER(AmcIcConfig)]),
    "amc_set_shard": (C.c_int, [_ctx, C.c_int64, C.c_int64]),
    "amc_mg_local": (C.c_int, [_ctx, C.c_double]),
    "amc_mg_exchange_view": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "amc_mg_pack": (C.c_int, [_ctx, C.c_int]),
    "amc_mg_sweep": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "amc_mg_candidates_view": (C.c_int, [_ctx, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "amc_mg_detect": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "amc_mg_resolve": (C.c_int, [_ctx, C.c_int]),
    "amc_mg_bounds": (C.c_int, [_ctx]),
    "amc_mg_finish": (C.c_int, [_ctx, C.POINTER(AmcStepStats)]),
    "amc_profile": (C.c_int, [_ctx, C.c_int]),
    "amc_kernel_times": (C.c_int, [_ctx, _dp, _i64p]),
    "amc_kernel_name": (C.c_char_p, [C.c_int]),
    "amc_overlap_stats": (C.c_int, [_ctx, _i64p]),
}


class ArgonMCError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libargonmc error {code}: {msg}")
        self.code = code


def build(verbose=False):
    """Compile libargonmc.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    if verbose or r.returncode != 0:
        print(r.stdout.decode(errors="replace")[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libargonmc.so failed")
    with open(STAMP_PATH, "w") as fh:
        fh.write(source_digest() + "\n")
    return LIB_PATH


STAMP_PATH = LIB_PATH + ".sources"


def _source_files():
    src_dir = os.path.join(_HERE, "csrc")
    files = [os.path.join(src_dir, f) for f in sorted(os.listdir(src_dir)) if f.endswith((".hip", ".h")) or f == "Makefile"]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "argonmc.h"))
    return files


def source_digest():
    """sha256 over the library's sources (csrc/*.hip, csrc/*.h, the Makefile, include/argonmc.h)."""
    import hashlib
    h = hashlib.sha256()
    for f in _source_files():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale_sources():
    """Non-empty when the built libargonmc.so does not belong to the sources in the tree.  The .so is not under version
    control (it travels with the working tree, and file times do not survive every copy): build() leaves the digest of
    the sources it compiled next to it, and a library without a matching digest is refused — it would silently run old
    kernels."""
    if not os.path.exists(LIB_PATH):
        return []
    try:
        with open(STAMP_PATH) as fh:
            built = fh.read().strip()
    except OSError:
        return ["(no source digest next to the library)"]
    return [] if built == source_digest() else ["(sources changed since the library was built)"]


def load():
    """dlopen libargonmc.so and type every entry point.  Raises if the library is absent — there is no CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ArgonMCError(-2, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(no CPU fallback exists)")
    stale = stale_sources()
    if stale:
        raise ArgonMCError(-3, f"{LIB_PATH} does not match the sources {stale[0]}: rebuild it "
                               "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = missing export
        fn.restype = res
        fn.argtypes = args
    if lib.amc_abi_version() != AMC_ABI_VERSION:
        raise ArgonMCError(-1, "ABI version mismatch between _abi.py and libargonmc.so")
    _LIB = lib
    return lib
