"""Test infrastructure (GPU box, run by hand through gpurun; not collected by pytest): the statistical tier of the parity
protocol against the ONE physics artefact the reference holds.

    python tests/mfp_stat.py [steps] > gpurun_out/mfp_stat.json        (copy to profiles/r03_mfp_stat.json)

graph_sim_data.py:14-89 of the reference is a pasted 200-bin histogram of 423,143 cube free paths with the exponential fit
the author made (decay length 71.2 nm against the kinetic-theory lambda_mfp = 79.7 nm, Open_Air_Cube_MC.py:53).  Here the
HIP path runs the cube at the reference's OWN parameters (N = 24,627, sigma = 3.6e-19, dt = tau / 25, the reference's own
initial state from tests/golden/step_cube_natural.npz) for many steps, the 200-bin device histogram of the total free path is
read back, and three numbers are reported next to the reference's: the decay length from the reference's own fit
procedure (scipy curve_fit of a exp(b x) on the bin left edges), the decay length from a log-linear least-squares fit on
the non-empty bins, and the chi-square distance between the two normalised histograms.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from argon_monte_carlo_amd import params as PR
from argon_monte_carlo_amd.engine import Engine

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STATE_KEYS = ["x_vals", "y_vals", "z_vals", "x_velocities", "y_velocities", "z_velocities", "dist_since_collision",
              "dist_x_since_collision", "dist_y_since_collision", "dist_z_since_collision", "full_path_traveled"]


def decay_fits(left_edges, density):
    from scipy.optimize import curve_fit
    popt, _ = curve_fit(lambda t, p, q: p * np.exp(q * np.array(t)), left_edges, density, p0=[14000000.0, -11000000.0], maxfev=25000)
    m = density > 0
    centres = left_edges + 0.5 * (left_edges[1] - left_edges[0])
    slope, icpt = np.polyfit(centres[m], np.log(density[m]), 1)
    return -1.0 / popt[1], -1.0 / slope, int(m.sum())


def one_run(steps, steps_per_mft, G, H):
    p, c = PR.cube_params(steps_per_mft=steps_per_mft)
    assert int(p.n) == int(G["num_molecules"]) == 24627
    p.max_paths = -1                    # histograms only
    eng = Engine(p)
    init = [G[f"s-001_{k}"] for k in STATE_KEYS]
    eng.upload(*init[:10], init[10])
    t0 = time.perf_counter()
    npp = 0
    done = 0
    while done < steps:
        k = min(10_000, steps - done)
        npp += eng.run(c["dt"], k)["n_pp"]
        done += k
    el = time.perf_counter() - t0
    counts, n_total = eng.histograms()
    tot = np.asarray(counts[0], dtype=np.float64)           # total free path, 200 bins on [0, 1e-6]
    nb = len(tot)
    width = 1e-6 / nb
    left = np.arange(nb) * width
    in_range = tot.sum()
    dens = tot / (in_range * width)                         # np.histogram(..., density=True)
    lam_fit, lam_log, nbins_used = decay_fits(left, dens)
    ref_d = np.asarray(H["density"], dtype=np.float64)
    ref_fit, ref_log, ref_bins = decay_fits(np.asarray(H["x"]), ref_d)
    pq, qq = dens * width, ref_d * width                    # probabilities per bin
    m = (pq + qq) > 0
    chi2 = 0.5 * float(np.sum((pq[m] - qq[m]) ** 2 / (pq[m] + qq[m])))
    # the same distance for two samples of the reference's size drawn from ONE exponential: what sampling noise alone gives
    rng = np.random.default_rng(1)
    noise = []
    for _ in range(20):
        a = np.histogram(rng.exponential(lam_fit, int(H["n_paths"])), bins=nb, range=(0, 1e-6))[0].astype(float)
        b = np.histogram(rng.exponential(lam_fit, int(in_range)), bins=nb, range=(0, 1e-6))[0].astype(float)
        a /= a.sum(); b /= b.sum()
        mm = (a + b) > 0
        noise.append(0.5 * float(np.sum((a[mm] - b[mm]) ** 2 / (a[mm] + b[mm]))))
    out = {
        "what": "cube at the reference's own parameters on the HIP path vs the histogram pasted in the reference's graph_sim_data.py:14-89",
        "n_particles": int(p.n), "dt": c["dt"], "dt_is_tau_over": steps_per_mft, "mean_displacement_per_step_in_collision_ranges": c["v_mean"] * c["dt"] / p.collision_range,
        "steps": steps, "gpu_seconds": el, "pp_collisions": int(npp),
        "completed_paths_total": int(n_total), "completed_paths_in_histogram_range": int(in_range),
        "lambda_mfp_kinetic_theory_nm": c["lambda_mfp"] * 1e9,
        "hip": {"decay_length_curve_fit_nm": lam_fit * 1e9, "decay_length_loglinear_nm": lam_log * 1e9, "non_empty_bins": nbins_used},
        "reference_histogram": {"paths": int(H["n_paths"]), "decay_length_curve_fit_nm": ref_fit * 1e9,
                                "decay_length_loglinear_nm": ref_log * 1e9, "non_empty_bins": ref_bins,
                                "authors_fit_b": float(H["fit_b"])},
        "chi2_distance_normalised_histograms": chi2,
        "chi2_distance_expected_from_sampling_noise": {"mean": float(np.mean(noise)), "max": float(np.max(noise))},
        "ratio_decay_length_hip_over_reference": lam_fit / ref_fit,
    }
    eng.close()
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    G = np.load(os.path.join(GOLD, "step_cube_natural.npz"))
    H = np.load(os.path.join(GOLD, "graph_hist.npz"))
    # the committed time step (tau / 25: a particle moves ~9 collision ranges per step and overlap-at-end-of-step detection
    # misses most encounters) and finer ones (tau / 1000 is what the reference's pore scripts use)
    runs = [one_run(steps, 25, G, H)] + [one_run(steps, k, G, H) for k in (200, 1000)]
    print(json.dumps({"runs": runs, "note": "the pasted histogram's provenance is unknown (SURVEY 6): its 71 nm decay length is not what the "
                      "committed cube constants (dt = tau / 25) produce with the reference's own end-of-step overlap detection — the HIP path "
                      "equals the reference bit for bit at those constants (tests/test_oracle_natural.py) — but what a finer time step gives"}, indent=1))


if __name__ == "__main__":
    main()
