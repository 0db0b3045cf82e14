"""Output files of the reference, byte for byte (Pore:571-630, Temp:867-933, Cube:340-418).

The reference calls ``ax.hist(data, range=(0, 1e-6), bins=200, density=True)`` (== ``np.histogram`` with the same
arguments) four times and dumps ``str(bins[0:len(n)])`` / ``str(n)`` under ``np.set_printoptions(threshold=maxsize)``
into eight text files; Temperature_Pore_MC additionally writes ``momentum_energy.csv`` through pandas.
Plotting itself is out of scope."""
from __future__ import annotations

import os
import sys

import numpy as np

HIST_FILES = [("total", "hist_x_axis_total_data.txt", "hist_y_axis_total_data.txt"),
              ("x", "hist_x_axis_x_data.txt", "hist_y_axis_x_data.txt"),
              ("y", "hist_x_axis_y_data.txt", "hist_y_axis_y_data.txt"),
              ("z", "hist_x_axis_z_data.txt", "hist_y_axis_z_data.txt")]


def density_from_counts(counts, lo=0.0, hi=10**-6):
    """np.histogram(..., density=True) from integer bin counts: n / diff(bin_edges) / n.sum()."""
    counts = np.asarray(counts)
    nb = counts.shape[-1]
    edges = np.linspace(lo, hi, nb + 1)
    db = np.array(np.diff(edges), float)
    n = counts.astype(np.int64)
    return n / db / n.sum(), edges


def density_from_paths(paths, nbins=200, lo=0.0, hi=10**-6):
    n, edges = np.histogram(np.asarray(paths, dtype=np.float64), bins=nbins, range=(lo, hi), density=True)
    return n, edges


def _str_full(a):
    with np.printoptions(threshold=sys.maxsize):
        return str(a)


def write_histograms(directory, densities, edges):
    """densities: dict total/x/y/z -> float64[nbins]; edges: float64[nbins+1].  Same text as Pore:607-630."""
    os.makedirs(directory, exist_ok=True)
    for key, fx, fy in HIST_FILES:
        n = densities[key]
        with open(os.path.join(directory, fx), "w") as f:
            f.write(_str_full(edges[0:len(n)]))
        with open(os.path.join(directory, fy), "w") as f:
            f.write(_str_full(n))


def write_momentum_energy_csv(path, momentum, energy_cold, energy_hot):
    """Temp:929-933: pandas DataFrame.from_dict({...}).to_csv('momentum_energy.csv') — index column, header
    ',Momentum,EnergyCold,EnergyHot', values through str() (mpmath mpf prints 15 significant digits; a step with no
    energised-wall hit is the int 0)."""
    with open(path, "w") as f:
        f.write(",Momentum,EnergyCold,EnergyHot\n")
        for k, (m, c, h) in enumerate(zip(momentum, energy_cold, energy_hot)):
            f.write(f"{k},{m},{c},{h}\n")
