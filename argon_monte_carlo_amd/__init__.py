"""argon_monte_carlo_amd — MI355X-native drift / wall-reflection / particle-particle collision sweep of the
Lightbrite88/Argon_Monte_Carlo simulation (hand-written HIP for gfx950 behind a C ABI, include/argonmc.h).

Importing the package does not need a GPU; constructing an ``Engine``/``Simulation`` does — there is no CPU path."""
from . import params  # noqa: F401
from ._abi import (AMC_GEOM_CELL, AMC_GEOM_CUBE, AMC_GEOM_PORE, AMC_GEOM_PORE_ENERGISED, AmcParams)  # noqa: F401

__all__ = ["params", "AmcParams", "AMC_GEOM_CELL", "AMC_GEOM_CUBE", "AMC_GEOM_PORE", "AMC_GEOM_PORE_ENERGISED"]
