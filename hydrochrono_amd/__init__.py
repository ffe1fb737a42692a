"""hydrochrono_amd -- MI355X-native HydroChrono hydro-force path (per-timestep Cummins-equation forces).

The product is the C ABI in include/hydrochrono_amd.h (libhydrochrono_amd.so: HIP kernels for gfx950 + host
bookkeeping).  This package is the thin Python face of it used by the tests and bench.py:
  capi       ctypes signatures of every exported symbol
  hydro      HydroForces: Python mirror of the reference's TestHydro surface over the C ABI
  synthetic  seeded many-body input generators (benchmark configurations C2/C3/C4 of SURVEY.md 8d)
  mock_chrono  stand-ins for the Chrono time loop (prescribed motion, 1-DOF heave integrator)
Importing the package does not load the library; HydroForces() does, and raises if it is missing.
"""
from .hydro import HydroError, HydroForces  # noqa: F401

__all__ = ["HydroForces", "HydroError"]
