"""hydrochrono_amd -- MI355X-native HydroChrono hydro-force path (per-timestep Cummins-equation forces).

The product is the C ABI in include/hydrochrono_amd.h (libhydrochrono_amd.so: HIP kernels for gfx950 + host
bookkeeping).  This package is the thin Python face of it used by the tests and bench.py:
  capi       ctypes signatures of every exported symbol
  hydro      HydroForces: Python mirror of the reference's TestHydro surface over the C ABI
  synthetic  seeded many-body input generators (benchmark configurations C2/C3/C4 of SURVEY.md 8d)
  mock_chrono  stand-ins for the Chrono time loop (prescribed motion, 1-DOF heave integrator)
Importing the package does not load the library; HydroForces() does, and raises if it is missing.
"""
import os as _os

# HC_QUEUE_DEV_MEM=1 (opt-in): the AQL packet rings of the process's HSA queues in device memory instead of host
# memory -- 1.0-1.8 us less per synchronous hc_step.  The HSA runtime reads its variable once, at the first HIP call of the process; the
# library sets it when it is loaded (hc_runtime.cpp), which in an interpreter can be too late (torch.cuda.is_available() initialises
# HIP), so the package sets it at import.  Not the default: it moves the HIP runtime's own queues too, whose dispatch path does not
# order its packet stores against its doorbells through the HDP (two of ~2 600 differential cases died with it, EXPERIMENTS.md round 6).
if _os.environ.get("HC_QUEUE_DEV_MEM", "0") not in ("", "0"):
    _os.environ.setdefault("HSA_ALLOCATE_QUEUE_DEV_MEM", "1")

from .hydro import HydroError, HydroForces  # noqa: F401,E402

__all__ = ["HydroForces", "HydroError"]
