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

# The AQL packet ring of every HSA queue in device memory instead of host memory: 1.4-1.9 us less per synchronous hc_step (the packet
# processor fetches its packets locally, profiles/r06/queue_dev_mem_ab.txt).  The HSA runtime reads the variable once, at its
# initialisation, i.e. at the first HIP call of the process -- the library asks for it when it is loaded (hc_runtime.cpp), which in an
# interpreter can be too late (torch.cuda.is_available() initialises HIP), so the package asks at import.  Not in the ranks of a
# multi-process launch (WORLD_SIZE > 1): there it would also move RCCL's queues, for 2 % of a wide shard's step.  HC_QUEUE_DEV_MEM=0: never.
if _os.environ.get("HC_QUEUE_DEV_MEM", "1") != "0" and int(_os.environ.get("WORLD_SIZE", "1") or 1) == 1:
    _os.environ.setdefault("HSA_ALLOCATE_QUEUE_DEV_MEM", "1")
elif _os.environ.get("HC_QUEUE_DEV_MEM") is None:
    _os.environ["HC_QUEUE_DEV_MEM"] = "0"  # (the library's own load-time request stays off as well)

from .hydro import HydroError, HydroForces  # noqa: F401,E402

__all__ = ["HydroForces", "HydroError"]
