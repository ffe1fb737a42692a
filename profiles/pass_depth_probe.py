"""The pass kernel alone at depth 16 / 32 / 64 (tuning build: hc_tuning_time_pass) over the K of C3 (64 bodies) and of one C4/8 rank
(rows of 64 of 512 bodies, 9.69 GB): mean launch time by HIP events, bytes the launch moves once, the rates against the HBM and FP64 MFMA
peaks.  HC_BLOCK64_MT / HC_BLOCK64_R select the depth-64 variant (default here: 6 / 3, the fastest of profiles/r05/depth64_sweep.txt).
HC_TUNING_PASS_PAUSE_US=<us> leaves the GPU idle for that long in front of every launch (default 0: the launches follow each other,
i.e. sustained matrix-pipe + HBM load).  Run under rocprofv3 --kernel-trace --stats for the kernel rows."""
import ctypes as C
import os
import sys

import numpy as np

os.environ.setdefault("HYDROCHRONO_AMD_FLAVOR", "tuning")
os.environ.setdefault("HC_BLOCK64_MT", "6")
os.environ.setdefault("HC_BLOCK64_R", "3")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench as B  # noqa: E402
from hydrochrono_amd import capi  # noqa: E402
from hydrochrono_amd.hydro import HydroForces  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402

lib = capi.load()
probe = lib.hc_tuning_time_pass
probe.restype = C.c_int
probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
for name, N, rng in (("C3 (64 bodies)", 64, (0, 64)), ("one C4/8 rank (rows of 64 of 512 bodies)", 512, (0, 64))):
    gpu = HydroForces(N, device=0, body_range=rng)
    gpu.synth_fill(20251031, B.S_RIRF, B.DT, 0, B.DT)
    gpu.finalize()
    gpu.add_waves_none()
    motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
    nhist = B.S_RIRF + 5
    t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
    gpu.set_history(t_hist, np.stack([motion.velocity6(t) for t in t_hist]))
    for depth in (16, 32, 64):
        us, once = C.c_double(), C.c_double()
        rc = probe(gpu.ctx, depth, 12, C.byref(us), C.byref(once))
        if rc:
            print(name, depth, "error:", lib.hc_last_error(gpu.ctx).decode())
            continue
        flops = 2.0 * depth * (once.value / 8.0)  # every K word meets `depth` step columns (upper bound: s_cut trims the head)
        print(f"{name:42s} depth {depth:2d}: {us.value:8.1f} us per launch = {us.value / depth:6.2f} us per step; {once.value / 1e9:6.3f} GB once -> "
              f"{once.value / us.value / 1e6 / 8.0:5.3f} of 8 TB/s; FP64 {flops / us.value / 1e6 / 78.6:5.3f} of 78.6 TFLOP/s", flush=True)
    gpu.close()
