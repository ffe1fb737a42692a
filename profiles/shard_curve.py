"""The 1 -> 8 GPU curve of C4 as far as ONE GPU can measure it: for G = 1, 2, 4, 8 the shard of rank 0 of a G-way row sharding (rows of
512 / G bodies of the coupled 512-body array, K slice 77.4 / G GB) runs alone on this GPU -- synchronous hc_step, back to back, library
defaults -- and, for comparison, the shard of the LAST rank (same size; the schedules must answer alike).  What a G-GPU node would add on
top is the exchange: every rank's GPU stores its rows into its own shared-memory result buffer and every rank polls all G buffers
(hydrochrono_amd/host_exchange.py) -- a wait for the slowest rank, not a collective.  So  max over ranks of these step times  is the
PREDICTED step time of the sharded array, and T(1) / T(G) the predicted strong-scaling speed-up.  A prediction, not a measurement.
    python profiles/shard_curve.py [steps = 256]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench as B  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, warm = B.N_BODIES_C4, 104
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
nhist = int(np.ceil(B.S_RIRF * B.DT / B.DT)) + 5
t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
n_all = warm + steps + 8
times = [B.T0 + k * B.DT for k in range(n_all)]
states = [motion.state(t) for t in times]
rows = []
for G in (8, 4, 2, 1):
    for which in ((0, G - 1) if G > 1 else (0,)):
        b0, b1 = which * (N // G), (which + 1) * (N // G)
        t_a = time.perf_counter()
        gpu = B.make_shard(N, b0, b1, 0, B.DT, B.T0 + (n_all + 8) * B.DT + 5.0, 32, t_hist, v_hist)
        t_setup = time.perf_counter() - t_a
        per = np.zeros(steps)
        for k in range(warm + steps):
            a = time.perf_counter()
            gpu.step(times[k], *states[k])
            if k >= warm:
                per[k - warm] = time.perf_counter() - a
        p = gpu.profile()
        row = {"G": G, "rank": which, "bodies": [b0, b1], "K_slice_GB": p["conv_kernel_bytes"] / 1e9, "ms_per_step": float(per.mean()) * 1e3,
               "median_ms_per_step": float(np.median(per)) * 1e3, "max_ms_per_step": float(per.max()) * 1e3,
               "blocks_ahead": int(p["schedule_blocks_ahead"]), "blocks_at_start": int(p["schedule_blocks_at_start"]), "setup_s": t_setup}
        rows.append(row)
        print(json.dumps(row), flush=True)
        gpu.close()
t1 = max(r["ms_per_step"] for r in rows if r["G"] == 1)
print("\nG   K slice   predicted ms per step (max over the ranks measured)   speed-up over G = 1   efficiency")
for G in (1, 2, 4, 8):
    t = max(r["ms_per_step"] for r in rows if r["G"] == G)
    gb = [r["K_slice_GB"] for r in rows if r["G"] == G][0]
    print(f"{G}   {gb:6.2f} GB   {t:8.4f}   {t1 / t:5.2f}x   {t1 / t / G:5.2f}")
