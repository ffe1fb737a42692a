"""Companion of shard_curve.py: the rank-0 shard of a G-way sharding of C4 (G = 1, 2, 4, 8) back to back under each pinned pass schedule
and under the default -- does "one block ahead" (what the adaptive default answers for wide systems) stay the better schedule as the
K slice grows from 9.7 to 77 GB?   python profiles/shard_curve_sched.py [steps = 192]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench as B  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 192
N, warm = B.N_BODIES_C4, 72
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
nhist = B.S_RIRF + 5
t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
n_all = 3 * (warm + steps) + 8
times = [B.T0 + k * B.DT for k in range(n_all)]
states = [motion.state(t) for t in times]
for G in (8, 4, 2, 1):
    gpu = B.make_shard(N, 0, N // G, 0, B.DT, B.T0 + (n_all + 8) * B.DT + 5.0, 32, t_hist, v_hist)
    k = 0
    row = {"G": G, "K_slice_GB": None}
    for name, sched in (("default", -1), ("at_block_start", 0), ("one_block_ahead", 1)):
        gpu.set_pass_schedule(sched)
        gpu.enable_profiling(1000000)
        gpu.reset_profile()
        per = np.zeros(steps)
        for i in range(warm + steps):
            a = time.perf_counter()
            gpu.step(times[k], *states[k])
            if i >= warm:
                per[i - warm] = time.perf_counter() - a
            k += 1
        p = gpu.profile()
        row["K_slice_GB"] = p["conv_kernel_bytes"] / 1e9
        launches = max(1, p["block_kernel_launches"])
        row[name] = {"ms_per_step": round(float(per.mean()) * 1e3, 5), "median": round(float(np.median(per)) * 1e3, 5), "max": round(float(per.max()) * 1e3, 4),
                     "pass_launch_us": round(1e6 * p["block_kernel_seconds"] / launches, 1), "pass_launches": int(p["block_kernel_launches"]),
                     "answers_ahead": int(p["schedule_blocks_ahead"]), "answers_at_start": int(p["schedule_blocks_at_start"])}
    print(json.dumps(row), flush=True)
    gpu.close()
