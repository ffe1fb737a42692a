"""Per-block GPU time of the look-ahead machinery at C3 under both pass schedules (the library's own kernel timings).
python profiles/ahead_breakdown.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bench as B
from hydrochrono_amd.hydro import HydroForces
from hydrochrono_amd.mock_chrono import PrescribedMotion
from hydrochrono_amd.synthetic import many_body_case, rest_positions

case = many_body_case(64, S=B.S_RIRF, dt_rirf=B.DT, n_exc=B.N_EXC, dt_exc=B.DT, seed=20251031)
motion = PrescribedMotion(64, rest_positions(case), seed=20251031)
nhist = B.S_RIRF + 5
t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
for sched, slices in ((0, 0), (1, 8), (1, 4), (1, 2), (1, 1)):
    gpu = HydroForces.from_case(case)
    gpu.add_waves_irregular(num_bodies=64, **dict(B.WAVES, simulation_dt=B.DT, simulation_duration=B.T0 + 10.0))
    gpu.set_pass_schedule(sched, slices)
    gpu.set_history(t_hist, v_hist)
    n0, n = 97, 320
    for k in range(n0):
        gpu.step(B.T0 + k * B.DT, *motion.state(B.T0 + k * B.DT))
    gpu.enable_profiling(1)
    gpu.reset_profile()
    import time
    t0 = time.perf_counter()
    for k in range(n0, n0 + n):
        gpu.step(B.T0 + k * B.DT, *motion.state(B.T0 + k * B.DT))
    wall = time.perf_counter() - t0
    p = gpu.profile()
    nb = n / 32
    print(f"schedule {sched} slices {slices}: per block of 32 steps: pass {p['block_kernel_seconds'] / nb * 1e6:7.1f} us in {p['block_kernel_launches'] / nb:4.1f} launches, "
          f"short passes {p['mini_pass_seconds'] / nb * 1e6:6.1f} us in {p['mini_pass_launches'] / nb:3.1f}, scatter {p['scatter_kernel_seconds'] / nb * 1e6:6.1f}, "
          f"step kernels {p['step_kernel_seconds'] / nb * 1e6:6.1f};  wall per step (profiling on) {wall / n * 1e6:6.1f} us", flush=True)
    gpu.close()
