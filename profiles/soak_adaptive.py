"""Soak of the ADAPTIVE pass schedule (the default) at C3 size: the caller changes its regime at random -- stretches issued back to back
through hc_step_many (the C ABI's own loop, gaps of a fraction of a microsecond), stretches with 20-200 us of host work between the calls,
stretches on another step size now and then -- so the schedule flips between "at block start" and "one block ahead" at arbitrary
places in a block, with plans dropped in between.  Every step against the flat-array CPU oracle.  (Steps back in time under the schedules:
tests/test_gpu_ahead.py -- the oracle has to be rebuilt from the kept history there.)  python profiles/soak_adaptive.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import bench as B  # noqa: E402
import oracle as orc_mod  # noqa: E402
from cases import load_into_oracle  # noqa: E402
from hydrochrono_amd.hydro import HydroForces  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402
from hydrochrono_amd.synthetic import many_body_case, rest_positions  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
case = many_body_case(64, S=B.S_RIRF, dt_rirf=B.DT, n_exc=B.N_EXC, dt_exc=B.DT, seed=20251031)
gpu = HydroForces.from_case(case)
motion = PrescribedMotion(64, rest_positions(case), seed=20251031)
duration = B.T0 + nsteps * 0.0103 + 30.0
kw = dict(B.WAVES, simulation_dt=B.DT, simulation_duration=duration)
gpu.add_waves_irregular(num_bodies=64, **kw)
orc_mod.set_num_threads(min(64, os.cpu_count() or 1))
orc = load_into_oracle(case)
orc.add_waves_irregular(**kw)
nhist = B.S_RIRF + 5
t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
gpu.set_history(t_hist, v_hist)
orc.prefill_history(t_hist, v_hist)
orc.flat_prepare()
rng = np.random.default_rng(5)
t, worst, done, t0, flips, last = B.T0, 0.0, 0, time.time(), 0, None
while done < nsteps:
    tight = rng.random() < 0.5
    n = int(rng.integers(5, 140))
    dt = B.DT if rng.random() > 0.1 else B.DT * rng.uniform(0.6, 1.3)  # a stretch on another step size now and then
    times = t + dt * np.arange(n)
    states = np.stack([motion.packed(x) for x in times])
    if tight:
        forces, _ = gpu.step_many(times, states)
    else:
        gap = rng.uniform(20e-6, 200e-6)
        forces = np.empty((len(times), gpu.D_local))
        for k, x in enumerate(times):
            forces[k] = gpu.step(x, *motion.state(x))
            b = time.perf_counter()
            while time.perf_counter() - b < gap:
                pass
    for k, x in enumerate(times):
        fo = orc.flat_step(x, *motion.state(x))
        e = float(np.max(np.abs(forces[k] - fo)) / np.max(np.abs(fo)))
        worst = max(worst, e)
        if e > 1e-9:
            print(f"step {done + k} t {x}: relative error {e:.3e} ({'tight' if tight else 'gaps'} stretch)")
            sys.exit(1)
    done += len(times)
    t = float(times[-1]) + dt
    p = gpu.profile()
    now = (p["schedule_blocks_ahead"], p["schedule_blocks_at_start"])
    if last is not None and (now[0] > last[0]) and (now[1] > last[1]):
        flips += 1
    last = now
    if done % 2000 < len(times):
        print(f"{done} steps, worst {worst:.2e}; schedule answers ahead / at start {now[0]} / {now[1]}, blocks adopted {p['ahead_blocks']}, rewinds {p['history_rewinds']}, "
              f"{time.time() - t0:.0f} s", flush=True)
p = gpu.profile()
print(f"soak ok: {done} steps, worst relative error {worst:.2e}; schedule answers ahead / at start {p['schedule_blocks_ahead']} / {p['schedule_blocks_at_start']}, "
      f"blocks that started with rows made ahead {p['ahead_blocks']}, stretches in which both answers occurred {flips}, rewinds {p['history_rewinds']}, "
      f"aql {p['direct_dispatches']}, hip {p['hip_launches']}")
