"""Soak of the default C3 path (64 bodies, S = 1024, irregular waves, depth-32 look-ahead, direct dispatch, queue parking when the host
stays away): N steps against the flat-array CPU oracle, with pauses and off-grid steps sprinkled in.  python profiles/soak_c3.py [steps]"""
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

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
case = many_body_case(64, S=B.S_RIRF, dt_rirf=B.DT, n_exc=B.N_EXC, dt_exc=B.DT, seed=20251031)
gpu = HydroForces.from_case(case)
motion = PrescribedMotion(64, rest_positions(case), seed=20251031)
duration = B.T0 + nsteps * 0.0101 + 30.0
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
rng = np.random.default_rng(1)
t, worst, t0 = B.T0, 0.0, time.time()
for n in range(nsteps):
    st = motion.state(t)
    fg, fo = gpu.step(t, *st), orc.flat_step(t, *st)
    e = float(np.max(np.abs(fg - fo)) / np.max(np.abs(fo)))
    worst = max(worst, e)
    if e > 1e-9:
        print(f"step {n} t {t}: relative error {e:.3e}")
        sys.exit(1)
    r = rng.random()
    t += B.DT if r > 0.002 else B.DT * rng.uniform(0.5, 1.5)   # an off-grid step now and then
    if r > 0.9:
        time.sleep(150e-6)                                      # the host is away: queue parking
    if n % 5000 == 4999:
        p = gpu.profile()
        print(f"{n + 1} steps, worst {worst:.2e}, parkings {p['queue_parkings']}, aql {p['direct_dispatches']}, hip {p['hip_launches']}, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {nsteps} steps, worst relative error {worst:.2e}")
