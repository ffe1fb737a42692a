"""Soak of the one-process multi-GPU call with its worker threads (hc_fanout.hpp): a coupled array in G row-shard contexts (all on GPU 0
here), N hc_step_multi calls with an hc_added_mass_mv_multi after every third one, pauses (workers fall asleep: 1 ms of spinning, then a
condition variable) and off-grid steps sprinkled in -- every gathered vector bitwise the unsharded context's, every 16th step against the
flat CPU oracle.  python profiles/soak_multi.py [steps] [G]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from cases import load_into_oracle  # noqa: E402
from hydrochrono_amd.hydro import HydroForces, HydroGroup  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402
from hydrochrono_amd.synthetic import many_body_case, rest_positions  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = 16
case = many_body_case(N, S=256, dt_rirf=0.01, n_exc=65, dt_exc=0.02, seed=777)
kw = dict(simulation_dt=0.01, simulation_duration=nsteps * 0.0102 + 30.0, ramp_duration=1.0, wave_height=2.0, wave_period=7.0, frequency_min=0.03,
          frequency_max=0.5, nfrequencies=64, peak_enhancement_factor=3.3)
full, group, orc = HydroForces.from_case(case), HydroGroup.from_case(case, G), load_into_oracle(case)
for h in (full, group, orc):
    h.add_waves_irregular(**kw)
motion = PrescribedMotion(N, rest_positions(case), seed=2)
rng = np.random.default_rng(5)
w, R0 = rng.normal(size=6 * N), rng.normal(size=6 * N)
t, worst, t0, naps = 0.0, 0.0, time.time(), 0
for n in range(nsteps):
    st = motion.state(t)
    fg = group.step(t, *st)
    ff = full.step(t, *st)
    if not np.array_equal(fg, ff):
        print(f"step {n}: the gathered vector differs from the unsharded context's")
        sys.exit(1)
    if n % 16 == 0:
        fo = orc.step(t, *st)
        worst = max(worst, float(np.max(np.abs(fg - fo)) / np.max(np.abs(fo))))
        if worst > 1e-9:
            print(f"step {n}: relative error {worst:.3e} against the oracle")
            sys.exit(1)
    else:
        orc.step(t, *st)
    if n % 3 == 0 and not np.array_equal(group.added_mass_mv(R0, w, 0.3), full.added_mass_mv(R0, w, 0.3)):
        print(f"step {n}: added-mass product differs")
        sys.exit(1)
    r = rng.random()
    t += 0.01 if r > 0.003 else 0.01 * rng.uniform(0.5, 1.5)
    if r > 0.97:
        time.sleep(0.003)  # longer than the workers spin: they sleep and are woken by the next call
        naps += 1
    if n % 5000 == 4999:
        print(f"{n + 1} steps, worst {worst:.2e}, naps {naps}, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {nsteps} steps of hc_step_multi over {G} contexts, bitwise the unsharded context, worst relative error against the oracle {worst:.2e}, {naps} naps")
