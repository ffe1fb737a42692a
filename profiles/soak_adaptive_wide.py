"""Soak of the adaptive pass schedule on a WIDE shard whose slice of K is above the 12 GB line (rows of 128 of the 512-body array, 19.4 GB):
there the default answers "at block start" for a back-to-back caller and "one block ahead" (pass lane, two-level form, short passes
towards the next block) once the caller leaves gaps -- so a caller that changes its regime at random makes the schedule flip in both
directions at arbitrary places in a block.  No CPU oracle at this size: the same states go through a second context afterwards, pinned to
the pass at block start and stepped back to back; the two runs must agree to rounding (the schedules group the chunks of K differently).
    python profiles/soak_adaptive_wide.py [steps = 4000] [rows = 128]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import bench as B  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = B.N_BODIES_C4
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
nhist = B.S_RIRF + 5
t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
rng = np.random.default_rng(11)
plan, total = [], 0
while total < nsteps:
    n = int(rng.integers(5, 150))
    plan.append((rng.random() < 0.5, n, float(rng.uniform(30e-6, 300e-6))))
    total += n
times = B.T0 + B.DT * np.arange(total)
states = np.ascontiguousarray(np.stack([motion.packed(t) for t in times]))
duration = B.T0 + (total + 8) * B.DT + 5.0

gpu = B.make_shard(N, 0, rows, 0, B.DT, duration, 32, t_hist, v_hist)
got = np.empty((total, gpu.D_local))
k, flips, last, t0 = 0, 0, None, time.time()
for tight, n, gap in plan:
    if tight:
        got[k:k + n], _ = gpu.step_many(times[k:k + n], states[k:k + n])
    else:
        for i in range(k, k + n):
            got[i] = gpu.step(times[i], *motion.state(times[i]))
            b = time.perf_counter()
            while time.perf_counter() - b < gap:
                pass
    k += n
    p = gpu.profile()
    now = (p["schedule_blocks_ahead"], p["schedule_blocks_at_start"])
    if last is not None and now[0] > last[0] and now[1] > last[1]:
        flips += 1
    last = now
p = gpu.profile()
print(f"adaptive run: {total} steps in {len(plan)} stretches, {time.time() - t0:.0f} s; schedule answers ahead / at start {p['schedule_blocks_ahead']} / "
      f"{p['schedule_blocks_at_start']}, blocks that started with rows made ahead {p['ahead_blocks']}, on the pass lane {p['pass_lane_launches']} launches, "
      f"stretches in which both answers occurred {flips}; aql {p['direct_dispatches']}, hip {p['hip_launches']}", flush=True)
gpu.close()

ref = B.make_shard(N, 0, rows, 0, B.DT, duration, 32, t_hist, v_hist)
ref.set_pass_schedule(0)
want, _ = ref.step_many(times, states)
ref.close()
err = np.max(np.abs(got - want), axis=1) / np.max(np.abs(want), axis=1)
worst = int(np.argmax(err))
print(f"against the same states under the pinned schedule (pass at block start, back to back): worst relative difference {err.max():.2e} at step {worst}, "
      f"bitwise equal steps {int(np.sum(np.all(got == want, axis=1)))} of {total}")
if not err.max() <= 1e-11:
    sys.exit(1)
print("soak ok")
