"""Soak of the pass lane at real size: one rank's share of C4 (rows of 64 of 512 bodies, 9.7 GB of K) stepped N times under the pass
schedule "one block ahead" with the passes on the pass lane -- pauses and off-grid steps sprinkled in -- against the same steps
under the schedule "pass at block start" (a context of its own, run first).  python profiles/soak_wide_lane.py [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from hydrochrono_amd.hydro import HydroForces
from hydrochrono_amd.mock_chrono import PrescribedMotion

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
N = 512
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
rng = np.random.default_rng(3)
dts, pauses = [], []
for n in range(nsteps):
    r = rng.random()
    dts.append(B.DT if r > 0.003 else B.DT * rng.uniform(0.5, 1.5))
    pauses.append(200e-6 if rng.random() > 0.8 else 0.0)
times = B.T0 + np.concatenate([[0.0], np.cumsum(dts[:-1])])
nhist = B.S_RIRF + 5
t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
runs = []
for sched in (0, 1):
    gpu = HydroForces(N, device=0, body_range=(0, 64))
    gpu.synth_fill(20251031, B.S_RIRF, B.DT, B.N_EXC, B.DT)
    gpu.finalize()
    gpu.add_waves_irregular(**dict(B.WAVES, num_bodies=N, simulation_dt=B.DT, simulation_duration=float(times[-1]) + 20.0))
    gpu.set_pass_schedule(sched)
    gpu.set_history(t_hist, v_hist)
    out = np.zeros((nsteps, gpu.D_local))
    t0 = time.time()
    for n in range(nsteps):
        out[n] = gpu.step(times[n], *motion.state(times[n]))
        if pauses[n]:
            time.sleep(pauses[n])
    p = gpu.profile()
    print(f"schedule {sched}: {nsteps} steps in {time.time() - t0:.1f} s; blocks without a pass of their own {p['ahead_blocks']}, launches on the pass lane {p['pass_lane_launches']}, "
          f"parkings {p['queue_parkings']}, aql {p['direct_dispatches']}, hip {p['hip_launches']}", flush=True)
    runs.append(out)
    gpu.close()
scale = np.max(np.abs(runs[0]), axis=1)
err = np.max(np.abs(runs[0] - runs[1]), axis=1) / scale
print(f"worst relative difference between the schedules over {nsteps} steps: {err.max():.2e} (step {int(err.argmax())})")
sys.exit(0 if err.max() <= 1e-11 else 1)
