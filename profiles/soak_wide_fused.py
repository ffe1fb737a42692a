"""One rank's share of C4 (rows of 64 of 512 bodies: 24 row tiles x 8 column slices per block step) stepped N times with the wide step
as ONE launch (wide_step_kernel, the default) and as two (HC_WIDE_FUSED=0), each in a process of its own: the forces must be bitwise
the same -- any partial that the tile's finishing workgroup read before it was written through would show.
python profiles/soak_wide_fused.py [steps]      (parent: runs both children and compares)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[2] == "child":
    import bench as B
    from hydrochrono_amd.hydro import HydroForces
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    nsteps, out = int(sys.argv[1]), sys.argv[3]
    N = 512
    motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
    gpu = HydroForces(N, device=0, body_range=(0, 64))
    gpu.synth_fill(20251031, B.S_RIRF, B.DT, B.N_EXC, B.DT)
    gpu.finalize()
    gpu.add_waves_irregular(**dict(B.WAVES, num_bodies=N, simulation_dt=B.DT, simulation_duration=B.T0 + nsteps * B.DT + 20.0))
    nhist = B.S_RIRF + 5
    t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
    gpu.set_history(t_hist, np.stack([motion.velocity6(t) for t in t_hist]))
    rng = np.random.default_rng(11)
    f = np.empty((nsteps, gpu.D_local))
    t = B.T0
    for n in range(nsteps):
        f[n] = gpu.step(t, *motion.state(t))
        t += B.DT if rng.random() > 0.004 else B.DT * rng.uniform(0.6, 1.4)
    np.save(out, f)
    p = gpu.profile()
    print(p["wide_fused_steps"], p["direct_dispatches"], p["hip_launches"])
    sys.exit(0)

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
with tempfile.TemporaryDirectory() as d:
    res = {}
    for fused in ("1", "0"):
        out = os.path.join(d, f"f{fused}.npy")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(nsteps), "child", out], capture_output=True, text=True,
                           env=dict(os.environ, HC_WIDE_FUSED=fused, HYDROCHRONO_AMD_FLAVOR="tuning"))
        if r.returncode != 0:
            print(r.stderr[-3000:])
            sys.exit(1)
        res[fused] = (np.load(out), r.stdout.strip().splitlines()[-1])
    same = np.array_equal(res["1"][0], res["0"][0])
    print(f"{nsteps} steps of a C4/8 rank: one launch per block step (fused steps, AQL dispatches, HIP launches: {res['1'][1]}) against two "
          f"({res['0'][1]}): forces {'bitwise equal' if same else 'DIFFER'}; max |f| {np.max(np.abs(res['1'][0])):.3e}")
    sys.exit(0 if same else 1)
