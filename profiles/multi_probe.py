"""hc_step_multi latency: one coupled array row-sharded over G contexts of ONE process (all on GPU 0 of this box), synchronous steps.
   python profiles/multi_probe.py [N=64] [G list=1,2,4,8] [steps=400]
Prints per G: mean / median / p90 microseconds per evaluation, and how the kernels reached the GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench as B  # noqa: E402
from hydrochrono_amd.hydro import HydroForces, HydroGroup  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402
from hydrochrono_amd.parallel_split import body_shard  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Gs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
sdt = 0.01
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
nhist = int(np.ceil(B.S_RIRF * B.DT / sdt)) + 5
t_hist = B.T0 - sdt * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
times = [B.T0 + k * sdt for k in range(nsteps + 64)]
states = [motion.state(t) for t in times]
ref = None
for G in Gs:
    shards = []
    for g in range(G):
        h = HydroForces(N, device=0, body_range=body_shard(N, G, g))
        h.synth_fill(20251031, B.S_RIRF, B.DT, B.N_EXC, B.DT)
        h.finalize()
        shards.append(h)
    grp = HydroGroup(shards)
    grp.add_waves_irregular(**dict(B.WAVES, num_bodies=N, simulation_dt=sdt, simulation_duration=B.T0 + (nsteps + 100) * sdt + 5.0))
    grp.set_lookahead(int(os.environ.get("LA", "32")))
    grp.set_history(t_hist, v_hist)
    out = []
    for k in range(64):
        out.append(grp.step(times[k], *states[k]))
    per = []
    for k in range(64, 64 + nsteps):
        a = time.perf_counter()
        f = grp.step(times[k], *states[k])
        per.append(time.perf_counter() - a)
        out.append(f)
    out = np.stack(out)
    if ref is None:
        ref = out
    per = np.array(per) * 1e6
    p = [h.profile() for h in shards]
    print(f"N={N} G={G}: mean {per.mean():.1f} us  median {np.median(per):.1f} us  p90 {np.percentile(per, 90):.1f} us   "
          f"bitwise_vs_first_G={bool(np.array_equal(out, ref))}  direct={[h.direct_dispatch()[0] for h in shards]}  "
          f"aql={sum(x['direct_dispatches'] for x in p)} hip={sum(x['hip_launches'] for x in p)}", flush=True)
    grp.close()
