"""One rank's share of C4 (512-body coupled array row-sharded over 8 ranks) on one GPU: synchronous hc_step timing + kernel profile."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hydrochrono_amd.hydro import HydroForces
from hydrochrono_amd.mock_chrono import PrescribedMotion

N, world, rank = 512, int(os.environ.get("W", "8")), 0
b0, b1 = 0, N // world
gpu = HydroForces(N, device=0, body_range=(b0, b1))
gpu.synth_fill(20251031, B.S_RIRF, B.DT, B.N_EXC, B.DT)
gpu.finalize()
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
sdt = 0.01
nsteps = 200
duration = B.T0 + (nsteps + 100) * sdt + 5.0
gpu.add_waves_irregular(**dict(B.WAVES, num_bodies=N, simulation_dt=sdt, simulation_duration=max(B.WAVES["simulation_duration"], duration)))
gpu.set_lookahead(int(os.environ.get("LA", "32")))
nhist = int(np.ceil(B.S_RIRF * B.DT / sdt)) + 5
t_hist = B.T0 - sdt * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
gpu.set_history(t_hist, v_hist)
times = [B.T0 + k * sdt for k in range(nsteps)]
states = [motion.state(t) for t in times]
for k in range(40):
    gpu.step(times[k], *states[k])
gpu.enable_profiling(1)
gpu.reset_profile()
per = []
for k in range(40, nsteps):
    a = time.perf_counter(); gpu.step(times[k], *states[k]); per.append(time.perf_counter() - a)
p = gpu.profile()
per = np.array(per) * 1e6
print("rows %d cols %d: step mean %.1f us median %.1f us p90 %.1f" % (gpu.D_local, 6 * N, per.mean(), np.median(per), np.percentile(per, 90)))
us = lambda s, n: s / max(n, 1) * 1e6
print("pass %.1f us x %d (bytes once %.3f GB -> %.2f TB/s)  step kernel %.1f us  scatter %.1f us" % (
    us(p["block_kernel_seconds"], p["block_kernel_launches"]), p["block_kernel_launches"], p["block_kernel_bytes_once"] / 1e9,
    p["block_kernel_bytes_once"] / max(p["block_kernel_seconds"] / max(p["block_kernel_launches"], 1), 1e-12) / 1e12,
    us(p["step_kernel_seconds"], p["step_kernel_launches"]), us(p["scatter_kernel_seconds"], p["scatter_kernel_launches"])))
print("short passes (two-level form): %.1f us x %d;  per step of the timed region: pass %.1f  short passes %.1f  scatter %.1f  step kernels %.1f us" % (
    us(p["mini_pass_seconds"], p["mini_pass_launches"]), p["mini_pass_launches"],
    p["block_kernel_seconds"] / len(per) * 1e6, p["mini_pass_seconds"] / len(per) * 1e6, p["scatter_kernel_seconds"] / len(per) * 1e6,
    p["step_kernel_seconds"] / len(per) * 1e6))
print("dispatch:", gpu.direct_dispatch(), " aql %d hip %d" % (p["direct_dispatches"], p["hip_launches"]))
