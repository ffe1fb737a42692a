"""Kernel-duration ablation of the per-step launches (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from hydrochrono_amd.hydro import HydroForces
from hydrochrono_amd.mock_chrono import PrescribedMotion
from hydrochrono_amd.synthetic import many_body_case, rest_positions
case = many_body_case(64, S=1024, dt_rirf=0.01, n_exc=1024, dt_exc=0.01)
gpu = HydroForces.from_case(case)
gpu.add_waves_irregular(simulation_dt=0.01, simulation_duration=60.0, wave_height=2.0, wave_period=8.0, frequency_min=0.02,
                        frequency_max=0.5, nfrequencies=512, peak_enhancement_factor=3.3)
m = PrescribedMotion(64, rest_positions(case))
mode = sys.argv[1]
if mode == "waves":
    for k in range(300): gpu.compute_waves(20.0 + 0.01 * k)
elif mode == "hs":
    st = m.state(0.0)
    for k in range(300): gpu.compute_hydrostatics(st[0], st[1])
elif mode == "nowave_steps":
    gpu.add_waves_none()
    t_hist = 20.0 - 0.01 * np.arange(1, 1030)
    gpu.set_history(t_hist, np.stack([m.velocity6(t) for t in t_hist]))
    for k in range(320): gpu.step(20.0 + 0.01 * k, *m.state(20.0 + 0.01 * k))
