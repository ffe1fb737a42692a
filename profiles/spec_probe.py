import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from cases import sphere_case
from hydrochrono_amd.hydro import HydroForces
kw = dict(simulation_dt=0.05, simulation_duration=300.0, ramp_duration=0.0, wave_height=2.0, wave_period=8.0,
          frequency_min=0.03, frequency_max=0.4, nfrequencies=256, peak_enhancement_factor=3.3, seed=3)
a = HydroForces.from_case(sphere_case()); a.add_waves_irregular(**kw)
b = HydroForces.from_case(sphere_case()); b.add_waves_irregular(spectral=True, **kw)
ts = 70.0 + 0.05*np.arange(3000)
fa = np.array([a.compute_waves(t) for t in ts]); fb = np.array([b.compute_waves(t) for t in ts])
for d in (0,2,4):
    num = np.sqrt(np.mean((fa[:,d]-fb[:,d])**2)); den = np.sqrt(np.mean(fa[:,d]**2))
    print(d, "rms_irf", den, "rms_diff", num, "rel", num/den if den>0 else None, "corr", np.corrcoef(fa[:,d], fb[:,d])[0,1] if den>0 else None)
