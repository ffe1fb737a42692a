import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hydrochrono_amd.mock_chrono import PrescribedMotion
N = 512
W = int(os.environ.get("W", "1"))
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
sdt = 0.01
nhist = int(np.ceil(B.S_RIRF * B.DT / sdt)) + 5
t_hist = B.T0 - sdt * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
gpu = B.make_shard(N, 0, N // W, 0, sdt, B.T0 + 100 * sdt, 0, t_hist, v_hist)
gpu.enable_profiling(1)
for k in range(12):
    t = B.T0 + k * sdt
    gpu.step(t, *motion.state(t))
p = gpu.profile()
us = p["conv_kernel_seconds"] / p["conv_kernel_launches"] * 1e6
print(f"W={W}: plain conv_step_kernel {us:.1f} us x {p['conv_kernel_launches']}  bytes {p['conv_kernel_bytes']/1e9:.3f} GB -> {p['conv_kernel_bytes']/us/1e6:.2f} TB/s")
