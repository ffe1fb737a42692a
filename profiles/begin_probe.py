"""Host cost of the first half of a step (hc_step_begin: state through the BAR + the step kernel's packets + what later steps need)
for one C4/8 row shard -- what hc_step_multi spends per context before the next context's GPU gets its doorbell."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import bench as B  # noqa: E402
from hydrochrono_amd import capi  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
sdt = 0.01
nhist = int(np.ceil(B.S_RIRF * B.DT / sdt)) + 5
t_hist = B.T0 - sdt * np.arange(1, nhist + 1)
v_hist = np.stack([motion.velocity6(t) for t in t_hist])
gpu = B.make_shard(N, 0, N // W, 0, sdt, B.T0 + 400 * sdt, 32, t_hist, v_hist)
lib = capi.load()
begin = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(("hc_step_begin", lib))
end = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(("hc_step_end", lib))
n = 300
times = [B.T0 + k * sdt for k in range(n)]
states = np.ascontiguousarray(np.stack([motion.packed(t) for t in times]))
out = np.zeros(gpu.D_local)
n3 = 3 * N
tb, te = [], []
for k in range(n):
    p = states.ctypes.data + k * states.strides[0]
    a = time.perf_counter()
    begin(gpu.ctx, times[k], p, p + 8 * n3, p + 16 * n3, p + 24 * n3)
    b = time.perf_counter()
    end(gpu.ctx, out.ctypes.data)
    c = time.perf_counter()
    tb.append(b - a)
    te.append(c - b)
tb, te = np.array(tb[60:]) * 1e6, np.array(te[60:]) * 1e6
print(f"N={N} shard 1/{W}: hc_step_begin median {np.median(tb):.2f} us (p90 {np.percentile(tb, 90):.2f}, max {tb.max():.1f});  hc_step_end median {np.median(te):.1f} us mean {te.mean():.1f}")
