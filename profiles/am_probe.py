"""hc_added_mass_mv at wide sizes: one C4/8 rank's rows (384 x 3072) and the whole C4 matrix (3072 x 3072) on one GPU, median latency of the
synchronous product through the C ABI (w and R in through the BAR, tagged results out).
(Round 4: a variant that copies w into LDS once per workgroup -- every wave reads all of w from uncached memory -- measured 13.3 / 25.5 us
against 13.1 / 23.0-23.5 us for the one-wave-per-row kernel on the same box: the re-reads of w are not what bounds the product.  Not kept.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hydrochrono_amd.hydro as hydro
from hydrochrono_amd import capi
import ctypes as C
lib = capi.load()
for label, shard in (("C4/8 rank: 384 x 3072", (0, 64)), ("C4 on one GPU: 3072 x 3072", None)):
    ctx = C.c_void_p()
    rc = lib.hc_create_sharded(512, shard[0], shard[1], 0, C.byref(ctx)) if shard else lib.hc_create(512, 0, C.byref(ctx))
    assert rc == 0
    assert lib.hc_synth_fill(ctx, 20251031, 8, 0.01, 0, 0.01) == 0 and lib.hc_finalize(ctx) == 0
    rng = np.random.default_rng(0)
    w, R = rng.normal(size=3072), rng.normal(size=3072)
    lat = []
    for k in range(400):
        a = time.perf_counter()
        rc = lib.hc_added_mass_mv(ctx, w.ctypes.data_as(C.POINTER(C.c_double)), 0.5, R.ctypes.data_as(C.POINTER(C.c_double)), 3072)
        lat.append(time.perf_counter() - a)
        assert rc == 0
    lat = np.array(lat[50:]) * 1e6
    print(f"{label}: hc_added_mass_mv median {np.median(lat):.1f} us  p10 {np.percentile(lat, 10):.1f}  p90 {np.percentile(lat, 90):.1f}")
    lib.hc_destroy(ctx)
