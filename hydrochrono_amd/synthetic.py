"""Seeded synthetic many-body inputs (C2 two-body stand-in, C3 64 bodies, C4 512 bodies).

All multi-body BEMIO files of the reference are missing blobs, so multi-body runs use these generators.  The values
are raw, unscaled "file" quantities in BEMIO layout, so they go through the same ingest as real data:
    K[b][i][c][s] = a * exp(-tau_s / tau_d) * cos(om * tau_s),   tau_s = s * dt
with (a, tau_d, om) drawn per (row, col) from a counter-based splitmix64 stream; same-body blocks x10.
The formula matches synth_rirf_kernel in csrc/hc_kernels.hip (hc_synth_fill), which generates the same tensor
directly in HBM for sizes that do not fit on the host.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def u01(h):
    return (np.asarray(h, dtype=np.uint64) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def rirf_params(rows, D, seed):
    """(amp, tau_d, om) arrays of shape [len(rows), D] for global rows `rows`."""
    rows = np.asarray(rows, dtype=np.uint64)[:, None]
    cols = np.arange(D, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) ^ ((rows << np.uint64(32)) | cols))
        ua, ud, uo = u01(splitmix64(base + np.uint64(1))), u01(splitmix64(base + np.uint64(2))), u01(splitmix64(base + np.uint64(3)))
    amp = 2.0 * ua - 1.0
    same_body = (rows // np.uint64(6)) == (cols // np.uint64(6))
    amp = np.where(same_body, amp * 10.0, amp)
    return amp, 1.0 + 3.0 * ud, 0.5 + 2.5 * uo


def rirf_body(b, N, S, dt, seed):
    """K_b[6][D][S], unscaled, file order."""
    D = 6 * N
    amp, tau_d, om = rirf_params(np.arange(6 * b, 6 * b + 6), D, seed)
    tau = np.arange(S) * dt
    return amp[:, :, None] * np.exp(-tau[None, None, :] / tau_d[:, :, None]) * np.cos(om[:, :, None] * tau[None, None, :])


def many_body_case(N, S=1024, dt_rirf=0.01, n_exc=1024, dt_exc=0.01, nw=64, seed=20251031, rho=1000.0, g=9.81,
                   water_depth=float("inf")):
    """Raw-array case dict (same schema as tests/cases.sphere_case) for N coupled bodies."""
    rng = np.random.default_rng(seed)
    D = 6 * N
    rirf_t = np.arange(S) * dt_rirf
    ex_t = (np.arange(n_exc) - (n_exc - 1) * 0.5) * dt_exc
    w = np.linspace(0.05, 0.05 * nw, nw)
    # symmetric-ish positive-definite-ish added mass for the whole array
    G = rng.normal(size=(D, D)) * 0.05
    A_full = G + G.T + np.diag(100.0 + 50.0 * rng.uniform(size=D))
    side = int(np.ceil(np.sqrt(N)))
    bodies = []
    for b in range(N):
        lin = rng.normal(size=(6, 6))
        lin = 0.5 * (lin + lin.T) + np.diag(50.0 + 50.0 * rng.uniform(size=6))
        cg = np.array([20.0 * (b % side), 20.0 * (b // side), -2.0])
        a, wd, om = 1.0 + rng.uniform(size=6), 1.0 + 2.0 * rng.uniform(size=6), 0.5 + 1.5 * rng.uniform(size=6)
        ex_f = a[:, None] * np.exp(-(ex_t[None, :] ** 2) / (wd[:, None] ** 2)) * np.cos(om[:, None] * ex_t[None, :])
        bodies.append(dict(
            disp_vol=200.0 + 100.0 * rng.uniform(), cg=cg, cb=cg + np.array([0.0, 0.0, 0.1]) + 0.01 * rng.normal(size=3),
            lin=lin, added_mass_inf=A_full[6 * b:6 * b + 6, :].copy(), rirf_t=rirf_t,
            rirf_K=rirf_body(b, N, S, dt_rirf, seed),
            w=w, ex_mag=rng.uniform(0.1, 2.0, size=(6, 1, nw)), ex_phase=rng.uniform(-np.pi, np.pi, size=(6, 1, nw)),
            ex_irf_t=ex_t, ex_irf_f=ex_f.reshape(6, 1, n_exc)))
    return dict(N=N, rho=rho, g=g, water_depth=water_depth, bodies=bodies)


def rest_positions(case):
    return np.stack([np.asarray(b["cg"], dtype=np.float64) for b in case["bodies"]])
