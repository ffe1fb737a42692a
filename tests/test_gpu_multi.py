"""Multi-GPU inside ONE host process through the C ABI (hc_step_multi / hc_added_mass_mv_multi, hc_step_begin / hc_step_end) and
through the C++ plugin surface (TestHydro / ChronoHydroSystem with a device list): SURVEY 8e's drop-in variant -- the host holds
all body states, every GPU gets a state store, the host gathers the force rows.  The box has one GPU, so the shard contexts share
it; the code path (G contexts, G queues, all doorbells before any wait, host gather) is the one an 8-GPU node runs.

Bar: the gathered vector is BITWISE the unsharded one (same kernels, same per-row arithmetic and order), and <= 1e-10 of the CPU
oracle."""
import os
import subprocess

import numpy as np
import pytest

from cases import GOLDEN_DIR, four_body_case, load_into_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIGHT_TOL = 1e-10


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


@pytest.fixture(scope="module")
def hydro():
    import torch  # noqa: F401
    from hydrochrono_amd import hydro as h
    return h


WAVES = dict(simulation_dt=0.01, simulation_duration=8.0, ramp_duration=0.5, wave_height=2.0, wave_period=6.0,
             frequency_min=0.05, frequency_max=0.6, nfrequencies=48, peak_enhancement_factor=3.3)


@pytest.mark.parametrize("n_shards", [2, 4, 8])
@pytest.mark.parametrize("lookahead", [32, 0])
def test_step_multi_is_bitwise_the_unsharded_context(hydro, n_shards, lookahead):
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 8
    case = many_body_case(N, S=160, dt_rirf=0.01, n_exc=65, dt_exc=0.02, seed=900 + n_shards)
    case["g_sys"] = [0.3, -0.2, -9.7]
    full = hydro.HydroForces.from_case(case)
    group = hydro.HydroGroup.from_case(case, n_shards)
    orc = load_into_oracle(case)
    for h in (full, group, orc):
        h.add_waves_irregular(**WAVES)
    full.set_lookahead(lookahead)
    group.set_lookahead(lookahead)
    motion = PrescribedMotion(N, rest_positions(case), seed=4)
    rng = np.random.default_rng(1)
    t = 0.0
    for n in range(330):
        st = motion.state(t)
        ref, got = full.step(t, *st), group.step(t, *st)
        assert np.array_equal(ref, got), f"step {n}: gathered vector differs from the unsharded one"
        assert relerr(got, orc.step(t, *st)) <= TIGHT_TOL, f"step {n} vs oracle"
        assert np.array_equal(group.step(t, *st), got)  # the per-time cache of every shard answers a repeated time
        if n % 50 == 7:
            for a, b in zip(full.components(), group.components()):
                assert np.array_equal(a, b)
            w, R0 = rng.normal(size=6 * N + 5), rng.normal(size=6 * N + 5)  # a system with more coordinates than the hydro block
            Rf, Rg = full.added_mass_mv(R0, w[:6 * N + 5], 0.75), group.added_mass_mv(R0, w, 0.75)
            assert np.array_equal(Rf, Rg) and np.array_equal(Rg[6 * N:], R0[6 * N:])
            assert relerr(Rg[:6 * N], R0[:6 * N] + 0.75 * (full.added_mass_matrix() @ w[:6 * N])) <= 1e-13
        t += 0.01 if n % 120 != 100 else 0.0123  # an off-grid step drops the look-ahead block in every shard alike
    if lookahead:
        # the shard contexts really used look-ahead blocks and, on this box, the direct queue
        for h in group.shards:
            assert h.direct_dispatch()[0], h.direct_dispatch()[1]
            p = h.profile()
            assert p["direct_dispatches"] > 300 and p["history_rewinds"] == 0


def test_step_begin_end_and_error_paths(hydro):
    from hydrochrono_amd import capi
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    import ctypes as C
    N = 4
    case = many_body_case(N, S=96, dt_rirf=0.01, n_exc=33, seed=41)
    a, b = hydro.HydroForces.from_case(case), hydro.HydroForces.from_case(case)
    group = hydro.HydroGroup.from_case(case, 2)
    for h in (a, b, group):
        h.add_waves_none()
    motion = PrescribedMotion(N, rest_positions(case), seed=8)
    lib = a.lib
    dp = lambda x: x.ctypes.data_as(capi.c_double_p)  # noqa: E731
    out = np.empty(6 * N)
    for n in range(120):
        t = 0.01 * n
        st = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in motion.state(t)]
        assert lib.hc_step_begin(a.ctx, t, *[dp(x) for x in st]) == 0
        ref = b.step(t, *st)  # the host is free between the two halves
        assert lib.hc_step_end(a.ctx, dp(out)) == 0
        assert np.array_equal(out, ref)
    # protocol errors: end without begin, begin twice, a full step while one is pending
    assert lib.hc_step_end(a.ctx, dp(out)) == capi.HC_ERR_INVALID
    t = 1.2
    st = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in motion.state(t)]
    assert lib.hc_step_begin(a.ctx, t, *[dp(x) for x in st]) == 0
    assert lib.hc_step_begin(a.ctx, t + 0.01, *[dp(x) for x in st]) == capi.HC_ERR_INVALID
    # (the failed call leaves nothing pending, as documented, so the next begin/end pair works again)
    assert lib.hc_step_begin(a.ctx, t + 0.02, *[dp(x) for x in st]) == 0
    assert lib.hc_step_end(a.ctx, dp(out)) == 0
    # a failing shard step: every context of the group reports the message and none is left pending
    group.step(0.0, *motion.state(0.0))
    group.add_waves_none(num_bodies=1)  # wave model for fewer bodies than N: the reference's short force vector, an error here
    with pytest.raises(hydro.HydroError) as e:
        group.step(0.01, *motion.state(0.01))
    assert e.value.status == capi.HC_ERR_RUNTIME and "fewer bodies" in str(e.value)
    for h in group.shards:
        assert b"fewer bodies" in lib.hc_last_error(h.ctx)
    group.add_waves_none()
    f = group.step(0.02, *motion.state(0.02))
    assert np.all(np.isfinite(f))
    # contexts of different systems in one group
    other = hydro.HydroForces.from_case(many_body_case(2, S=32, n_exc=9, seed=3))
    ctxs = (C.c_void_p * 2)(a.ctx, other.ctx)
    z = np.zeros(3 * N)
    assert lib.hc_step_multi(ctxs, 2, 5.0, dp(z), dp(z), dp(z), dp(z), dp(out)) == capi.HC_ERR_INVALID


def _build(tmp_path, name):
    from hydrochrono_amd import build as hb
    hb.build()
    if not os.path.exists(hb.BEMIO_LIB):
        pytest.skip("libhdf5 not available: BEMIO reader not built")
    libdir = os.path.join(ROOT, "hydrochrono_amd", "lib")
    out = str(tmp_path / name)
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "cpp", "chrono_stub"),
                    os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", out, "-L", libdir, "-lhydrochrono_amd", f"-Wl,-rpath,{libdir}"],
                   check=True)
    return out


@pytest.mark.parametrize("mode", ["irregular", "regular"])
def test_sharded_plugin_surface_through_componentfunc(tmp_path, mode):
    """ChronoHydroSystem with 1, 2 and 4 shard contexts on the one GPU, forces read through ChForce -> ComponentFunc::GetVal (24
    callbacks per time, one hc_step_multi), the added mass through ChLoadAddedMass::LoadIntLoadResidual_Mv."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    exe = _build(tmp_path, "shards_test")
    case = four_body_case()
    N, nsteps, dt = 4, 260, 0.005  # IRF window 0.62 s: depth-32 blocks of 0.16 s are planned
    motion = PrescribedMotion(N, np.stack([b["cg"] for b in case["bodies"]]), seed=21)
    states = np.stack([motion.packed(n * dt) for n in range(nsteps)])
    spath = str(tmp_path / "states.bin")
    states.tofile(spath)
    outs = {}
    for G in (1, 2, 4):
        r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "four_body.h5"), str(N), spath, str(nsteps), repr(dt), str(G), mode],
                           check=True, capture_output=True, text=True)
        lines = r.stdout.strip().splitlines()
        prof = [[int(x) for x in ln.split()[1:]] for ln in lines if ln.startswith("PROF")]
        assert len(prof) == G
        for blocks, scatters, direct, hip, active in prof:
            assert blocks >= 4 and scatters >= 100, "look-ahead blocks were not in use"
            assert active == 1 and direct > nsteps, "the shard contexts did not use the direct queue"
        outs[G] = [ln for ln in lines if not ln.startswith("PROF")]
        assert len(outs[G]) == nsteps + 2 and outs[G][-2].startswith("MV ") and outs[G][-1].startswith("MVHOST ")
    assert outs[2] == outs[1] and outs[4] == outs[1]  # 17 significant digits: bitwise the unsharded object, forces and R
    # ... and the oracle on the same inputs
    orc = load_into_oracle(case)
    orc.set_gravity([0.3, -0.2, -9.7])
    if mode == "regular":
        orc.add_waves_regular(0.8, 0.55)
    else:
        orc.add_waves_irregular(**dict(WAVES, simulation_dt=dt))
    got = np.array([[float(x) for x in ln.split()] for ln in outs[4][:-2]])
    n3 = 3 * N
    for n in range(nsteps):
        st = states[n]
        fo = orc.step(n * dt, st[:n3], st[n3:2 * n3], st[2 * n3:3 * n3], st[3 * n3:])
        assert relerr(got[n], fo) <= TIGHT_TOL, f"step {n}"
    n_sys = 6 * N + 6
    w, R0 = 0.1 * (np.arange(n_sys) + 1) - 0.7, 1.0 + 0.01 * np.arange(n_sys)
    M = np.concatenate([case["rho"] * np.asarray(b["added_mass_inf"]).reshape(6, 6 * N) for b in case["bodies"]])
    expect = R0.copy()
    expect[:6 * N] += 0.5 * (M @ w[:6 * N])
    # R += c M w: on the shards' GPUs (hc_added_mass_mv_multi, forced) and on the load's host copy (its default at 24 coordinates)
    for line in outs[4][-2:]:
        R = np.array([float(x) for x in line.split()[1:]])
        assert relerr(R, expect) <= 1e-13, line.split()[0]


@pytest.mark.parametrize("waves", ["regular", "irregular"])
def test_setup_hydro_from_yaml_with_one_and_two_shard_contexts(tmp_path, waves):
    """The YAML runner's lines -- ReadHydroYAML, then SetupHydroFromYAML(hydro_data, every body of the system, dt, duration, ramp
    [, devices]) (src/hydrochrono_runner/run_hydrochrono_from_yaml.cpp:440-457, src/setup_hydro_from_yaml.h:33-39) -- return an
    object that is WIRED into the ChSystem: forces are read through ChForce -> ComponentFunc::GetVal, the added mass through the
    load the constructor registered.  One, two and four shard contexts print the same 17 digits; totals against the oracle
    configured the way src/setup_hydro_from_yaml.cpp:28-79 maps the YAML (regular: A = H/2, omega = 2 pi/T; irregular: PM defaults)."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    exe = _build(tmp_path, "shards_test")
    case = four_body_case()
    N, nsteps, dt = 4, 200, 0.005
    h5 = os.path.join(GOLDEN_DIR, "four_body.h5")
    wave_block = ("    type: regular\n    height: 1.6\n    period: 11.0\n" if waves == "regular"
                  else "    type: irregular\n    height: 2.0\n    period: 6.0\n    seed: 3\n")
    bodies = "".join(f"    - name: body{b}\n      h5_file: {h5}\n" for b in range(1, N + 1))
    ypath = tmp_path / "array.hydro.yaml"
    ypath.write_text(f"hydrodynamics:\n  bodies:\n{bodies}  waves:\n{wave_block}")
    motion = PrescribedMotion(N, np.stack([b["cg"] for b in case["bodies"]]), seed=22)
    states = np.stack([motion.packed(n * dt) for n in range(nsteps)])
    spath = str(tmp_path / "states.bin")
    states.tofile(spath)
    outs = {}
    for G in (1, 2, 4):
        r = subprocess.run([exe, str(ypath), str(N), spath, str(nsteps), repr(dt), str(G), "yaml"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.strip().splitlines()
        assert sum(ln.startswith("PROF") for ln in lines) == G
        outs[G] = [ln for ln in lines if not ln.startswith("PROF")]
        assert len(outs[G]) == nsteps + 2 and outs[G][-2].startswith("MV ") and outs[G][-1].startswith("MVHOST ")
    assert outs[2] == outs[1] and outs[4] == outs[1]
    orc = load_into_oracle(case)
    orc.set_gravity([0.3, -0.2, -9.7])
    if waves == "regular":
        orc.add_waves_regular(0.8, 2.0 * np.pi / 11.0)
    else:
        orc.add_waves_irregular(simulation_dt=dt, simulation_duration=8.0, ramp_duration=0.5, wave_height=2.0, wave_period=6.0, seed=3)
    got = np.array([[float(x) for x in ln.split()] for ln in outs[2][:-2]])
    n3 = 3 * N
    for n in range(nsteps):
        st = states[n]
        fo = orc.step(n * dt, st[:n3], st[n3:2 * n3], st[2 * n3:3 * n3], st[3 * n3:])
        assert relerr(got[n], fo) <= TIGHT_TOL, f"step {n}"


@pytest.mark.parametrize("mode", [(32, 1), (16, 1), (32, 0), (0, 1)], ids=["la32-aql", "la16-aql", "la32-hip", "plain-aql"])
def test_wide_system_two_level_lookahead_against_oracle(hydro, mode, monkeypatch):
    """A WIDE system (D = 1056 >= 1024): the own-sample part of a block step is split over column slices (near_split_kernel), and the
    look-ahead runs in its two-level form (sub-blocks of 8 steps, a short pass after each, scatter inside the sub-block only) --
    what every rank of C4 (D = 3072) runs.  Totals and components against the CPU oracle, and three row shards bitwise."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    lookahead, direct = mode
    N = 176
    case = many_body_case(N, S=48, dt_rirf=0.02, n_exc=17, dt_exc=0.05, seed=1056)
    monkeypatch.setenv("HC_DIRECT", str(direct))
    full = hydro.HydroForces.from_case(case)
    group = hydro.HydroGroup.from_case(case, 3)
    assert full.direct_dispatch()[0] == bool(direct), full.direct_dispatch()[1]
    orc = load_into_oracle(case)
    kw = dict(WAVES, simulation_duration=4.0)
    for h in (full, group, orc):
        h.add_waves_irregular(**kw)
    full.set_lookahead(lookahead)
    group.set_lookahead(lookahead)
    motion = PrescribedMotion(N, rest_positions(case), seed=2)
    full.enable_profiling(1)
    t = 0.0
    for n in range(150):
        st = motion.state(t)
        fg = full.step(t, *st)
        assert relerr(fg, orc.step(t, *st)) <= TIGHT_TOL, f"step {n}"
        for g, o in zip(full.components(), orc.components()):
            assert relerr(g, o) <= TIGHT_TOL
        assert np.array_equal(group.step(t, *st), fg), f"step {n}: shards"
        t += 0.01 if n != 100 else 0.0137  # one off-grid step: back to plain steps and into a new block
    p = full.profile()
    if lookahead:
        assert p["block_kernel_launches"] >= 3 and p["mini_pass_launches"] >= 3 and p["scatter_kernel_launches"] >= 60, p
    else:
        assert p["block_kernel_launches"] == 0 and p["mini_pass_launches"] == 0
    assert (p["direct_dispatches"] > 0, p["hip_launches"] > 0) == (bool(direct), not direct)


def test_soak_eight_contexts_sixteen_queues_and_teardown(hydro):
    """Hardening of the hand-written queues: eight shard contexts in one process, each with BOTH lanes in use (the step path's queue
    and the added-mass queue: 16 HSA queues on the one GPU), a few thousand evaluations, then teardown and re-creation several times
    over (hsa_queue_destroy / hsa_shut_down against the HIP runtime's own reference) -- same forces every cycle, no error, no hang."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 8
    case = many_body_case(N, S=96, dt_rirf=0.01, n_exc=33, seed=808)
    motion = PrescribedMotion(N, rest_positions(case), seed=1)
    rng = np.random.default_rng(0)
    w, R0 = rng.normal(size=6 * N), rng.normal(size=6 * N)
    ref = None
    for cycle in range(4):
        group = hydro.HydroGroup.from_case(case, 8)
        group.add_waves_regular(0.5, 0.9)
        out = []
        for n in range(900 if cycle == 0 else 250):
            t = 0.01 * n
            out.append(group.step(t, *motion.state(t)))
            if n % 3 == 0:
                out.append(group.added_mass_mv(R0, w, 0.3))  # LoadIntLoadResidual_Mv between force evaluations, as under HHT
        for h in group.shards:
            assert h.direct_dispatch()[0]
            p = h.profile()
            assert p["hip_launches"] == 0 and p["direct_dispatches"] >= len(out)
        out = np.stack(out[:300])
        if ref is None:
            ref = out
        assert np.array_equal(out, ref), f"cycle {cycle}"
        if cycle % 2 == 0:
            group.close()      # explicit teardown ...
        else:
            del group          # ... or through the destructors


def test_hydro_yaml_setup_with_shards(hydro, tmp_path):
    """ReadHydroYAML + SetupHydroFromYAML (src/setup_hydro_from_yaml.cpp:126-193) for a system row-sharded over contexts of this
    process (hc_create_from_hydro_yaml_sharded): four bodies from a BEMIO file, irregular waves and TaperedDirect from the YAML,
    three shards -- bitwise the single-context setup of the same file."""
    import ctypes as C
    from hydrochrono_amd import build as hb, capi
    if not os.path.exists(hb.BEMIO_LIB):
        pytest.skip("libhdf5 not available: BEMIO reader not built")
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    case = four_body_case()
    y = tmp_path / "array.hydro.yaml"
    y.write_text("hydrodynamics:\n  bodies:\n" + "".join(f"    - name: body{b}\n      h5_file: {os.path.join(GOLDEN_DIR, 'four_body.h5')}\n" for b in (1, 2, 3, 4)) +
                 "  waves:\n    type: irregular\n    height: 1.5\n    period: 5.0\n    seed: 3\n"
                 "  convolution:\n    mode: TaperedDirect\n    taper:\n      start_percent: 0.6\n")
    names = ["ground", "body1", "body2", "body3", "body4"]
    single, matched = hydro.HydroForces.from_hydro_yaml(y, names, 0.005, 6.0, ramp_duration=0.3)
    assert matched == [1, 2, 3, 4]
    lib = capi.load()
    cfg, err = C.c_void_p(), C.create_string_buffer(2048)
    assert lib.hc_yaml_read(str(y).encode(), C.byref(cfg), err, 2048) == 0, err.value
    cn = (C.c_char_p * len(names))(*[n.encode() for n in names])
    devs = (C.c_int * 3)(0, 0, 0)
    ctxs = (C.c_void_p * 3)()
    mi, nm = (C.c_int * len(names))(), C.c_int()
    rc = lib.hc_create_from_hydro_yaml_sharded(cfg, cn, len(names), 0.005, 6.0, 0.3, devs, 3, ctxs, mi, C.byref(nm), err, 2048)
    lib.hc_yaml_free(cfg)
    assert rc == 0, err.value
    assert nm.value == 4 and list(mi[:4]) == [1, 2, 3, 4]
    shards = []
    for g in range(3):
        h = hydro.HydroForces.__new__(hydro.HydroForces)
        h.lib, h.ctx, h.N, h.D = lib, C.c_void_p(ctxs[g]), 4, 24
        b0, b1 = C.c_int(), C.c_int()
        assert lib.hc_get_shard(h.ctx, C.byref(b0), C.byref(b1)) == 0
        h.b0, h.b1, h.n_local, h.D_local = b0.value, b1.value, b1.value - b0.value, 6 * (b1.value - b0.value)
        shards.append(h)
    assert [(h.b0, h.b1) for h in shards] == [(0, 2), (2, 3), (3, 4)]
    group = hydro.HydroGroup(shards)
    motion = PrescribedMotion(4, np.stack([b["cg"] for b in case["bodies"]]), seed=3)
    for n in range(120):
        st = motion.state(0.005 * n)
        assert np.array_equal(group.step(0.005 * n, *st), single.step(0.005 * n, *st)), f"step {n}"
    # more shards than bodies / a missing body list are refused and leave nothing behind
    cfg = C.c_void_p()
    assert lib.hc_yaml_read(str(y).encode(), C.byref(cfg), err, 2048) == 0
    devs5, ctxs5 = (C.c_int * 5)(0, 0, 0, 0, 0), (C.c_void_p * 5)()
    assert lib.hc_create_from_hydro_yaml_sharded(cfg, cn, len(names), 0.005, 6.0, 0.3, devs5, 5, ctxs5, mi, C.byref(nm), err, 2048) == capi.HC_ERR_INVALID
    assert all(c is None for c in ctxs5)
    lib.hc_yaml_free(cfg)


def test_wide_system_short_passes_reach_far_when_the_step_exceeds_the_irf_spacing(hydro):
    """Two-level form with dt = 1.4 x dt_rirf: the short pass after the first sub-block covers ~46 IRF samples in chunks of half a
    sample (the largest partials footprint of the scheme); against the oracle and against the plain evaluation."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 176
    case = many_body_case(N, S=100, dt_rirf=0.01, n_exc=9, dt_exc=0.05, seed=77)
    a, b = hydro.HydroForces.from_case(case), hydro.HydroForces.from_case(case)
    orc = load_into_oracle(case)
    for h in (a, b, orc):
        h.add_waves_none()
    b.set_lookahead(0)
    a.enable_profiling(1)
    motion = PrescribedMotion(N, rest_positions(case), seed=2)
    for n in range(80):
        t = 0.014 * n
        st = motion.state(t)
        fa = a.step(t, *st)
        assert relerr(fa, b.step(t, *st)) <= 1e-11, f"step {n}: two-level vs plain"
        if n % 4 == 0 or n > 70:
            assert relerr(fa, orc.step(t, *st)) <= TIGHT_TOL, f"step {n}"
        else:
            orc.step(t, *st)
    p = a.profile()
    assert p["block_kernel_launches"] >= 2 and p["mini_pass_launches"] >= 4, p


def test_c4_size_step_multi_eight_contexts_one_gpu(hydro):
    """Configuration C4 at FULL size (512 bodies, D = 3072, K = 77 GB generated in HBM) as eight row-shard contexts of this process,
    all on the one GPU, evaluated by hc_step_multi -- against ONE context holding the whole array: bitwise, through a pass, the short
    passes of the two-level form and an off-grid step."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.parallel_split import body_shard
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs 2 x 77 GB of HBM")
    N, S = 512, 1024

    def make(b0, b1):
        h = hydro.HydroForces(N, device=0, body_range=(b0, b1))
        h.synth_fill(20251031, S, 0.01, 0, 0.0)
        h.finalize()
        h.add_waves_none()
        return h

    full = make(0, N)
    group = hydro.HydroGroup([make(*body_shard(N, 8, g)) for g in range(8)])
    motion = PrescribedMotion(N, np.zeros((N, 3)), seed=3)
    t_hist = 5.0 - 0.01 * np.arange(1, S + 6)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    full.set_history(t_hist, v_hist)
    group.set_history(t_hist, v_hist)
    t = 5.0
    for n in range(45):
        st = motion.state(t)
        assert np.array_equal(group.step(t, *st), full.step(t, *st)), f"step {n}"
        t += 0.01 if n != 20 else 0.0123
    for h in group.shards + [full]:
        assert h.direct_dispatch()[0]
    p = group.shards[3].profile()
    assert p["hip_launches"] == 0 and p["direct_dispatches"] > 90
    group.close()
    full.close()


def test_wide_system_soak_two_level_vs_plain(hydro):
    """3 000 steps of a wide system (D = 1056) in the two-level look-ahead form against the plain per-step evaluation of the same
    inputs: many ring wrap-arounds, step-size changes (blocks dropped and re-planned), jitter, and steps back in time."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 176
    case = many_body_case(N, S=64, dt_rirf=0.02, n_exc=9, dt_exc=0.05, seed=5)
    a, b = hydro.HydroForces.from_case(case), hydro.HydroForces.from_case(case)
    for h in (a, b):
        h.add_waves_regular(0.4, 0.9)
    b.set_lookahead(0)
    a.enable_profiling(1)
    motion = PrescribedMotion(N, rest_positions(case), seed=9)
    rng = np.random.default_rng(3)
    t, dt, worst = 0.0, 0.01, 0.0
    for n in range(3000):
        st = motion.state(t)
        worst = max(worst, relerr(a.step(t, *st), b.step(t, *st)))
        if n % 400 == 399:
            dt = float(rng.choice([0.01, 0.007, 0.013, 0.02]))
        if n % 701 == 700:
            t -= 2.5 * dt                      # the integrator rejected the last steps
        elif 1200 <= n < 1260:
            t += dt * rng.uniform(0.6, 1.4)    # jitter: every prediction misses
        else:
            t += dt
    assert worst <= 1e-10, worst
    p = a.profile()
    assert p["block_kernel_launches"] >= 60 and p["mini_pass_launches"] >= 150 and p["history_rewinds"] == 4, p


@pytest.mark.parametrize("threads", ["1", "2"])
def test_step_multi_with_fewer_worker_threads_than_contexts(threads):
    """HC_MULTI_THREADS < n_ctx - 1: the calling thread then runs the surplus items as well as item 0, and the HIP device
    it has current is whatever its last item set -- the device cache of the fan-out is keyed on the kind of thread, not on the item.
    Five shard contexts, one or two workers (items 2 .. 4 or 3 .. 4 on the caller): hc_step_multi and hc_added_mass_mv_multi call
    after call, bitwise the unsharded context.  (One device here; the CPU test tests/cpp/fanout_test.cpp pins which thread runs what.)"""
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch\n"
        "from hydrochrono_amd import hydro\n"
        "from hydrochrono_amd.mock_chrono import PrescribedMotion\n"
        "from hydrochrono_amd.synthetic import many_body_case, rest_positions\n"
        "N = 5\n"
        "case = many_body_case(N, S=120, dt_rirf=0.01, n_exc=33, dt_exc=0.02, seed=77)\n"
        "full, group = hydro.HydroForces.from_case(case), hydro.HydroGroup.from_case(case, 5)\n"
        "for h in (full, group): h.add_waves_none()\n"
        "motion = PrescribedMotion(N, rest_positions(case), seed=3)\n"
        "w = np.linspace(-1.0, 1.0, 6 * N); R0 = np.linspace(0.5, 1.5, 6 * N)\n"
        "for n in range(260):\n"
        "    st = motion.state(0.01 * n)\n"
        "    assert np.array_equal(full.step(0.01 * n, *st), group.step(0.01 * n, *st)), n\n"
        "    if n %% 13 == 0: assert np.array_equal(full.added_mass_mv(R0, w, 0.5), group.added_mass_mv(R0, w, 0.5)), n\n"
        "print('calls', group.shards[4].profile()['multi_calls'])\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HC_MULTI_THREADS=threads))
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == "calls 260", r.stdout[-500:]
