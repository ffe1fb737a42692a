"""Pass schedule "one block ahead" (hc_set_pass_schedule(ctx, 1)): the pass of the NEXT look-ahead block is computed
while the current one is stepped -- in slices behind its first steps -- and the current block's own samples reach the next block's
steps through short passes.  Parity of that schedule against the CPU oracle (same tolerance as every other path), against the
default schedule, and bitwise between row shards; the counters prove that blocks really started without a pass of their own."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cases import load_into_oracle  # noqa: E402

pytestmark = pytest.mark.gpu

TIGHT_TOL = 1e-11
WAVES = dict(simulation_dt=0.01, ramp_duration=1.0, wave_height=2.5, wave_period=8.0, frequency_min=0.02, frequency_max=0.5,
             nfrequencies=64, peak_enhancement_factor=3.3, seed=1)


@pytest.fixture(scope="module")
def hydro():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import hydrochrono_amd.hydro as hydro
    return hydro


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, float(np.max(np.abs(b)))))


@pytest.mark.parametrize("direct", [1, 0], ids=["aql", "hip"])
@pytest.mark.parametrize("lookahead,sub", [(32, 0), (16, 0), (32, 8), (16, 8), (32, 4)], ids=["la32", "la16", "la32-sub8", "la16-sub8", "la32-sub4"])
@pytest.mark.parametrize("N,dt", [(2, 0.01), (3, 0.007), (4, 0.013), (8, 0.01)])
def test_one_block_ahead_against_oracle(hydro, N, dt, lookahead, sub, direct, monkeypatch, tuning_build):
    """Small systems, both look-ahead forms (single level; sub-blocks forced with HC_SUB_BLOCK), both dispatch paths, step sizes
    equal to, below and above the IRF spacing; irregular waves (the excitation rows travel with the pass in the making)."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_DIRECT", str(direct))
    monkeypatch.setenv("HC_SUB_BLOCK", str(sub))
    case = many_body_case(N, S=257, dt_rirf=0.01, n_exc=101, dt_exc=0.02, seed=300 + N)
    case["g_sys"] = [0.3, -0.2, -9.7]
    gpu, orc = hydro.HydroForces.from_case(case), load_into_oracle(case)
    assert gpu.direct_dispatch()[0] == bool(direct), gpu.direct_dispatch()[1]
    kw = dict(WAVES, simulation_dt=dt, simulation_duration=9.0)
    gpu.add_waves_irregular(**kw)
    orc.add_waves_irregular(**kw)
    gpu.set_lookahead(lookahead)
    gpu.set_pass_schedule(1)
    gpu.enable_profiling(1)
    motion = PrescribedMotion(N, rest_positions(case), seed=N)
    nsteps = 560
    for n in range(nsteps):
        t = dt * n
        st = motion.state(t)
        fg, fo = gpu.step(t, *st), orc.step(t, *st)
        assert relerr(fg, fo) <= TIGHT_TOL, f"step {n} (t = {t})"
        if n % 7 == 0:
            for g, o in zip(gpu.components(), orc.components()):
                assert relerr(g, o) <= TIGHT_TOL, f"step {n}: components"
    p = gpu.profile()
    # the history covers the IRF window after 257 * 0.01 / dt steps; from then on blocks start without a pass of their own
    full_at = int(2.57 / dt) + 2 * lookahead + 2
    assert p["ahead_blocks"] >= (nsteps - full_at) // lookahead - 1 and p["ahead_blocks"] >= 3, p
    assert p["ahead_pass_slices"] >= p["ahead_blocks"], p
    assert (p["direct_dispatches"] > 0, p["hip_launches"] > 0) == (bool(direct), not direct) or not direct, p
    # with direct dispatch (and the device to itself) the passes in the making run on the pass lane, beside the steps
    assert (p["pass_lane_launches"] > 0) == bool(direct), p


@pytest.mark.parametrize("slices", [1, 3, 16, 31])
@pytest.mark.parametrize("sub", [0, 8])
def test_one_block_ahead_slice_counts(hydro, slices, sub, monkeypatch, tuning_build):
    """The number of launches the pass in the making is spread over (and with it its chunk length) is the caller's choice."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_SUB_BLOCK", str(sub))
    N = 5
    case = many_body_case(N, S=200, dt_rirf=0.01, n_exc=41, dt_exc=0.02, seed=77)
    gpu, orc = hydro.HydroForces.from_case(case), load_into_oracle(case)
    kw = dict(WAVES, simulation_duration=6.0)
    gpu.add_waves_irregular(**kw)
    orc.add_waves_irregular(**kw)
    gpu.set_pass_schedule(1, slices)
    motion = PrescribedMotion(N, rest_positions(case), seed=N)
    for n in range(400):
        st = motion.state(0.01 * n)
        assert relerr(gpu.step(0.01 * n, *st), orc.step(0.01 * n, *st)) <= TIGHT_TOL, f"step {n}"
    p = gpu.profile()
    assert p["ahead_blocks"] >= 3, p
    per_block = p["ahead_pass_slices"] / (p["ahead_blocks"] + 1)
    assert per_block <= min(slices, sub if sub else 31) + 1e-9, p


@pytest.mark.parametrize("sub", [0, 8])
def test_one_block_ahead_without_the_pass_lane(hydro, sub, monkeypatch, tuning_build):
    """HC_PASS_CONCURRENT=0: the slices and the short passes towards the next block stay on the step path's lane (what contexts
    that share a device do) -- the same arithmetic, bitwise the forces of a context that uses the pass lane."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_SUB_BLOCK", str(sub))
    N = 4
    case = many_body_case(N, S=200, dt_rirf=0.01, n_exc=41, dt_exc=0.02, seed=78)
    kw = dict(WAVES, simulation_duration=6.0)
    motion = PrescribedMotion(N, rest_positions(case), seed=N)
    runs = []
    for concurrent in ("1", "0"):
        monkeypatch.setenv("HC_PASS_CONCURRENT", concurrent)
        gpu = hydro.HydroForces.from_case(case)
        gpu.add_waves_irregular(**kw)
        gpu.set_pass_schedule(1)
        f = np.stack([gpu.step(0.01 * n, *motion.state(0.01 * n)) for n in range(420)])
        p = gpu.profile()
        assert p["ahead_blocks"] >= 4 and (p["pass_lane_launches"] > 0) == (concurrent == "1"), p
        runs.append(f)
        gpu.close()
    assert np.array_equal(runs[0], runs[1])
    orc = load_into_oracle(case)
    orc.add_waves_irregular(**kw)
    for n in range(420):
        assert relerr(runs[0][n], orc.step(0.01 * n, *motion.state(0.01 * n))) <= TIGHT_TOL, f"step {n}"


def test_one_block_ahead_survives_off_grid_steps_and_steps_back(hydro):
    """The pass in the making is dropped with the block it belongs to: off-grid steps (plain evaluation, new plan) and steps back in
    time (the newer samples are dropped; the oracle is rebuilt with the kept history, as in test_gpu_boundary.py) -- every force
    against the oracle, and blocks without a pass of their own again afterwards."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 3
    case = many_body_case(N, S=161, dt_rirf=0.01, n_exc=33, seed=41)
    gpu = hydro.HydroForces.from_case(case)
    gpu.add_waves_none()
    gpu.set_pass_schedule(1)
    motion = PrescribedMotion(N, rest_positions(case), seed=4)
    log = []  # (t, velocity) of the samples the history should hold, oldest first

    def fresh_oracle():
        o = load_into_oracle(case)
        o.add_waves_none()
        if log:
            o.prefill_history(np.array([t for t, _ in reversed(log)]), np.stack([v for _, v in reversed(log)]))
        return o

    orc = fresh_oracle()
    rng = np.random.default_rng(5)
    t, backs = 0.0, 0
    for n in range(1500):
        st = motion.state(t)
        assert relerr(gpu.step(t, *st), orc.step(t, *st)) <= TIGHT_TOL, f"step {n} (t = {t})"
        log.append((t, np.concatenate([st[2].reshape(N, 3), st[3].reshape(N, 3)], axis=1).reshape(-1)))
        log = log[-400:]
        r = rng.random()
        if n > 250 and r < 0.004:
            t -= 0.023          # a rejected step: back in time
            while log and log[-1][0] >= t:
                log.pop()
            orc = fresh_oracle()
            backs += 1
        elif n > 250 and r < 0.012:
            t += 0.01 * rng.uniform(0.3, 1.7)
        else:
            t += 0.01
    p = gpu.profile()
    assert p["ahead_blocks"] >= 8 and p["history_rewinds"] == backs and backs >= 2, p


@pytest.mark.parametrize("wait_each_step", [False, True], ids=["runs-ahead", "waits"])
def test_one_block_ahead_device_steps_on_a_callers_stream(hydro, wait_each_step):
    """hc_step_device on a caller's stream under the schedule -- enqueued ahead of the GPU (everything on that stream in order) and
    with a caller that waits for every step (slices and short passes go to the context's own stream behind events): bitwise the
    forces of synchronous hc_step under the same schedule, and both against the default schedule."""
    import torch
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 3
    case = many_body_case(N, S=150, n_exc=33, seed=91)
    a, b, ref = (hydro.HydroForces.from_case(case) for _ in range(3))
    for h in (a, b, ref):
        h.add_waves_none()
    a.set_pass_schedule(1)
    b.set_pass_schedule(1)
    motion = PrescribedMotion(N, rest_positions(case), seed=6)
    nsteps = 420
    states = torch.tensor(np.stack([motion.packed(0.01 * n) for n in range(nsteps)]), device="cuda")
    out = torch.zeros(nsteps, 6 * N, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    for n in range(nsteps):
        b.step_device(0.01 * n, states[n].data_ptr(), out[n].data_ptr(), stream.cuda_stream)
        if wait_each_step:
            stream.synchronize()
    torch.cuda.synchronize()
    host = np.stack([a.step(0.01 * n, *motion.state(0.01 * n)) for n in range(nsteps)])
    plain = np.stack([ref.step(0.01 * n, *motion.state(0.01 * n)) for n in range(nsteps)])
    assert np.array_equal(out.cpu().numpy(), host)
    assert relerr(host, plain) <= TIGHT_TOL
    assert a.profile()["ahead_blocks"] >= 5 and b.profile()["ahead_blocks"] >= 5


def test_one_block_ahead_reconfiguration_and_teardown_with_a_busy_pass_lane(hydro):
    """Calls that void the plan while a pass is in the making on the pass lane -- another look-ahead depth, a history reset, an
    injected history, the schedule itself -- and contexts destroyed right after a step (the lane still busy), several times over:
    every force against the oracle, no hang."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 6
    case = many_body_case(N, S=180, dt_rirf=0.01, n_exc=33, seed=17)
    motion = PrescribedMotion(N, rest_positions(case), seed=3)
    for cycle in range(4):
        gpu, orc = hydro.HydroForces.from_case(case), load_into_oracle(case)
        gpu.add_waves_none()
        orc.add_waves_none()
        gpu.set_pass_schedule(1)
        t, lane_seen = 0.0, 0
        for n in range(700):
            st = motion.state(t)
            assert relerr(gpu.step(t, *st), orc.step(t, *st)) <= TIGHT_TOL, f"cycle {cycle} step {n}"
            t += 0.01
            if n == 300:
                gpu.set_lookahead(16)            # mid-block: the plan and the pass in the making are dropped
            if n == 420:
                lane_seen = gpu.profile()["pass_lane_launches"]
                gpu.set_pass_schedule(0)
                gpu.set_pass_schedule(1, 3)
                gpu.set_lookahead(32)
            if n == 520:                          # a fresh start on both sides
                gpu.reset_history()
                orc = load_into_oracle(case)
                orc.add_waves_none()
        p = gpu.profile()
        assert lane_seen > 0 and p["pass_lane_launches"] > lane_seen and p["ahead_blocks"] >= 6, p
        gpu.step(t, *motion.state(t))            # leaves work on both lanes ...
        gpu.close()                               # ... for the destructor to wait for


def test_one_block_ahead_with_tapered_direct_and_per_body_excitation_grids(hydro):
    """The passes of the schedule read the processed kernel when TaperedDirect is on, and the excitation rows of the pass in the making
    cover bodies with excitation-IRF grids of their own (column groups of Kex): both against the oracle."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = 4
    case = many_body_case(N, S=200, dt_rirf=0.01, n_exc=81, dt_exc=0.02, seed=23)
    # bodies 2 and 3 get an excitation IRF on a grid of their own (other spacing and length)
    other = many_body_case(N, S=200, dt_rirf=0.01, n_exc=57, dt_exc=0.03, seed=24)
    for b in (2, 3):
        case["bodies"][b]["ex_irf_t"] = other["bodies"][b]["ex_irf_t"]
        case["bodies"][b]["ex_irf_f"] = other["bodies"][b]["ex_irf_f"]
    gpu, orc = hydro.HydroForces.from_case(case), load_into_oracle(case)
    kw = dict(WAVES, simulation_duration=7.0)
    for h in (gpu, orc):
        h.set_convolution_mode(1)
        h.set_tapered_direct_options(smoothing=0, window_length=5, rirf_end_time=1.6, taper_start_percent=0.6, taper_end_percent=0.9,
                                     taper_final_amplitude=0.1)
        h.add_waves_irregular(**kw)
    gpu.set_pass_schedule(1)
    motion = PrescribedMotion(N, rest_positions(case), seed=8)
    for n in range(450):
        st = motion.state(0.01 * n)
        assert relerr(gpu.step(0.01 * n, *st), orc.step(0.01 * n, *st)) <= TIGHT_TOL, f"step {n}"
        if n % 9 == 0:
            for g, o in zip(gpu.components(), orc.components()):
                assert relerr(g, o) <= TIGHT_TOL, f"step {n}: components"
    assert gpu.profile()["ahead_blocks"] >= 4


def test_one_block_ahead_c3_size_against_flat_oracle(hydro):
    """Full-size C3 (64 bodies, S = 1024, Nf = 512) from a steady-state history: 200 steps under the schedule -- plain boundary step,
    a block with its own pass, then blocks whose rows were made ahead -- against the flat-array CPU oracle."""
    import bench as B
    import oracle as orc_mod
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    case = many_body_case(64, S=B.S_RIRF, dt_rirf=B.DT, n_exc=B.N_EXC, dt_exc=B.DT, seed=20251031)
    gpu = hydro.HydroForces.from_case(case)
    motion = PrescribedMotion(64, rest_positions(case), seed=20251031)
    kw = dict(B.WAVES, simulation_dt=B.DT, simulation_duration=B.T0 + 8.0)
    gpu.add_waves_irregular(num_bodies=64, **kw)
    gpu.set_pass_schedule(1)
    orc_mod.set_num_threads(min(64, os.cpu_count() or 1))
    orc = load_into_oracle(case)
    orc.add_waves_irregular(**kw)
    nhist = B.S_RIRF + 5
    t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    gpu.set_history(t_hist, v_hist)
    orc.prefill_history(t_hist, v_hist)
    orc.flat_prepare()
    gpu.enable_profiling(1)
    worst = 0.0
    for n in range(200):
        t = B.T0 + n * B.DT
        st = motion.state(t)
        e = relerr(gpu.step(t, *st), orc.flat_step(t, *st))
        worst = max(worst, e)
        assert e <= TIGHT_TOL, f"step {n}"
    p = gpu.profile()
    assert p["ahead_blocks"] >= 5 and p["conv_kernel_launches"] == 1, p
    assert gpu.direct_dispatch()[0] and p["hip_launches"] == 0 and p["pass_lane_launches"] >= 5 * 5, p
    print(f"C3 one block ahead: worst relative error {worst:.2e}, {p['ahead_blocks']} blocks without a pass of their own, {p['ahead_pass_slices']} slices")


@pytest.mark.parametrize("direct", [1, 0], ids=["aql", "hip"])
def test_one_block_ahead_wide_system_and_row_shards(hydro, direct, monkeypatch):
    """A WIDE system (D = 1056: two-level look-ahead, sliced own-sample kernel) under the schedule: against the oracle, and three row
    shards evaluated by hc_step_multi bitwise equal to the unsharded context (the schedule is a function of the configuration only)."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    import oracle as orc_mod
    N = 176
    case = many_body_case(N, S=140, dt_rirf=0.01, n_exc=17, dt_exc=0.05, seed=1056)
    monkeypatch.setenv("HC_DIRECT", str(direct))
    monkeypatch.setenv("HC_PASS_AHEAD", "1")  # the default of every context created below
    full = hydro.HydroForces.from_case(case)
    group = hydro.HydroGroup.from_case(case, 3)
    orc_mod.set_num_threads(min(64, os.cpu_count() or 1))
    orc = load_into_oracle(case)
    kw = dict(WAVES, simulation_duration=6.0)
    for h in (full, group, orc):
        h.add_waves_irregular(**kw)
    motion = PrescribedMotion(N, rest_positions(case), seed=2)
    # start from a history that covers the IRF window (the flat-array oracle is prepared on it)
    t_hist = 1.0 - 0.01 * np.arange(1, 146)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    full.set_history(t_hist, v_hist)
    group.set_history(t_hist, v_hist)
    orc.prefill_history(t_hist, v_hist)
    orc.flat_prepare()
    full.enable_profiling(1)
    t = 1.0
    for n in range(260):
        st = motion.state(t)
        fg = full.step(t, *st)
        assert relerr(fg, orc.flat_step(t, *st)) <= TIGHT_TOL, f"step {n}"
        assert np.array_equal(group.step(t, *st), fg), f"step {n}: shards"
        t += 0.01 if n != 150 else 0.0137
    p = full.profile()
    assert p["ahead_blocks"] >= 4 and p["mini_pass_launches"] >= 20, p


@pytest.mark.parametrize("concurrent", ["1", "0"], ids=["pass-lane", "in-order"])
@pytest.mark.parametrize("sub", [0, 8])
def test_one_block_ahead_with_the_ring_nearly_full(hydro, sub, concurrent, monkeypatch, tuning_build):
    """The pass one block ahead keeps reading its view of the history while the block's steps push new samples into the ring, so
    the slots those pushes take must not belong to the view.  The ring is allocated with 64 slots beyond the IRF window; steps
    3-6 % below the IRF spacing fill them with KEPT samples (the history grows to Hcap - 1 or - 2 without triggering a grow), and
    a step-size change inside the window makes the older history non-uniform, so the bracket search of the pass falls back to
    the binary search over ring_t -- the reads an overwritten slot would corrupt.  The library re-allocates the ring before it
    takes such a view (ring_grows_for_pass); forces stay on the oracle throughout.  (With the guard disabled this small system
    still passes the parity assertions -- its pass is one launch that ends long before the overwriting pushes arrive, and the
    samples a next-block pass needs have aged past the overwritten ones unless the step shrank by more than L/(L - 3); the
    guard is for wide systems whose pass is spread over the block.  What is pinned here: the room is made, and making it
    mid-run -- a ring re-allocation under a running block -- leaves every force on the oracle.)"""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_SUB_BLOCK", str(sub))
    monkeypatch.setenv("HC_PASS_CONCURRENT", concurrent)
    N, S, dt_rirf = 4, 200, 0.01
    case = many_body_case(N, S=S, dt_rirf=dt_rirf, n_exc=41, dt_exc=0.02, seed=91)
    gpu, orc = hydro.HydroForces.from_case(case), load_into_oracle(case)
    for h in (gpu, orc):
        h.add_waves_none()
    gpu.set_pass_schedule(1, 31)  # in-order mode: slice k goes out behind k pushes, the last slices read the oldest samples
    cap0 = gpu.sizes()["Hcap"]
    assert cap0 == S + 2 + 64
    # steady H at step dt: (S - 1) * dt_rirf / dt + 2 (+ 1); aim two slots below the capacity
    dt_b = (S - 1) * dt_rirf / (cap0 - 4.5)
    dt_a = 1.5 * dt_b
    motion = PrescribedMotion(N, rest_positions(case), seed=9)
    t, worst, seen_tight = 0.0, 0.0, False
    for n in range(900):
        t += dt_a if n < 120 else dt_b  # the change lies inside the IRF window for the next ~260 steps
        st = motion.state(t)
        worst = max(worst, relerr(gpu.step(t, *st), orc.step(t, *st)))
        assert worst <= TIGHT_TOL, f"step {n} (t = {t}, H = {gpu.sizes()['H']})"
        sz = gpu.sizes()
        seen_tight = seen_tight or (sz["Hcap"] == cap0 and sz["Hcap"] - sz["H"] < 32)
    p, sz = gpu.profile(), gpu.sizes()
    assert p["ahead_blocks"] >= 10, p
    # the ring was re-allocated under a running block: by the pass's guard, or -- since the ring also keeps room for the samples the prune
    # rule has retired (a step back in time re-admits them, hc_history.hpp) -- by the history rule before the guard had to
    assert sz["Hcap"] > cap0, (p, sz, seen_tight)
    assert p["history_rewinds"] == 0


@pytest.mark.parametrize("direct", [1, 0], ids=["aql", "hip"])
@pytest.mark.parametrize("schedule", [0, 1], ids=["pass-at-block-start", "one-block-ahead"])
def test_narrow_short_pass_is_bitwise_the_wide_one(hydro, schedule, direct, monkeypatch, tuning_build):
    """The in-block short passes of the two-level form run in their NARROW form by default (16 step columns per chunk, offset per
    chunk, BlockArgs::mini_narrow): the same brackets, the same chunk sums, the same order in the reduction -- bitwise the forces of
    the wide form (HC_MINI_NARROW=0), through uniform steps equal to / below / above the IRF spacing and a change of step size."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_DIRECT", str(direct))
    monkeypatch.setenv("HC_SUB_BLOCK", "8")
    N = 6
    case = many_body_case(N, S=180, dt_rirf=0.01, n_exc=41, dt_exc=0.02, seed=415)
    orc = load_into_oracle(case)
    orc.add_waves_none()
    motion = PrescribedMotion(N, rest_positions(case), seed=3)
    times, t = [], 0.0
    for n in range(520):
        t += 0.01 if n < 200 else (0.0071 if n < 360 else 0.0137)
        times.append(t)
    runs = []
    for narrow in ("1", "0"):
        monkeypatch.setenv("HC_MINI_NARROW", narrow)
        gpu = hydro.HydroForces.from_case(case)
        gpu.add_waves_none()
        gpu.set_pass_schedule(schedule)
        gpu.enable_profiling(1)
        f = np.stack([gpu.step(tt, *motion.state(tt)) for tt in times])
        p = gpu.profile()
        assert p["mini_pass_launches"] >= 30, p
        runs.append(f)
        gpu.close()
    assert np.array_equal(runs[0], runs[1])
    for n, tt in enumerate(times):
        assert relerr(runs[0][n], orc.step(tt, *motion.state(tt))) <= TIGHT_TOL, f"step {n}"


@pytest.mark.parametrize("direct", [1, 0], ids=["aql", "hip"])
def test_wide_step_in_one_launch_is_bitwise_the_two_launch_form(hydro, direct, monkeypatch, tuning_build):
    """A block step of a wide system (D >= 1024) is ONE launch by default (wide_step_kernel: column slices of the own-sample part, then
    the workgroup that arrives last at a row tile's counter runs the tile's step kernel; the hand-off across XCDs uses agent-scope atomic
    stores / loads of the partials and no fence); HC_WIDE_FUSED=0 keeps near_split_kernel + finalize_kernel.  The slices are added in
    slice order from memory either way: bitwise the same forces over 400 steps incl. a step-size change, and on the oracle."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_DIRECT", str(direct))
    N = 176  # D = 1056
    case = many_body_case(N, S=48, dt_rirf=0.02, n_exc=17, dt_exc=0.05, seed=1057)
    orc = load_into_oracle(case)
    kw = dict(WAVES, simulation_duration=6.0)
    orc.add_waves_irregular(**kw)
    motion = PrescribedMotion(N, rest_positions(case), seed=5)
    times, t = [], 0.0
    for n in range(400):
        times.append(t)
        t += 0.01 if n < 250 else 0.0073
    import subprocess, sys, json  # the switch is read once per process: the two forms run in processes of their own
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import hydrochrono_amd.hydro as hydro\n"
        "from hydrochrono_amd.mock_chrono import PrescribedMotion\n"
        "from hydrochrono_amd.synthetic import many_body_case, rest_positions\n"
        "case = many_body_case(176, S=48, dt_rirf=0.02, n_exc=17, dt_exc=0.05, seed=1057)\n"
        "gpu = hydro.HydroForces.from_case(case)\n"
        "gpu.add_waves_irregular(**%r)\n"
        "motion = PrescribedMotion(176, rest_positions(case), seed=5)\n"
        "f = np.stack([gpu.step(t, *motion.state(t)) for t in %r])\n"
        "np.save(sys.argv[1], f)\n"
        "print(gpu.profile()['wide_fused_steps'])\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), kw, times)
    runs, counts = [], []
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        # ... and once more on the SHIPPED library, which has the fused form only (it does not read the switch): the A/B reaches the
        # release build's code object too (round-5 review: "the A/B test lives only in the tuning build")
        for fused, flavor in (("1", "tuning"), ("0", "tuning"), ("1", "release")):
            out = os.path.join(d, f"f{fused}_{flavor}.npy")
            r = subprocess.run([sys.executable, "-c", code, out], capture_output=True, text=True, env=dict(os.environ, HC_WIDE_FUSED=fused, HC_DIRECT=str(direct), HYDROCHRONO_AMD_FLAVOR=flavor))
            assert r.returncode == 0, r.stderr[-2000:]
            counts.append(int(r.stdout.strip().splitlines()[-1]))
            runs.append(np.load(out))
    assert counts[0] >= 300 and counts[1] == 0 and counts[2] >= 300, counts
    assert np.array_equal(runs[0], runs[1])
    assert np.array_equal(runs[2], runs[1])
    for n, tt in enumerate(times):
        assert relerr(runs[0][n], orc.step(tt, *motion.state(tt))) <= TIGHT_TOL, f"step {n}"


# ---- the adaptive schedule (hc_set_pass_schedule(ctx, -1), the default) ---------------------------------------------------------
def test_adaptive_schedule_follows_the_callers_gaps_c3_size(hydro):
    """C3 size under the DEFAULT schedule: a caller that steps back to back (hc_step_many: the C ABI's own loop, no interpreter
    between the calls) gets the pass at block start, the same caller with an interpreter's work between its calls (tens of
    microseconds: more than the rule's 4 us) gets it one block ahead, and back again -- 0 -> gaps -> 0 in one run, every step against
    the flat oracle.  The counters show the rule's answer block by block."""
    import bench as B
    import oracle as orc_mod
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    case = many_body_case(64, S=B.S_RIRF, dt_rirf=B.DT, n_exc=B.N_EXC, dt_exc=B.DT, seed=20251031)
    gpu = hydro.HydroForces.from_case(case)
    motion = PrescribedMotion(64, rest_positions(case), seed=20251031)
    kw = dict(B.WAVES, simulation_dt=B.DT, simulation_duration=B.T0 + 12.0)
    gpu.add_waves_irregular(num_bodies=64, **kw)
    orc_mod.set_num_threads(min(64, os.cpu_count() or 1))
    orc = load_into_oracle(case)
    orc.add_waves_irregular(**kw)
    nhist = B.S_RIRF + 5
    t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    gpu.set_history(t_hist, v_hist)
    orc.prefill_history(t_hist, v_hist)
    orc.flat_prepare()
    phases = [("tight", 192), ("gaps", 160), ("tight", 192)]
    n0, worst, answers = 0, 0.0, []
    for kind, count in phases:
        before = gpu.profile()
        times = B.T0 + B.DT * np.arange(n0, n0 + count)
        if kind == "tight":
            forces, _ = gpu.step_many(times, np.stack([motion.packed(t) for t in times]))
        else:
            forces = np.stack([gpu.step(t, *motion.state(t)) for t in times])
        for t, f in zip(times, forces):
            e = relerr(f, orc.flat_step(t, *motion.state(t)))
            worst = max(worst, e)
            assert e <= 1e-10, f"{kind} phase, t = {t}"
        after = gpu.profile()
        answers.append((kind, after["schedule_blocks_ahead"] - before["schedule_blocks_ahead"],
                        after["schedule_blocks_at_start"] - before["schedule_blocks_at_start"], after["ahead_blocks"] - before["ahead_blocks"]))
        n0 += count
    print(f"adaptive schedule at C3 size: worst relative error {worst:.2e}; (phase, answers ahead, answers at start, blocks that started with rows made ahead): {answers}")
    # a decision belongs to the gaps of the block BEFORE it: one block of either phase may still carry the previous phase's answer
    (_, a0, s0, _), (_, a1, s1, r1), (_, a2, s2, _) = answers
    assert s0 >= 5 and a0 == 0, answers      # 64 bodies are not a wide system: the first answer is "at block start"
    assert a1 >= 3 and s1 <= 1 and r1 >= 2, answers
    assert s2 >= 4 and a2 <= 1, answers
    assert gpu.direct_dispatch()[0]


def test_adaptive_schedule_row_shards_decide_alike(hydro, monkeypatch, tuning_build):
    """The rule on a small system (size floor off, threshold raised to 2 ms so that an interpreter's loop counts as back to back and
    a 4 ms sleep as a gap): an unsharded context and three row shards behind hc_step_multi, driven through the same
    0 -> gaps -> 0 pattern.  hc_step_multi measures the gap once for its group, so every shard answers like the others -- and like
    the unsharded context: rows bitwise equal; both against the oracle."""
    import time
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_PASS_AHEAD_MIN_MB", "0")
    monkeypatch.setenv("HC_PASS_AHEAD_GAP_US", "2000")
    N = 6
    case = many_body_case(N, S=200, dt_rirf=0.01, n_exc=41, dt_exc=0.02, seed=505)
    kw = dict(WAVES, simulation_duration=8.0)
    one, grp, orc = hydro.HydroForces.from_case(case), hydro.HydroGroup.from_case(case, 3), load_into_oracle(case)
    for h in (one, grp, orc):
        h.add_waves_irregular(**kw)
    motion = PrescribedMotion(N, rest_positions(case), seed=N)
    slow = lambda n: 330 <= n < 460 or 560 <= n < 600  # noqa: E731
    states = [motion.state(0.01 * n) for n in range(700)]  # (made beforehand: nothing but the two calls sits between the steps)
    got = []
    for n in range(700):
        t = 0.01 * n
        f1 = one.step(t, *states[n])
        if slow(n):
            time.sleep(0.004)
        fg = grp.step(t, *states[n])
        if slow(n):
            time.sleep(0.004)
        assert np.array_equal(f1, fg), f"step {n}: the shards' rows differ from the unsharded context's"
        got.append(f1)
    for n in range(700):  # (the oracle afterwards: its milliseconds per step must not count as caller gaps)
        assert relerr(got[n], orc.step(0.01 * n, *states[n])) <= TIGHT_TOL, f"step {n}"
    p1 = one.profile()
    assert p1["schedule_blocks_ahead"] >= 4 and p1["schedule_blocks_at_start"] >= 8 and p1["ahead_blocks"] >= 3, p1
    for h in grp.shards:
        p = h.profile()
        assert (p["schedule_blocks_ahead"], p["schedule_blocks_at_start"], p["ahead_blocks"]) == \
               (p1["schedule_blocks_ahead"], p1["schedule_blocks_at_start"], p1["ahead_blocks"]), (p, p1)


def test_adaptive_schedule_of_a_wide_system_goes_by_the_size_of_its_slice(hydro):
    """Wide systems (6N >= 1024) under the DEFAULT schedule, back to back through the C ABI's own loop: the rows of 64 of 512 bodies
    (a C4/8 rank, 9.7 GB of K: latency-bound step kernels, threshold 0) answer "one block ahead" from the first block on; the rows of
    128 bodies (a C4/4 rank, 19.4 GB: threshold a tenth of the pass's cost per step, about 10 us) answer "at block start" -- and
    "ahead" as soon as the caller leaves 300 us between its calls (profiles/r05/ahead_probe_shard_sizes.txt).  The rows the two
    shards share agree to rounding."""
    import time
    import bench as B
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    N = B.N_BODIES_C4
    motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
    nhist = B.S_RIRF + 5
    t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    nsteps = 160
    times = B.T0 + B.DT * np.arange(2 * nsteps)
    states = np.stack([motion.packed(t) for t in times])
    rows = {}
    for nb in (64, 128):
        gpu = B.make_shard(N, 0, nb, 0, B.DT, B.T0 + 20.0, 32, t_hist, v_hist)
        f_tight, _ = gpu.step_many(times[:nsteps], states[:nsteps])
        p0 = gpu.profile()
        f_gaps = []
        for k in range(nsteps, 2 * nsteps):
            f_gaps.append(gpu.step(times[k], *motion.state(times[k])))
            b = time.perf_counter()
            while time.perf_counter() - b < 300e-6:
                pass
        p1 = gpu.profile()
        rows[nb] = np.concatenate([f_tight, np.stack(f_gaps)])
        tight = (p0["schedule_blocks_ahead"], p0["schedule_blocks_at_start"])
        gaps = (p1["schedule_blocks_ahead"] - p0["schedule_blocks_ahead"], p1["schedule_blocks_at_start"] - p0["schedule_blocks_at_start"])
        print(f"rows of {nb} of {N} bodies: answers (ahead, at start) back to back {tight}, with 300 us between the calls {gaps}")
        if nb == 64:
            assert tight[0] >= 4 and tight[1] == 0, tight
        else:
            assert tight[1] >= 4 and tight[0] == 0, tight
        assert gaps[0] >= 3 and gaps[1] <= 1, gaps  # (a decision belongs to the gaps of the block before it)
        assert gpu.direct_dispatch()[0]
        gpu.close()
    # (the two schedules group the chunks of K differently: the rows the shards share agree to rounding, not bitwise)
    worst = max(relerr(a, b) for a, b in zip(rows[64], rows[128][:, :6 * 64]))
    print(f"rows shared by the two shards, different schedules: worst relative difference {worst:.2e}")
    assert worst <= 1e-12, worst


def test_wave_model_change_while_the_next_blocks_rows_are_made_ahead(hydro, monkeypatch, tuning_build):
    """Under "one block ahead" the pass of the NEXT block also leaves that block's excitation rows -- computed with the wave model of the
    moment.  A model that changes in the middle of a block must not reach the next block through them (profiles/fuzz_parity.py, seed 40:
    it did), whether the pass in the making is complete by then or its last slice -- the one with the excitation work items -- is still
    to be issued: irregular -> other irregular -> regular -> none -> irregular, each change a few steps into a block, every step and
    its components against the oracle."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    monkeypatch.setenv("HC_PASS_AHEAD_MIN_MB", "0")
    case = many_body_case(4, S=191, dt_rirf=0.015, n_exc=65, dt_exc=0.05, nw=40, seed=1040)
    dt = 0.015
    for lookahead, slices, start_without_waves in ((32, 0, False), (16, 0, False), (32, 12, False), (32, 0, True)):  # (12 slices: the last one goes out late in the block)
        gpu, orc = hydro.HydroForces.from_case(case), load_into_oracle(case)
        kw1 = dict(simulation_dt=dt, simulation_duration=30.0, wave_height=1.5, wave_period=7.0, nfrequencies=32, frequency_min=0.02, frequency_max=0.5, seed=1)
        kw2 = dict(simulation_dt=dt, simulation_duration=30.0, wave_height=2.6, wave_period=9.5, nfrequencies=64, frequency_min=0.02, frequency_max=0.5, seed=5,
                   peak_enhancement_factor=3.3)
        for h in (gpu, orc):
            # (start_without_waves: the first irregular model then arrives in the middle of a block with a pass in the making, and needs
            # a larger partials buffer than that pass is using -- seed 2037 of the fuzz run)
            h.add_waves_none() if start_without_waves else h.add_waves_irregular(**kw1)
        gpu.set_lookahead(lookahead)
        gpu.set_pass_schedule(1, slices)
        motion = PrescribedMotion(4, rest_positions(case), seed=40)
        span = 191 * 0.015
        nh = int(np.ceil(span / dt)) + 4
        t0 = span + 1.0
        th = t0 - dt * np.arange(1, nh + 1)
        vh = np.stack([motion.velocity6(t) for t in th])
        gpu.set_history(th, vh)
        orc.prefill_history(th, vh)
        n = [0]

        def run(steps):
            for _ in range(steps):
                t = t0 + n[0] * dt
                st = motion.state(t)
                assert relerr(gpu.step(t, *st), orc.step(t, *st)) <= TIGHT_TOL, f"lookahead {lookahead}, step {n[0]}"
                for g, o in zip(gpu.components(), orc.components()):
                    sc = max(float(np.max(np.abs(o))), 1e-300)
                    assert float(np.max(np.abs(g - o))) / sc <= 1e-9 or float(np.max(np.abs(o))) == 0.0 and not np.any(g), f"lookahead {lookahead}, components of step {n[0]}"
                n[0] += 1
        run(2 * lookahead + 5)                      # rows are being made ahead; 4 steps into a block
        for h in (gpu, orc):
            h.add_waves_irregular(**kw2)
        run(lookahead + lookahead // 2)             # through the next block start (the adopted rows) and on
        for h in (gpu, orc):
            h.add_waves_regular(0.4, 1.3)
        run(lookahead + 3)
        for h in (gpu, orc):
            h.add_waves_none()
        run(lookahead + 7)
        for h in (gpu, orc):
            h.add_waves_irregular(**kw1)
        run(2 * lookahead)
        p = gpu.profile()
        assert p["ahead_blocks"] >= 4, p
        gpu.close()
        orc.close()


@pytest.mark.parametrize("schedule", [0, 1], ids=["pass_at_block_start", "pass_one_block_ahead"])
def test_pinned_schedule_is_bitwise_repeatable_whatever_the_callers_timing_c3_size(hydro, schedule):
    """The reproducibility contract (reference: a run is repeatable at a fixed thread count, "deterministic
    combine", src/hydro_forces.cpp:641-647).  Under the adaptive DEFAULT the pass schedule follows the caller's gaps, the two schedules
    group the chunks of K differently, and two runs with different caller timing agree to rounding only.  With the schedule PINNED
    (hc_set_pass_schedule(ctx, 0 | 1, 0)) nothing the library computes depends on when the caller calls: the same states give the same
    bits -- here at C3 size, 544 steps, one run back to back through the C ABI's own loop and one with 0 ... 120 us of host work
    (busy waits of changing length) between the calls."""
    import time
    import bench as B
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    case = many_body_case(64, S=B.S_RIRF, dt_rirf=B.DT, n_exc=B.N_EXC, dt_exc=B.DT, seed=20251031)
    motion = PrescribedMotion(64, rest_positions(case), seed=20251031)
    kw = dict(B.WAVES, simulation_dt=B.DT, simulation_duration=B.T0 + 12.0)
    nhist = B.S_RIRF + 5
    t_hist = B.T0 - B.DT * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    n = 544
    times = B.T0 + B.DT * np.arange(n)
    states = np.stack([motion.packed(t) for t in times])
    runs, made_ahead = [], []
    for gaps in (False, True):
        gpu = hydro.HydroForces.from_case(case)
        gpu.add_waves_irregular(num_bodies=64, **kw)
        gpu.set_pass_schedule(schedule)
        gpu.set_history(t_hist, v_hist)
        if not gaps:
            forces, _ = gpu.step_many(times, states)
        else:
            rng = np.random.default_rng(5)
            forces = np.empty((n, gpu.D_local))
            for k, t in enumerate(times):
                forces[k] = gpu.step(t, *motion.state(t))
                stop = time.perf_counter() + float(rng.choice([0.0, 5e-6, 30e-6, 120e-6]))
                while time.perf_counter() < stop:
                    pass
        p = gpu.profile()
        made_ahead.append(int(p["ahead_blocks"]))
        assert (p["schedule_blocks_ahead"] == 0) if schedule == 0 else (p["schedule_blocks_at_start"] == 0), p  # the pin held
        runs.append(forces)
        gpu.close()
    assert np.array_equal(runs[0], runs[1])
    if schedule == 1:
        assert min(made_ahead) >= 10, made_ahead  # blocks did start with rows made one block ahead, in both runs
