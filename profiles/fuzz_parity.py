"""Randomised differential run of the HIP path against the CPU oracle through the C ABI (tuning build: the knobs that force the wide
machinery onto small systems are read from the environment).  Each case draws: bodies 1..9, IRF samples 8..260, IRF spacing, the
caller's stepping pattern (uniform stretches on random step sizes -- equal to, below and above the IRF spacing --, jittered stretches,
repeated times = cache hits), the wave model (none / regular / irregular), the convolution mode (Baseline / TaperedDirect), the look-ahead
depth (0 / 16 / 32), the pass schedule (adaptive / at block start / one block ahead, with the size floor off), the sub-block size of the
two-level form (default / 0 / 4 / 8), direct dispatch or HIP launches, one context or 2-3 row shards behind hc_step_multi, a pre-filled
history or a cold start, gravity -- and, between two steps now and then, a change of depth / schedule / wave model / taper options, the kept
history taken out and injected again, an added-mass product, a step back in time (a rejected step), a reset of the history or of the profiling stride, the
diagnostic entry points (hc_compute_hydrostatics / hc_compute_waves), a switch Baseline <-> TaperedDirect; in a quarter of the
unsharded cases a third of the steps go through hc_step_device on a caller's stream.  Every step's total and its three components against the oracle, each at 1e-9 relative to its own largest
entry; a failure prints the case's seed and stops.   python profiles/fuzz_parity.py [seconds = 300] [first seed = 1]"""
import os
import sys
import time

import numpy as np

# FUZZ_RELEASE=1: the SHIPPED library instead (other compile-time capacities; it reads none of the knobs below, so the schedule "one block
# ahead" needs a system of 256 MB of K and more: a third of the cases are drawn that large)
os.environ["HYDROCHRONO_AMD_FLAVOR"] = os.environ.get("FUZZ_FLAVOR") or ("release" if os.environ.get("FUZZ_RELEASE") else "tuning")  # (FUZZ_FLAVOR: the draw of one mode on the other library)
os.environ["HC_PASS_AHEAD_MIN_MB"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import oracle as orc_mod  # noqa: E402
from cases import load_into_oracle  # noqa: E402
from hydrochrono_amd.hydro import HydroForces, HydroGroup  # noqa: E402
from hydrochrono_amd.mock_chrono import PrescribedMotion  # noqa: E402
from hydrochrono_amd.synthetic import many_body_case, rest_positions  # noqa: E402

orc_mod.set_num_threads(min(32, os.cpu_count() or 1))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
TOL = 1e-9


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-300))


def one_case(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 8, 9]))
    S = int(rng.integers(8, 260))
    if os.environ.get("FUZZ_RELEASE") and rng.random() < 0.33:
        N, S = int(rng.choice([28, 30, 32, 36, 40])), int(rng.integers(1000, 1150))  # 8 (6N)^2 S >= 256 MB
    elif rng.random() < float(os.environ.get("FUZZ_WIDE", "0.02")):  # now and then (FUZZ_WIDE=1: always) a genuinely wide system (6N >= 1024: column slices, fused wide step, two-level form by default)
        N, S = int(rng.choice([171, 172, 176, 180, 192, 200])), int(rng.integers(12, 48) if rng.random() < 0.3 else rng.integers(70, 110))  # (S >= 70: room for look-ahead blocks)
    if os.environ.get("FUZZ_SHARDS") and N < 3:  # (FUZZ_SHARDS=1: every case behind hc_step_multi, 2-4 row shards)
        N = int(rng.choice([3, 4, 5, 8, 9]))
    dt_r = float(rng.choice([0.01, 0.015, 0.02, 0.0125]))
    n_exc = int(rng.choice([21, 33, 65]))
    nw = int(rng.choice([16, 40]))
    case = many_body_case(N, S=S, dt_rirf=dt_r, n_exc=n_exc, dt_exc=float(rng.choice([0.02, 0.05])), nw=nw, seed=1000 + seed)
    if rng.random() < 0.5:
        case["g_sys"] = [float(rng.normal() * 0.3), float(rng.normal() * 0.3), -9.81 + float(rng.normal() * 0.2)]
    lookahead = int(rng.choice([0, 16, 32, 32]))
    sched = int(rng.choice([-1, 0, 1, 1]))
    sub = int(rng.choice([-1, -1, 0, 4, 8]))
    direct = int(rng.random() < 0.7)
    shards = int(rng.choice([1, 1, 1, 2, 3])) if N >= 3 else 1
    if os.environ.get("FUZZ_SHARDS"):
        shards = int(rng.choice([2, 3, 4])) if N >= 4 else 2
    wave = str(rng.choice(["none", "regular", "irregular", "irregular"]))
    mode = int(rng.random() < 0.2)  # 1: TaperedDirect
    cold = rng.random() < 0.3
    os.environ["HC_SUB_BLOCK"] = str(sub)
    os.environ["HC_DIRECT"] = str(direct)
    os.environ["HC_SLOT_STATE"] = str(int(rng.random() < 0.8))   # the step kernel's state behind its argument block, or in the context's buffer
    os.environ["HC_MINI_NARROW"] = str(int(rng.random() < 0.7))  # narrow / wide form of the short passes
    slices = int(rng.choice([0, 0, 1, 3, 7, 12]))
    dev_mix = shards == 1 and rng.random() < 0.25                # some of the steps through hc_step_device on a caller's stream
    desc = (f"seed {seed}: N {N} S {S} dt_rirf {dt_r} lookahead {lookahead} schedule {sched} sub {sub} direct {direct} shards {shards} "
            f"waves {wave} mode {mode} cold {int(cold)} slices {slices} dev {int(dev_mix)} slot {os.environ['HC_SLOT_STATE']} narrow {os.environ['HC_MINI_NARROW']}")
    gpu = HydroGroup.from_case(case, shards) if shards > 1 else HydroForces.from_case(case)
    orc = load_into_oracle(case)
    base_dt = float(rng.choice([dt_r, dt_r, dt_r, 0.7 * dt_r, 1.3 * dt_r, 0.5 * dt_r, 0.01, 0.2 * dt_r, 3.0 * dt_r]))  # (0.2: the ring grows; 3.0: blocks would span the window)
    n_steps = int(rng.integers(150, 420)) if N < 20 else int(rng.integers(80, 160))
    span = S * dt_r
    t0 = 0.0 if cold else span + 1.0 + float(rng.uniform(0, 1))
    dur = t0 + n_steps * 2.2 * max(base_dt, dt_r) + 10.0
    if mode == 1:
        opts = dict(smoothing=int(rng.choice([0, 1])), window_length=5, rirf_end_time=float(rng.uniform(0.5, 1.0) * span),
                    taper_start_percent=float(rng.uniform(0.5, 0.9)), taper_end_percent=1.0, taper_final_amplitude=float(rng.choice([0.0, 0.1])))
        for h in (gpu, orc):
            h.set_convolution_mode(1)
            h.set_tapered_direct_options(**opts)
    def draw_waves(kind):
        if kind == "none":
            for h in (gpu, orc):
                h.add_waves_none()
        elif kind == "regular":
            amp, om = float(rng.uniform(0.1, 1.0)), float(rng.uniform(0.1, 0.045 * nw))  # inside the BEM frequency list (0.05 .. 0.05 nw)
            for h in (gpu, orc):
                h.add_waves_regular(amp, om)
        else:
            kw = dict(simulation_dt=base_dt, simulation_duration=dur, ramp_duration=float(rng.choice([0.0, 2.0])), wave_height=float(rng.uniform(0.5, 3.0)),
                      wave_period=float(rng.uniform(5.0, 11.0)), frequency_min=0.02, frequency_max=0.5, nfrequencies=int(rng.choice([16, 64])),
                      peak_enhancement_factor=float(rng.choice([1.0, 3.3])), seed=int(rng.integers(1, 9)))
            for h in (gpu, orc):
                h.add_waves_irregular(**kw)
    draw_waves(wave)
    gpu.set_lookahead(lookahead)
    gpu.set_pass_schedule(sched, slices)
    gpu.enable_profiling(1 if (seed % 4 == 0 or os.environ.get('FUZZ_TRACE')) else 1000000)  # (the launch counters below: passes are always counted, the per-step launches only when timed)
    motion = PrescribedMotion(N, rest_positions(case), seed=seed)
    log = []  # (t, velocity) of the samples pushed so far, oldest first: what the oracle is rebuilt from after a step back in time
    if not cold:
        nh = int(np.ceil(span / base_dt)) + 4
        th = t0 - base_dt * np.arange(1, nh + 1)
        vh = np.stack([motion.velocity6(t) for t in th])
        gpu.set_history(th, vh)
        orc.prefill_history(th, vh)
        log = [(float(a), b) for a, b in zip(th[::-1], vh[::-1])]
    # the caller's times
    times, t = [], t0
    while len(times) < n_steps:
        kind = rng.random()
        n = int(rng.integers(3, 90))
        if kind < 0.6:
            d = base_dt
            seq = [d] * n
        elif kind < 0.8:
            d = float(rng.choice([dt_r, 0.7 * dt_r, 1.3 * dt_r, 2.0 * dt_r, 0.4 * dt_r]))
            seq = [d] * n
        else:
            seq = list(rng.uniform(0.4 * base_dt, 1.8 * base_dt, n))
        for d in seq:
            times.append(t)
            if rng.random() < 0.03:
                times.append(t)  # the same time again: a cache hit (src/hydro_forces.cpp:742-744)
            t += d
    times = times[:n_steps]
    worst = 0.0
    events = 0
    first = gpu.shards[0] if shards > 1 else gpu
    offset, t_prev, rewinds = 0.0, None, 0
    if dev_mix:
        stream = torch.cuda.Stream()
        d_out = torch.zeros(6 * N, dtype=torch.float64, device="cuda")
    for k in range(len(times)):
        tt = times[k] - offset
        if k > 0 and rng.random() < 0.02 and tt > t_prev:
            # something changes between two force evaluations, at an arbitrary place in a look-ahead block
            ev = int(rng.integers(0, 11))
            events += 1
            if os.environ.get("FUZZ_VERBOSE"):
                print(f"   event {ev} before step {k} (t = {tt!r})", flush=True)
            if ev == 0:
                gpu.set_lookahead(int(rng.choice([0, 16, 32])))
            elif ev == 1:
                gpu.set_pass_schedule(int(rng.choice([-1, 0, 1])))
            elif ev == 2:
                draw_waves(str(rng.choice(["none", "regular", "irregular"])))
            elif ev == 3:  # the kept history taken out and injected again into both (a restart from a checkpoint)
                th_, vh_ = first.get_history()
                gpu.set_history(th_, vh_)
                orc.prefill_history(th_, vh_)
                log = [(float(a), b.copy()) for a, b in zip(th_[::-1], vh_[::-1])]
            elif ev == 9 and shards == 1 and k > 0:
                # the diagnostic entry points in the middle of a block: the hydrostatic and wave terms alone, for the state / time of the
                # step before -- they must reproduce that step's components and leave the look-ahead state alone
                st_p = motion.state(t_prev)
                hs_d, wv_d = gpu.compute_hydrostatics(st_p[0], st_p[1]), gpu.compute_waves(t_prev)
                hs_o, _, wv_o = orc.components()
                for nm, a_, b_ in (("hydrostatics", hs_d, hs_o), ("waves", wv_d, wv_o)):
                    sc = float(np.max(np.abs(b_)))
                    if (sc > 0.0 and float(np.max(np.abs(a_ - b_))) / sc > TOL) or (sc == 0.0 and np.any(a_ != 0.0)):
                        print(f"FAIL {desc}: hc_compute_{nm} before step {k}", flush=True)
                        return False, desc, worst, None
            elif ev == 10:  # Baseline <-> TaperedDirect in the middle of a run
                mode = 1 - mode
                opts = dict(smoothing=int(rng.choice([0, 1])), window_length=5, rirf_end_time=float(rng.uniform(0.5, 1.0) * span),
                            taper_start_percent=float(rng.uniform(0.5, 0.9)), taper_end_percent=1.0, taper_final_amplitude=float(rng.choice([0.0, 0.1])))
                for h in (gpu, orc):
                    h.set_convolution_mode(mode)
                    if mode == 1:
                        h.set_tapered_direct_options(**opts)
            elif ev == 7:  # the history thrown away on both sides: a cold start in the middle of a run
                gpu.reset_history()
                orc.prefill_history(np.zeros(0), np.zeros((0, 6 * N)))
                log = []
            elif ev == 8:
                gpu.enable_profiling(int(rng.choice([1, 3, 1000000])))
            elif ev == 6 and k > 5 and len(log) > 8:
                # a rejected step: the caller comes back at an EARLIER time (at most a few samples back: what the retired-sample slack
                # of the ring covers exactly); the library drops the newer samples itself, the oracle is rebuilt from the log
                tt = t_prev - float(rng.uniform(0.3, 2.6)) * max(t_prev - log[-2][0], 1e-4) if log[-1][0] == t_prev else tt
                if tt < t_prev:
                    offset = times[k] - tt
                    while log and log[-1][0] >= tt:
                        log.pop()
                    keep = log  # (everything pushed so far: the oracle prunes by itself)
                    orc.prefill_history(np.array([a for a, _ in reversed(keep)]), np.stack([b for _, b in reversed(keep)]))
                    rewinds += 1
            elif ev == 4 and mode == 1:
                opts = dict(smoothing=int(rng.choice([0, 1])), window_length=5, rirf_end_time=float(rng.uniform(0.5, 1.0) * span),
                            taper_start_percent=float(rng.uniform(0.5, 0.9)), taper_end_percent=1.0, taper_final_amplitude=float(rng.choice([0.0, 0.1])))
                for h in (gpu, orc):
                    h.set_tapered_direct_options(**opts)
            else:  # Chrono's added-mass product between two steps (its own queue lane)
                wv = rng.normal(size=6 * N + 3)
                rg, ro = gpu.added_mass_mv(np.ones(6 * N + 3), wv, 0.7), orc.added_mass_mv(np.ones(6 * N + 3), wv, 0.7)
                if not relerr(rg, ro) <= 1e-12:
                    print(f"FAIL {desc}: added-mass product before step {k}: {relerr(rg, ro):.3e}", flush=True)
                    return False, desc, worst, None
        st = motion.state(tt)
        if dev_mix and rng.random() < 0.3:
            d_st = torch.tensor(motion.packed(tt), device="cuda")
            torch.cuda.synchronize()
            gpu.step_device(tt, d_st.data_ptr(), d_out.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            fg = d_out.cpu().numpy()
        else:
            fg = gpu.step(tt, *st)
        fo = orc.step(tt, *st)
        if t_prev is None or tt != t_prev:
            log.append((tt, motion.velocity6(tt)))
            log = log[-6000:]
        t_prev = tt
        e = relerr(fg, fo)
        if e <= TOL:  # the three components, each relative to ITS OWN largest entry (a radiation error must not hide behind the hydrostatics)
            for a, b in zip(gpu.components(), orc.components()):
                sc = float(np.max(np.abs(b)))
                if sc > 0.0:
                    e = max(e, float(np.max(np.abs(a - b))) / sc)
                elif np.any(a != 0.0):
                    e = 1.0
        worst = max(worst, e)
        if os.environ.get("FUZZ_TRACE") and k >= int(os.environ["FUZZ_TRACE"]):
            pq = first.profile()
            print(f"   step {k} t {tt!r} err {e:.2e}; passes {pq['block_kernel_launches']} plain {pq['conv_kernel_launches']} "
                  f"adopted {pq['ahead_blocks']} slices {pq['ahead_pass_slices']} answers {pq['schedule_blocks_ahead']}/{pq['schedule_blocks_at_start']} "
                  f"short {pq['mini_pass_launches']} scatters {pq['scatter_kernel_launches']}", flush=True)
        if not e <= TOL:
            print(f"FAIL {desc}: step {k} t {tt!r}: relative error {e:.3e}", flush=True)
            if os.environ.get("FUZZ_VERBOSE"):
                for name, a, b in zip(("hs", "rad", "waves"), gpu.components(), orc.components()):
                    print(f"   {name}: max |gpu - oracle| {np.max(np.abs(a - b)):.3e}, max |oracle| {np.max(np.abs(b)):.3e}")
                print(f"   total: max |gpu - oracle| {np.max(np.abs(fg - fo)):.3e}, max |oracle| {np.max(np.abs(fo)):.3e}; history kept: gpu {len(first.get_history()[0])}, oracle {orc.history_size()}")
            return False, desc, worst, None
    prof = (gpu.shards[0] if shards > 1 else gpu).profile()
    if prof["history_rewinds"] != rewinds:
        print(f"FAIL {desc}: {rewinds} steps back in time, the library counted {prof['history_rewinds']}", flush=True)
        return False, desc, worst, None
    gpu.close()
    orc.close()
    return True, desc, worst, np.array([prof["block_kernel_launches"], prof["mini_pass_launches"], prof["ahead_blocks"], prof["scatter_kernel_launches"],
                                        prof["conv_kernel_launches"], prof["direct_dispatches"], prof["hip_launches"]], dtype=np.int64)


t_end = time.time() + budget
n_ok, worst_all, passes = 0, 0.0, np.zeros(7, dtype=np.int64)
while time.time() < t_end:
    if os.environ.get("FUZZ_PRINT_SEEDS"):  # (a case that takes the process down -- a GPU memory fault -- is the last one named)
        print(f"case {seed}", flush=True)
    ok, desc, worst, npass = one_case(seed)
    if not ok:
        sys.exit(1)
    n_ok += 1
    worst_all = max(worst_all, worst)
    passes += npass
    if n_ok % (200 if (float(os.environ.get('FUZZ_WIDE', '0.02')) < 0.5 and not os.environ.get('FUZZ_RELEASE')) else 10) == 0:
        print(f"{n_ok} cases ok (last: {desc}; worst so far {worst_all:.2e})", flush=True)
    seed += 1
from hydrochrono_amd import capi  # noqa: E402
print(f"library: {capi.load().hc_version().decode()}, flavour {os.environ['HYDROCHRONO_AMD_FLAVOR']}")
print(f"fuzz ok: {n_ok} cases, seeds up to {seed - 1}, worst relative error {worst_all:.2e}; launches over all cases (first shard of a group): passes {passes[0]}, "
      f"short passes {passes[1]} and scatters {passes[3]} (counted in every fourth case only), blocks that started with rows made ahead {passes[2]}, plain per-step kernels {passes[4]}; "
      f"AQL dispatches {passes[5]}, HIP launches {passes[6]}")
