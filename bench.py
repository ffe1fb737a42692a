#!/usr/bin/env python3
"""Headline benchmark of the hydro-force path: all-body force evaluations per second.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

N = 1 (default) -- configuration C3 (BASELINE.json configs[2]): synthetic 64-body array, 1024 radiation-IRF samples,
irregular JONSWAP sea state with 512 wave components, prescribed body motion with dt = dt_rirf = 0.01 s, steady state
(velocity history pre-filled over the whole 10.23 s IRF window).  A "step" = one SYNCHRONOUS hc_step: host pointers to
the body state in, all 6N hydrodynamic forces (hydrostatic - radiation + waves) out on the host -- what one Chrono
update costs through ComponentFunc::GetVal (src/hydro_forces.cpp:79-85,727-767): the next state depends on these forces,
so nothing is pipelined across steps.  value = K / wall time of K consecutive calls, the look-ahead passes included (the
timed region is phase-aligned so that it always contains a pass, however small K is); the median call is reported next to it.  Secondary figures in the same line: `device_pipelined` (hc_step_device with
states resident in HBM, enqueued ahead -- an upper bound no Chrono loop can use), `plain_per_step_mode` (look-ahead off: K streamed
from HBM every step), `steady_state` (256 more synchronous steps), `chrono_like_loop` (hc_step with 100 / 30 us of host work between
calls under the library's default pass schedule -- adaptive, hc_set_pass_schedule(ctx, -1, 0) -- and under each pinned schedule),
`c4_rank_share` (what ONE rank of C4/8 does: back to back and with 300 us of host work between calls, default and pinned schedules),
`c4_one_gpu` (the whole C4 array on this GPU: the one-GPU figure of the workload the N > 1 lines shard), `c2_two_body` and
`c5_one_body_2048` (BASELINE.json's other single-GPU configs, synthetic stand-ins, each beside the CPU oracle at one thread and at its
best thread count) and `added_mass_mv` (hc_added_mass_mv per call at 6 ... 3072 coordinates beside the host product, and the size
below which the C++ binding keeps the product on the host).  MEASURED.md indexes the committed figures of the round.

N > 1 (default: --scaling strong --bodies 512) -- configuration C4: ONE coupled 512-body array (K = 77.3 GB FP64,
generated in HBM by hc_synth_fill) row-sharded over the ranks (hydrochrono_amd.parallel.body_shard).  Every rank holds the
full 6N force vector ON THE HOST before the next step starts (where a Chrono integrator needs it): each rank runs the
synchronous hc_step path on its shard and collects all shards' rows straight from the shared-memory result buffers the GPUs
write (--exchange host, the default: SURVEY 8e "outputs -> host gather"); --exchange rccl all-gathers the rows on the device
over RCCL instead (hc_step_device + all_gather_into_tensor + stream synchronise); the mode not chosen is reported as a
secondary.  value = K / max-over-ranks time; `per_rank` lists every rank's own loop time and kernel split (pass / short passes /
scatter / step kernels), so that one run on a multi-GPU node says which rank was slow and where.  `--scaling weak` runs independent
64-body farms instead (replicas, no data-path collective).  `--scaling strong` also runs on one GPU (77 GB fits in 288 GB).
`--stub-context` rehearses this whole flow on CPU over gloo with tests/stub_context.StubShard (tests only: no GPU, no physics).

N > 1 in ONE process (`python bench.py --gpus N` without a launcher, or `--single-process` under one): the same coupled array
row-sharded over N contexts of this process, one per visible GPU (all on GPU 0 when the box has fewer: a functional mode), evaluated
with ONE hc_step_multi per step -- the C-ABI multi-GPU path of a Chrono host (no torch.distributed, no collective; host gather).
The timed region is aligned to a look-ahead block boundary like the N = 1 line; the line carries `roofline` (each shard's pass),
`per_shard` kernel splits, `passes_in_timed_region`, an `exchange_check` (the rows of hc_step_multi against each shard's own hc_step
on a second set of contexts), `cpu_baseline: null` with the reason -- and `rccl_ranks`: BEFORE this process touches a GPU it runs the
launcher form as a child (`python -m torch.distributed.run --nproc-per-node N bench.py --exchange rccl --no-secondary`: one rank per
GPU, RCCL all-gather of the force rows every step) and attaches that line, so that the one command also says whether RCCL saw N ranks.
Under torch.distributed.run rank 0 also reports the one-process mode as the secondary `single_process_c_abi` (the other ranks wait).

The JSON line also carries
  roofline      HBM roofline of the dominant kernel (the look-ahead pass): bytes the launch has to move ONCE / mean
                HIP-event duration / 8 TB/s -- a fraction; the reuse over the 32 (or 16) steps one launch serves is
                reported separately; `whole_step`: the bytes of ONE steady-state step (pass / 32 + scatter + own samples)
                against the memory at the mean and at the median step -- the step itself is latency-bound, and says so
  init          the init half of the path stage by stage, each beside its bound and beside the oracle's init: BEMIO ingest of
                a C3-size file (HDF5 read / PCIe copy / re-layout kernel), irregular-wave set-up (resample, spectrum, free-
                surface synthesis: direct sum and rocFFT; C3 and the sphere-irregular size), TaperedDirect build, generator
  cpu_baseline  the CPU oracle (reference-faithful OpenMP restatement, oracle/) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N=1 only)
  parity        every timed step compared with the CPU oracle (flat-array variant; the faithful one on its sample)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# rocprofv3 --kernel-trace --stats of this very command (python bench.py --gpus 1 --steps 20 --warmup 5 --no-c4-share --no-c4-one-gpu
# --no-small-configs --no-init, so that the pass kernel's row holds C3 launches only): roofline.frac = algorithmic_bytes_per_launch / its AverageNs / 8 TB/s
ROOFLINE_PROFILE = ("profiles/r06/c3_driver_cmd_kernel_stats_by_grid.csv, row hc::conv_block_kernel<6, 4, 2, 1> on 256 workgroups "
                    "(MEASURED.md: recomputing roofline.frac); PMC: profiles/r06/pmc_traffic.json, pmc_step_traffic.json")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
HBM_STREAM_CEILING_GBS = 6300.0  # what a pure dwordx4 load stream reaches on this chip (profiles/r05/mfma_vmem_probe.txt: 6.30-6.38 TB/s; the guide: 6.29)
FP64_MFMA_PEAK_TF = 78.6

WAVES = dict(simulation_dt=0.01, simulation_duration=60.0, ramp_duration=0.0, wave_height=2.0, wave_period=8.0,
             frequency_min=0.02, frequency_max=0.5, nfrequencies=512, peak_enhancement_factor=3.3, seed=1)
N_BODIES, N_BODIES_C4, S_RIRF, N_EXC, DT = 64, 512, 1024, 1024, 0.01
T0 = 20.0  # start of the timed window (history covers [T0 - 10.29, T0))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--bodies", type=int, default=None, help="default: 64 (C3) for independent farms, 512 (C4) for --scaling strong")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong (one coupled array sharded over the ranks) when N > 1, else the single-GPU C3 case")
    ap.add_argument("--profile-stride", type=int, default=17, help="HIP-event sampling stride for the per-step launches (co-prime "
                    "with the 16-step look-ahead period); every look-ahead pass, the roofline kernel, is timed regardless")
    ap.add_argument("--lookahead", type=int, default=32, help="steps per look-ahead block (16 or 32; 64: the experimental depth of profiles/r05); 0: plain per-step evaluation (K streamed every step)")
    ap.add_argument("--step-dt", type=float, default=DT, help="caller's step size (default = the IRF grid spacing, the common "
                    "case; e.g. 0.007 makes every IRF sample a true interpolation, SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the device_pipelined / plain_per_step_mode measurements")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline time budget")
    ap.add_argument("--steady-steps", type=int, default=256, help="synchronous steps of the steady_state block (median + mean, SURVEY 8d: >= 200)")
    ap.add_argument("--single-process", action="store_true", help="N > 1: all shards in THIS process through hc_step_multi (no launcher needed)")
    ap.add_argument("--python-loop", action="store_true", help="N = 1: time hc_step calls issued one by one from the Python interpreter (rounds 1-4a) "
                                                             "instead of the library's own loop hc_step_many")
    ap.add_argument("--no-c4-share", action="store_true", help="skip the c4_rank_share secondary (one C4/8 shard on this GPU)")
    ap.add_argument("--stub-context", action="store_true", help="TEST ONLY (tests/test_parallel_gloo.py): run the N > 1 control flow on CPU over gloo with "
                    "tests/stub_context.StubShard in place of the GPU contexts -- no physics, no timings worth reading")
    ap.add_argument("--no-small-configs", action="store_true", help="skip the c2_two_body / c5_one_body_2048 / added_mass_mv secondaries (each beside its CPU figure)")
    ap.add_argument("--no-c4-one-gpu", action="store_true", help="skip the c4_one_gpu secondary (the whole 512-body array, 77 GB of K, on this GPU)")
    ap.add_argument("--no-pin", action="store_true", help="do not bind the stepping thread to the CPUs local to its GPU (hc_bind_thread_to_device)")
    ap.add_argument("--no-init", action="store_true", help="skip the `init` block (BEMIO ingest of a C3-size file, irregular-wave set-up, TaperedDirect build: each stage timed beside its bound)")
    ap.add_argument("--exchange", choices=["host", "rccl"], default="host",
                    help="N > 1 under a launcher: how every rank gets all force rows each step.  host (default): hc_step on every rank (direct "
                         "dispatch, results on the host) and a host gather through shared-memory result buffers (hc_set_result_buffer); "
                         "rccl: hc_step_device + RCCL all-gather of the rows on the device + stream synchronise.  The other one is run as a secondary")
    return ap.parse_args()


def cpu_baseline(case, motion, t_hist, v_hist, budget_s, step_dt, duration):
    """Times the CPU oracle on this box's host cores on a bounded sample of the same workload.

    Two variants: the reference-faithful restatement (OpenMP over IRF steps, per-element
    accessor, nested-vector history -- `value`, thread count chosen by a short sweep because the reference's
    `schedule(static)` over 1024 steps does not scale to every core count) and an optimised flat-array CPU variant
    (`optimized_port`) so that the GPU speed-up is not flattered by reference overheads.
    Returns (dict, [forces of the first steps from T0 on, faithful oracle])."""
    import oracle as orc_mod
    from cases import load_into_oracle
    cores = os.cpu_count() or 1
    orc = load_into_oracle(case)
    orc.add_waves_irregular(**dict(WAVES, simulation_dt=step_dt, simulation_duration=duration))
    orc.prefill_history(t_hist, v_hist)
    k = [0]
    forces = []

    def timed(fn, nsteps, keep):
        d = []
        for _ in range(nsteps):
            t = T0 + k[0] * step_dt
            st = motion.state(t)
            a = time.perf_counter()
            f = fn(t, *st)
            d.append(time.perf_counter() - a)
            if keep:
                forces.append(f)
            k[0] += 1
        return d

    t_begin = time.perf_counter()
    sweep = {}
    for th in sorted({cores, max(1, cores // 2), min(cores, 64), min(cores, 16)}, reverse=True):
        orc_mod.set_num_threads(th)
        sweep[th] = float(np.median(timed(orc.step, 3, True)))
    best = min(sweep, key=sweep.get)
    orc_mod.set_num_threads(best)
    n_more = int(max(3, min(40, (0.5 * budget_s - (time.perf_counter() - t_begin)) / sweep[best])))
    med = float(np.median(timed(orc.step, n_more, True)))
    orc_mod.set_num_threads(1)  # the reference-faithful form on ONE thread as well
    single_ms = float(np.median(timed(orc.step, 2, True))) * 1e3
    flat_sweep = {}
    for th in sorted({cores, min(cores, 64), min(cores, 16)}, reverse=True):
        orc_mod.set_num_threads(th)
        orc.flat_prepare()  # re-laid out (first touch) with this thread count
        flat_sweep[th] = float(np.median(timed(orc.flat_step, 8, False)[2:]))
    best_flat = min(flat_sweep, key=flat_sweep.get)
    med_flat = flat_sweep[best_flat]
    info = {"value": 1.0 / med, "unit": "evals/s", "cores": best, "kind": "port", "ms_per_step": med * 1e3,
            "sample": f"{n_more} steady-state steps (median), faithful oracle -O2 -fopenmp, {best} threads (best of sweep; {cores} logical cores)",
            "threads_sweep_ms": {str(th): v * 1e3 for th, v in sweep.items()},
            "single_thread_ms": single_ms,
            "optimized_port": {"value": 1.0 / med_flat, "unit": "evals/s", "cores": best_flat, "ms_per_step": med_flat * 1e3,
                               "threads_sweep_ms": {str(th): v * 1e3 for th, v in flat_sweep.items()},
                               "sample": "6 steps per thread count (median), flat-array OpenMP-over-rows CPU variant of the same math"}}
    return info, forces, best_flat


def oracle_all_steps(case, motion, t_hist, v_hist, step_dt, duration, nsteps, threads):
    """Forces of steps T0 + k*dt, k < nsteps, from the flat-array variant of the CPU oracle (checked against the faithful
    restatement in tests/test_oracle_flat.py and, below, on the faithful oracle's own sample)."""
    import oracle as orc_mod
    from cases import load_into_oracle
    orc_mod.set_num_threads(threads)
    orc = load_into_oracle(case)
    orc.add_waves_irregular(**dict(WAVES, simulation_dt=step_dt, simulation_duration=duration))
    orc.prefill_history(t_hist, v_hist)
    orc.flat_prepare()
    out = np.empty((nsteps, 6 * case["N"]))
    for k in range(nsteps):
        t = T0 + k * step_dt
        out[k] = orc.flat_step(t, *motion.state(t))
    return out


def max_rel_err(a, b):
    """Largest per-step error, each relative to the largest force component of that step (the tests' vector-relative norm)."""
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    return float(np.max(np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), 1e-300)))


def make_shard(N, b0, b1, device, sdt, duration, lookahead, t_hist, v_hist, cls=None):
    """One row shard of the synthetic coupled N-body array (K generated in HBM), waves attached, history pre-filled."""
    if cls is None:
        from hydrochrono_amd.hydro import HydroForces as cls  # noqa: N813
    gpu = cls(N, device=device, body_range=(b0, b1))
    gpu.synth_fill(20251031, S_RIRF, DT, N_EXC, DT)
    gpu.finalize()
    gpu.add_waves_irregular(**dict(WAVES, num_bodies=N, simulation_dt=sdt, simulation_duration=duration))
    gpu.set_lookahead(lookahead)
    gpu.set_history(t_hist, v_hist)
    return gpu


def dispatch_info(shards):
    """How the kernels of the step path reach the GPU (hc_direct_dispatch_active / hc_dispatch_mode_reason), per context."""
    modes = [h.direct_dispatch() for h in shards]
    allq = all(m[0] for m in modes)
    return {"dispatch_mode": "direct AQL packets (library-owned HSA queue)" if allq else ("HIP launches" if not any(m[0] for m in modes) else "mixed"),
            "dispatch_mode_reason": sorted({m[1] for m in modes})}


def kterm_mean(units):
    """Term slots a block step reads on average (one per earlier step of the block whose sample reaches it: (units - 1) / 2)."""
    return (units - 1) / 2.0


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_rccl_ranks(args, G, ndev):
    """`python bench.py --gpus G` without a launcher is ONE process (hc_step_multi, host gather): no collective runs in it.  So that the
    same command also answers "did RCCL see G ranks", this process -- BEFORE it makes a single GPU call -- starts the launcher form as a
    child (`python -m torch.distributed.run --nproc-per-node G bench.py --exchange rccl --no-secondary`, one rank per GPU, all-gather
    of the force rows over RCCL every step), waits for it and attaches its line.  With fewer GPUs than ranks the child runs in the
    functional mode (HC_BENCH_SHARE_GPU=1: all ranks on device 0, rows over gloo)."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    if ndev < G:
        env["HC_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={G}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__), "--gpus", str(G), "--steps", str(args.steps), "--warmup", str(args.warmup), "--exchange", "rccl", "--no-secondary",
           "--lookahead", str(args.lookahead), "--step-dt", str(args.step_dt)] + (["--bodies", str(args.bodies)] if args.bodies else []) + (
               ["--stub-context"] if args.stub_context else [])
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=float(os.environ.get("HC_BENCH_CHILD_TIMEOUT_S", "1500")))
    except Exception as e:  # noqa: BLE001
        return {"error": f"the launcher child did not finish: {e}", "command": " ".join(cmd[1:])}
    line = None
    for ln in reversed(r.stdout.strip().splitlines()):
        if ln.startswith("{"):
            try:
                line = json.loads(ln)
                break
            except Exception:  # noqa: BLE001
                continue
    if line is None:
        return {"error": "the launcher child printed no JSON line", "returncode": r.returncode, "stderr_tail": r.stderr[-600:], "command": " ".join(cmd[1:])}
    keep = ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "median_ms_per_step", "passes_in_timed_region", "exchange", "exchange_check",
            "collective_backend", "collective_world_size", "rccl_world_size", "per_rank_ms_per_step", "pass_schedule_pinned_for_all_ranks", "note", "stub_context")
    out = {k: line[k] for k in keep if k in line}
    out["per_rank"] = [{k: r_[k] for k in ("rank", "device", "rows", "ms_per_step_own_loop", "median_ms_per_step", "passes_in_timed_region") if k in r_}
                       for r_ in line.get("per_rank", [])]
    out["wall_seconds"] = time.perf_counter() - t0
    out["command"] = "python -m torch.distributed.run --nnodes=1 --nproc-per-node=%d ... bench.py --gpus %d --exchange rccl --no-secondary" % (G, G)
    out["functional_mode_one_device"] = ndev < G
    return out


def block_alignment(lookahead, warm, steps):
    """Untimed steps in front of the warm-up that put a look-ahead block boundary into the middle of the timed region (one look-ahead
    pass serves a block and is paid by the first step of the next: a short region could fall between two passes)."""
    if lookahead <= 0:
        return 0
    Lb = 16 if lookahead <= 16 else 32
    first = warm + steps // 2
    return ((first + Lb - 1) // Lb) * Lb - first


def shard_view(p, nst, rows, b0, b1, device):
    """One shard's kernel split over the timed region (hc_profile_stats deltas): the `per_rank` schema of the launcher lines."""
    pass_s = p["block_kernel_seconds"] / max(1, p["block_kernel_launches"])
    return {"bodies": [b0, b1], "device": device, "rows": rows,
            "kernel_us_per_step": {"pass": p["block_kernel_seconds"] / nst * 1e6, "short_passes": p["mini_pass_seconds"] / nst * 1e6,
                                   "scatter": p["scatter_kernel_seconds"] / nst * 1e6, "step_kernels": p["step_kernel_seconds"] / nst * 1e6,
                                   "note": "the per-step launches are sampled (every 17th step), every pass is timed"},
            "passes_in_timed_region": int(p["block_kernel_launches"]), "blocks_with_rows_made_ahead": int(p["ahead_blocks"]),
            "pass_us_per_launch": pass_s * 1e6, "pass_bytes_once": p["block_kernel_bytes_once"],
            "pass_frac_of_hbm_peak": (p["block_kernel_bytes_once"] / pass_s / 1e9 / HBM_PEAK_GBS) if pass_s > 0 else None,
            "doorbell_offset_us_mean": (1e6 * p["multi_doorbell_offset_sum"] / p["multi_calls"]) if p["multi_calls"] else None,
            "aql_dispatches": int(p["direct_dispatches"]), "hip_launches": int(p["hip_launches"])}


def run_group_sync(N, G, devices, sdt, lookahead, warm, steps, motion=None, check_steps=0, stub_cls=None):
    """ONE process, G row-shard contexts (devices[g]), `steps` synchronous hc_step_multi calls after an alignment stretch and `warm`
    untimed ones (host state in, the gathered 6N forces out on the host).  check_steps > 0: afterwards a SECOND set of shard contexts
    (same data, same history) takes the first check_steps states one shard at a time through its own hc_step, and the rows must be
    those hc_step_multi gathered (`exchange_check`).  Returns (dict, forces[steps][6N])."""
    from hydrochrono_amd import capi
    from hydrochrono_amd.hydro import HydroGroup
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.parallel_split import body_shard
    motion = motion or PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
    nhist = int(np.ceil(S_RIRF * DT / sdt)) + 5
    t_hist = T0 - sdt * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    align = block_alignment(lookahead, warm, steps)
    pre = align + warm
    n_all = pre + steps
    duration = max(WAVES["simulation_duration"], T0 + (n_all + 8) * sdt + 5.0)
    bounds = [body_shard(N, G, g) for g in range(G)]
    shards = [make_shard(N, *bounds[g], devices[g], sdt, duration, lookahead, t_hist, v_hist, cls=stub_cls) for g in range(G)]
    times = [T0 + k * sdt for k in range(n_all)]
    states = np.ascontiguousarray(np.stack([motion.packed(t) for t in times]))
    forces = np.zeros((n_all, 6 * N))
    n3 = 3 * N
    if stub_cls is None:
        grp = HydroGroup(shards)
        ctxs = grp._ctxs
        step = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(("hc_step_multi", capi.load()))
    else:
        grp, ctxs = None, None  # (CPU rehearsal, tests only: the stub shards one after the other stand in for hc_step_multi)

        def step(_c, _g, t, p_pos, p_rpy, p_lin, p_ang, out_addr):
            for h in shards:
                h.begin_raw(h.ctx, t, p_pos, p_rpy, p_lin, p_ang)
                h.end_raw(h.ctx, out_addr + 8 * 6 * h.b0)
            return 0
    sp = [states.ctypes.data + k * states.strides[0] for k in range(n_all)]
    fp = [forces.ctypes.data + k * forces.strides[0] for k in range(n_all)]
    per = np.zeros(n_all)
    pc = time.perf_counter

    def run(k0, k1):
        for k in range(k0, k1):
            a = pc()
            rc = step(ctxs, G, times[k], sp[k], sp[k] + 8 * n3, sp[k] + 16 * n3, sp[k] + 24 * n3, fp[k])
            per[k] = pc() - a
            if rc:
                raise RuntimeError(capi.load().hc_last_error(shards[0].ctx).decode())

    for h in shards:
        h.enable_profiling(17)
    run(0, pre)
    for h in shards:
        h.reset_profile()
    t0 = pc()
    run(pre, n_all)
    elapsed = pc() - t0
    profs = [h.profile() for h in shards]
    views = [shard_view(profs[g], steps, shards[g].D_local, *bounds[g], devices[g]) for g in range(G)]
    fr = [v["pass_frac_of_hbm_peak"] for v in views if v["pass_frac_of_hbm_peak"]]
    info = {"evals_per_s": steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "median_ms_per_step": float(np.median(per[pre:])) * 1e3,
            "p90_ms_per_step": float(np.percentile(per[pre:], 90)) * 1e3, "max_ms_per_step": float(per[pre:].max()) * 1e3, "steps": steps, "warmup": warm,
            "alignment_steps": align, "contexts": G, "devices": list(devices), "bodies": N, "lookahead": shards[0].schedule()["lookahead"],
            "passes_in_timed_region": max(v["passes_in_timed_region"] for v in views),
            "pass_us_max_over_shards": max(v["pass_us_per_launch"] for v in views),
            "roofline": {"bound": "hbm", "kernel": "hc::conv_block_kernel (the look-ahead pass of each shard)", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac_max_over_shards": max(fr) if fr else None, "frac_min_over_shards": min(fr) if fr else None,
                         "achieved_max": max(fr) * HBM_PEAK_GBS if fr else None, "achieved_min": min(fr) * HBM_PEAK_GBS if fr else None,
                         "units_per_launch": shards[0].schedule()["lookahead"],
                         "note": "per shard: bytes a launch of its pass moves once / mean duration of its launches in the timed region; a pass made one block "
                                 "ahead goes out in slices (each launch a slice on the CU-masked pass lane): judge those shards by ms_per_step"},
            "per_shard": views,
            "aql_dispatches": int(sum(p["direct_dispatches"] for p in profs)), "hip_launches": int(sum(p["hip_launches"] for p in profs))}
    info.update(dispatch_info(shards))
    if grp is not None:
        grp.close()
    if check_steps > 0:
        # the exchange of this mode is the host gather inside hc_step_multi: its rows against each shard's own hc_step on the same states
        n_chk = min(check_steps, n_all)
        worst, bitwise = 0.0, True
        for g in range(G):
            h = make_shard(N, *bounds[g], devices[g], sdt, duration, lookahead, t_hist, v_hist, cls=stub_cls)
            own = np.stack([h.step(times[k], *motion.state(times[k])) for k in range(n_chk)])
            got = forces[:n_chk, 6 * bounds[g][0]:6 * bounds[g][1]]
            bitwise = bitwise and bool(np.array_equal(own, got))
            worst = max(worst, max_rel_err(got, own))
            h.close()
        info["exchange_check"] = {"rows_of_hc_step_multi_equal_each_shards_own_hc_step": bool(worst <= 1e-12), "bitwise": bitwise, "max_rel_diff": worst,
                                  "steps_checked": n_chk, "finite": bool(np.isfinite(forces).all()),
                                  "note": "a second set of shard contexts, each stepped alone through hc_step from the same history; under the adaptive pass "
                                          "schedule two runs agree to rounding (bitwise with the schedule pinned, tests/test_gpu_multi.py)"}
    return info, forces[pre:]


def c4_rank_share(sdt, lookahead):
    """What ONE rank of C4/8 does, on this GPU: the rows of bodies [0, 64) of the coupled 512-body array (K slice 9.66 GB),
    synchronous hc_step.  A driver-run figure for multi-GPU readiness while no 8-GPU node is available.  ms_per_step is measured
    under the library's DEFAULT pass schedule (adaptive; for a wide system with at most 12 GB of K in the context that is "one block ahead" at every caller gap) after a
    run-in of three blocks -- the first block under a new schedule runs two passes, its own and the next one's -- and each pinned
    schedule follows, back to back and with 300 us of host work between the calls."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    N, warm, steps = N_BODIES_C4, 104, 256
    n_gap, n_in, gap, n_prof = 96, 72, 300e-6, 96  # the gap loops and the profiled stretch below
    motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
    nhist = int(np.ceil(S_RIRF * DT / sdt)) + 5
    t_hist = T0 - sdt * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    n_all = warm + steps + n_prof + 3 * (n_gap + n_in) + 2 * (n_in + steps)
    gpu = make_shard(N, 0, N // 8, 0, sdt, T0 + (n_all + 8) * sdt + 5.0, lookahead, t_hist, v_hist)
    times = [T0 + k * sdt for k in range(n_all)]
    states = [motion.state(t) for t in times]
    pos = [0]

    def back_to_back(skip, n):
        lat = np.zeros(n)
        for i in range(skip + n):
            k = pos[0] + i
            a = time.perf_counter()
            gpu.step(times[k], *states[k])
            if i >= skip:
                lat[i - skip] = time.perf_counter() - a
        pos[0] += skip + n
        return lat
    per = back_to_back(warm, steps)
    p_def = gpu.profile()
    # the per-kernel figures come from a stretch of their own with the library's kernel timing on (it costs the loop a few us per step)
    gpu.enable_profiling(1)
    gpu.reset_profile()
    back_to_back(0, n_prof)
    p = gpu.profile()
    gpu.enable_profiling(False)
    pass_s = p["block_kernel_seconds"] / max(1, p["block_kernel_launches"])
    out = {"workload": f"rows of bodies [0, {N // 8}) of the coupled {N}-body array (D_local = {gpu.D_local}, D = {6 * N}, K slice "
                       f"{p['conv_kernel_bytes'] / 1e9:.2f} GB), synchronous hc_step through the Python wrapper, {steps} steps after {warm}",
           "pass_schedule": "adaptive (the library's default): " + ("one block ahead, on the pass lane beside the steps" if p["ahead_blocks"] > 0 else "at block start"),
           "schedule_answers": {"one_block_ahead": int(p_def["schedule_blocks_ahead"]), "at_block_start": int(p_def["schedule_blocks_at_start"])},
           "ms_per_step": float(per.mean()) * 1e3, "median_ms_per_step": float(np.median(per)) * 1e3, "max_ms_per_step": float(per.max()) * 1e3,
           "pass_us": pass_s * 1e6, "pass_launches": int(p["block_kernel_launches"]),
           "pass_frac_of_hbm_peak": p["block_kernel_bytes_once"] / pass_s / 1e9 / HBM_PEAK_GBS if pass_s > 0 else None,
           "pass_frac_note": "one block ahead: a launch is a slice on 224 CUs beside the step kernels; judge by ms_per_step",
           "per_step_us": {"pass": p["block_kernel_seconds"] / n_prof * 1e6, "short_passes": p["mini_pass_seconds"] / n_prof * 1e6,
                           "scatter": p["scatter_kernel_seconds"] / n_prof * 1e6, "step_kernels": p["step_kernel_seconds"] / n_prof * 1e6}}
    out.update(dispatch_info([gpu]))
    out["synth_fill"] = synth_view(gpu)
    # back to back with each schedule pinned (same run-in)
    for name, sched in (("back_to_back_pass_at_block_start", 0), ("back_to_back_pass_one_block_ahead", 1)):
        gpu.set_pass_schedule(sched)
        lat = back_to_back(n_in, steps)
        out[name] = {"ms_per_step": float(lat.mean()) * 1e3, "median_ms_per_step": float(np.median(lat)) * 1e3, "max_ms_per_step": float(lat.max()) * 1e3, "steps": steps}

    # a caller that leaves the GPU idle between two force evaluations (300 us of host work: a busy wait): default and pinned schedules
    def gap_loop(n_skip, n):
        lat = np.zeros(n)
        for i in range(n_skip + n):
            k = pos[0] + i
            a = time.perf_counter()
            gpu.step(times[k], *states[k])
            b = time.perf_counter()
            if i >= n_skip:
                lat[i - n_skip] = b - a
            while time.perf_counter() - b < gap:
                pass
        pos[0] += n_skip + n
        return {"mean_step_us": float(lat.mean()) * 1e6, "median_step_us": float(np.median(lat)) * 1e6,
                "p90_step_us": float(np.percentile(lat, 90)) * 1e6, "max_step_us": float(lat.max()) * 1e6}
    loops = {"host_work_between_calls_us": gap * 1e6, "steps": n_gap}
    gpu.set_pass_schedule(-1)
    loops["adaptive_default"] = gap_loop(n_in, n_gap)
    gpu.set_pass_schedule(0)
    loops["pass_at_block_start"] = gap_loop(n_in, n_gap)
    gpu.set_pass_schedule(1)
    a0 = int(gpu.profile()["ahead_blocks"])
    loops["pass_one_block_ahead"] = gap_loop(n_in, n_gap)
    loops["pass_one_block_ahead"]["blocks_without_a_pass_of_their_own"] = int(gpu.profile()["ahead_blocks"]) - a0
    out["chrono_like_loop"] = loops
    gpu.close()
    return out


def small_system(name, case, poses, waves_kw, sdt, nsteps, what):
    """A BASELINE.json config that is a parity case rather than the headline (C2, C5 -- synthetic stand-ins, the reference's
    rm3.h5 / deepcwind.h5 are missing blobs): synchronous hc_step from a steady-state history through the C ABI's own loop
    (hc_step_many), beside the CPU oracle on this box's cores -- reference-faithful at ONE thread and at its best thread count, and
    the optimised flat variant.  For systems this small the GPU step is a PCIe round trip; SURVEY 8d (iii): "CPU may win; report honestly"."""
    import oracle as orc_mod
    from cases import load_into_oracle
    from hydrochrono_amd.hydro import HydroForces
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    N = case["N"]
    gpu, orc = HydroForces.from_case(case), load_into_oracle(case)
    gpu.add_waves_irregular(**waves_kw)
    orc.add_waves_irregular(**waves_kw)
    motion = PrescribedMotion(N, poses, seed=12)
    span = float(np.asarray(case["bodies"][0]["rirf_t"])[-1])
    nhist = int(np.ceil(span / sdt)) + 5
    t0 = nhist * sdt + 1.0
    t_hist = t0 - sdt * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    gpu.set_history(t_hist, v_hist)
    orc.prefill_history(t_hist, v_hist)
    n_warm = 72
    times = t0 + sdt * np.arange(n_warm + nsteps)
    states = np.stack([motion.packed(t) for t in times])
    forces, secs = gpu.step_many(times, states)
    lat = secs[n_warm:] * 1e6
    cores = os.cpu_count() or 1
    k = [0]
    f_orc = []

    def timed(fn, n, keep):
        d = []
        for _ in range(n):
            t = times[k[0]]
            st = motion.state(t)
            a = time.perf_counter()
            f = fn(t, *st)
            d.append(time.perf_counter() - a)
            if keep:
                f_orc.append(f)
            k[0] += 1
        return d
    sweep = {}
    for th in sorted({1, min(cores, 4), min(cores, 16), min(cores, 64)}):
        orc_mod.set_num_threads(th)
        sweep[th] = float(np.median(timed(orc.step, 12, True))) * 1e6
    best = min(sweep, key=sweep.get)
    flat = {}
    for th in sorted({1, min(cores, 16)}):
        orc_mod.set_num_threads(th)
        orc.flat_prepare()
        flat[th] = float(np.median(timed(orc.flat_step, 12, False)[2:])) * 1e6
    n_chk = len(f_orc)
    p = gpu.profile()
    out = {"workload": what, "bodies": N, "steps": nsteps, "step_dt": sdt,
           "gpu_hc_step_us": {"median": float(np.median(lat)), "mean": float(lat.mean()), "p90": float(np.percentile(lat, 90)), "max": float(lat.max())},
           "evals_per_s": 1e6 / float(lat.mean()),
           "cpu_oracle_us": {"faithful_one_thread": sweep[1], "faithful_best": sweep[best], "faithful_best_threads": best,
                             "faithful_threads_sweep": {str(th): v for th, v in sweep.items()},
                             "flat_one_thread": flat[1], "flat_best": min(flat.values()), "flat_best_threads": min(flat, key=flat.get),
                             "sample": "12 steps per thread count (median), the steps the GPU ran"},
           "gpu_vs_cpu": {"vs_faithful_one_thread": sweep[1] / float(lat.mean()), "vs_faithful_best": sweep[best] / float(lat.mean()),
                          "vs_flat_best": min(flat.values()) / float(lat.mean())},
           "parity_max_rel_err_vs_oracle": max_rel_err(forces[:n_chk], np.stack(f_orc)), "parity_steps": n_chk,
           "aql_dispatches": int(p["direct_dispatches"]), "hip_launches": int(p["hip_launches"])}
    out.update(dispatch_info([gpu]))
    gpu.close()
    return out


def c2_two_body():
    from hydrochrono_amd.synthetic import many_body_case
    case = many_body_case(2, S=1001, dt_rirf=0.015, n_exc=1001, dt_exc=0.125, seed=2)
    case["bodies"][0]["cg"], case["bodies"][1]["cg"] = np.array([0.0, 0.0, -0.72]), np.array([0.0, 0.0, -21.29])  # demos/rm3 poses
    for b in case["bodies"]:
        b["cb"] = b["cg"] + np.array([0.0, 0.0, 0.3])
    kw = dict(simulation_dt=0.01, simulation_duration=40.0, ramp_duration=5.0, wave_height=2.5, wave_period=8.0, frequency_min=0.02,
              frequency_max=0.5, nfrequencies=512, peak_enhancement_factor=3.3, seed=1)
    return small_system("c2", case, np.stack([b["cg"] for b in case["bodies"]]), kw, 0.01, 320,
                        "C2 stand-in (rm3.h5 is a missing blob): 2 bodies, S 1001 @ 0.015 s, JONSWAP Hs 2.5 / Tp 8 / gamma 3.3 / nf 512, dt 0.01")


def c5_one_body_2048():
    from hydrochrono_amd.synthetic import many_body_case
    case = many_body_case(1, S=401, dt_rirf=0.05, n_exc=401, dt_exc=0.25, seed=5)
    kw = dict(simulation_dt=0.08, simulation_duration=1000.0, ramp_duration=20.0, wave_height=6.0, wave_period=10.0, frequency_min=0.01,
              frequency_max=0.6, nfrequencies=2048, peak_enhancement_factor=2.0, seed=4)
    return small_system("c5", case, [case["bodies"][0]["cg"]], kw, 0.08, 320,
                        "C5 stand-in (deepcwind.h5 is a missing blob): 1 body, dt 0.08, 1000 s, nf 2048 (eta table by the direct FP64 sum)")


ADDED_MASS_HOST_MAX_DOFS = 96  # include/hydroc_amd/chloadaddedmass.h: kHostProductMaxDofs


def added_mass_product():
    """hc_added_mass_mv (Chrono's LoadIntLoadResidual_Mv, src/chloadaddedmass.cpp:55-70: one Eigen `R += c M w` on the host) at
    D = 6 / 12 / 96 / 192 / 288 / 384 / 3072: the GPU round trip per call beside the oracle's host loop -- and with it the size below which the
    C++ binding (include/hydroc_amd/chloadaddedmass.h) keeps the product on the host copy of the matrix it already holds."""
    import oracle as orc_mod
    from hydrochrono_amd import capi
    from hydrochrono_amd.hydro import HydroForces
    lib = capi.load()
    mv = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int)(("hc_added_mass_mv", lib))
    rows = {}
    for N in (1, 2, 16, 32, 48, 64, 512):
        D = 6 * N
        gpu = HydroForces(N, device=0)
        gpu.synth_fill(20251031, 8, DT, 0, DT)
        gpu.finalize()
        M = gpu.added_mass_matrix()
        rng = np.random.default_rng(N)
        w = rng.standard_normal(D)
        R = np.zeros(D)
        reps = 400 if N <= 64 else 120
        lat = np.zeros(reps)
        for i in range(40 + reps):
            R[:] = 0.0
            a = time.perf_counter()
            rc = mv(gpu.ctx, w.ctypes.data, 0.5, R.ctypes.data, D)
            b = time.perf_counter()
            if rc:
                raise RuntimeError(lib.hc_last_error(gpu.ctx).decode())
            if i >= 40:
                lat[i - 40] = b - a
        cpu_s, R_cpu = orc_mod.dense_mv_seconds(M, w, 0.5, max(3, int(1e8 / (D * D))))
        R_ref = 0.5 * (M @ w)
        rows[str(D)] = {"bodies": N, "gpu_us_per_call": {"median": float(np.median(lat)) * 1e6, "mean": float(lat.mean()) * 1e6},
                        "cpu_oracle_us_per_call_one_thread": cpu_s * 1e6, "gpu_over_cpu": float(np.median(lat)) / cpu_s,
                        "max_rel_err": float(np.max(np.abs(R - R_ref)) / np.max(np.abs(R_ref)))}
        gpu.close()
    ds = sorted(int(d) for d in rows)
    gpu_wins = [d for d in ds if rows[str(d)]["gpu_over_cpu"] < 1.0]
    return {"rows_by_D": rows, "crossover": {"host_wins_up_to_D": max([d for d in ds if rows[str(d)]["gpu_over_cpu"] >= 1.0], default=None),
                                             "gpu_wins_from_D": min(gpu_wins, default=None)},
            "binding_threshold": {"kHostProductMaxDofs": ADDED_MASS_HOST_MAX_DOFS,
                                  "meaning": "the C++ binding multiplies on its host copy for 6N <= this; the C ABI entry is always the GPU"},
            "note": "GPU: synchronous hc_added_mass_mv through ctypes (~1 us of interpreter included); CPU: the oracle's loop, -O2, one thread"}


PCIE_GEN5_X16_GBS = 63.0  # what a x16 PCIe 5.0 link carries (MI355X_MICROARCH.md: host link); the pinned-copy rate of this box is measured live


def _rate(nbytes, sec):
    return (nbytes / sec / 1e9) if sec and sec > 0 else None


def _ingest_view(st, file_bytes=None):
    """hc_init_stats of a context as GB/s per stage, each beside its bound."""
    out = {"rirf_h2d": {"seconds": st["rirf_h2d_seconds"], "bytes": st["rirf_h2d_bytes"], "GBps": _rate(st["rirf_h2d_bytes"], st["rirf_h2d_seconds"]),
                        "bound": "PCIe host link (pageable source: the runtime stages it through pinned buffers)"},
           "relayout_rirf_kernel": {"seconds": st["rirf_relayout_seconds"], "bytes_read_and_written": st["rirf_relayout_bytes"],
                                    "GBps": _rate(st["rirf_relayout_bytes"], st["rirf_relayout_seconds"]),
                                    "frac_of_hbm_peak": (_rate(st["rirf_relayout_bytes"], st["rirf_relayout_seconds"]) or 0.0) / HBM_PEAK_GBS, "bound": "hbm"},
           "hc_finalize": {"seconds": st["finalize_seconds"], "of_which_direct_dispatch_setup_and_selftests": st["direct_setup_seconds"]}}
    if st["h5_read_seconds"] > 0:
        out["hdf5_read"] = {"seconds": st["h5_read_seconds"], "bytes": st["h5_read_bytes"], "GBps": _rate(st["h5_read_bytes"], st["h5_read_seconds"]),
                            "file_bytes": file_bytes, "bound": "libhdf5 contiguous dataset reads from the page cache (the file was written a moment ago)"}
    return out


def _wave_view(st, oracle_s=None):
    nt, nf = st["wave_eta_samples"], st["wave_eta_components"]
    out = {"total_seconds": st["wave_total_seconds"], "resample_excitation_irf_seconds": st["wave_resample_seconds"],
           "spectrum_phases_wavenumber_seconds": st["wave_spectrum_seconds"],
           "eta_synthesis": {"seconds": st["wave_eta_seconds"], "mode": "rocFFT chirp-z" if st["wave_eta_mode"] == 1 else "direct FP64 sum (eta_kernel)",
                             "nt": int(nt), "nf": int(nf), "component_evaluations_per_s": (nt * nf / st["wave_eta_seconds"]) if st["wave_eta_seconds"] > 0 else None},
           "uploads_and_kex_relayout": {"seconds": st["wave_upload_seconds"], "bytes": st["wave_upload_bytes"]}}
    if oracle_s is not None:
        out["cpu_oracle_seconds_one_thread"] = oracle_s
    return out


def synth_view(gpu):
    """hc_synth_fill of a context: the generator kernel writes K once (counter-based values computed in registers)."""
    st = gpu.init_stats()
    gb = _rate(st["synth_bytes"], st["synth_seconds"])
    return {"synth_rirf_kernel_seconds": st["synth_seconds"], "bytes_written": st["synth_bytes"], "GBps": gb, "frac_of_hbm_peak": (gb or 0.0) / HBM_PEAK_GBS,
            "bound": "hbm (write stream)", "hc_finalize_seconds": st["finalize_seconds"]}


def _diff_stats(b, a):
    return {k: (b[k] - a[k]) if k.endswith(("_seconds", "_bytes")) else b[k] for k in b}


def init_block(case, c3_wave_stats, sdt, duration):
    """The INIT half of the path, timed stage by stage beside its bound and beside the CPU oracle's init (SURVEY 7 hard part (vii)):
    (a) hc_load_bemio_h5 of a C3-size BEMIO file written to /tmp by the committed generator (tests/golden/make_multibody_bemio.py), forces
    checked bitwise against the raw-setter ingest of the same arrays; (b) hc_set_wave_irregular at C3 (direct sum and rocFFT) and at the
    sphere-irregular size (nt 56 668 x nf 1 000); (c) the TaperedDirect build at C3.  (d), hc_synth_fill at C4, rides in c4_one_gpu."""
    import importlib.util
    import tempfile
    import torch
    import oracle as orc_mod
    from cases import load_into_oracle
    from hydrochrono_amd.hydro import HydroForces
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions
    N = case["N"]
    out = {}
    # the host link of THIS box: a pinned 256 MB copy
    try:
        hsrc = torch.empty(32 * 1024 * 1024, dtype=torch.float64).pin_memory()
        ddst = torch.empty_like(hsrc, device="cuda")
        ddst.copy_(hsrc, non_blocking=True)
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(4):
            ddst.copy_(hsrc, non_blocking=True)
        torch.cuda.synchronize()
        pinned = 4 * hsrc.numel() * 8 / (time.perf_counter() - a) / 1e9
        del hsrc, ddst
    except Exception:  # noqa: BLE001
        pinned = None
    out["host_link"] = {"pinned_h2d_GBps_measured": pinned, "pcie_gen5_x16_GBps": PCIE_GEN5_X16_GBS}
    # ---- (a) BEMIO ingest at C3 size ----
    path = os.path.join(tempfile.gettempdir(), f"hc_bench_c3_{os.getpid()}.h5")
    ing = {}
    try:
        spec = importlib.util.spec_from_file_location("make_multibody_bemio", os.path.join(ROOT, "tests", "golden", "make_multibody_bemio.py"))
        mm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mm)
        a = time.perf_counter()
        mm.write_bemio(case, [(path, False)])
        ing["file"] = {"bytes": os.path.getsize(path), "written_in_seconds": time.perf_counter() - a,
                       "generator": "tests/golden/make_multibody_bemio.py: write_bemio (the arrays of the C3 case; datasets of src/h5fileinfo.cpp:41-90)"}
        a = time.perf_counter()
        g = HydroForces(N)
        g.load_bemio_h5(path)
        t_load = time.perf_counter() - a
        g.finalize()
        t_all = time.perf_counter() - a
        st_file = g.init_stats()
        ing["hc_load_bemio_h5"] = {"seconds": t_load, "then_hc_finalize_seconds": t_all - t_load, **_ingest_view(st_file, ing["file"]["bytes"])}
        a = time.perf_counter()
        ref = HydroForces.from_case(case)
        t_raw = time.perf_counter() - a
        ing["raw_setters_same_arrays"] = {"seconds_including_finalize": t_raw, **_ingest_view(ref.init_stats())}
        a = time.perf_counter()
        orc = load_into_oracle(case)
        ing["cpu_oracle_ingest_seconds"] = time.perf_counter() - a
        ing["cpu_oracle_ingest_note"] = "the oracle's setters + construct(): rho / rho*g scaling and the nested-vector copies of the reference's H5FileInfo, no file I/O"
        # same forces from the file as from the arrays, bit for bit
        motion = PrescribedMotion(N, rest_positions(case), seed=7)
        for h in (g, ref):
            h.add_waves_regular(0.5, 0.8)
        same = True
        for k in range(24):
            st = motion.state(0.01 * (k + 1))
            same = same and bool(np.array_equal(g.step(0.01 * (k + 1), *st), ref.step(0.01 * (k + 1), *st)))
        ing["forces_from_file_bitwise_equal_to_raw_setters"] = {"steps": 24, "equal": same}
        out["bemio_ingest_c3"] = ing
        # ---- (b) irregular waves at C3: the main context's own call (direct sum) and the rocFFT form on the context read from the file ----
        waves = {"c3_direct_sum": _wave_view(c3_wave_stats)}
        kw = dict(WAVES, num_bodies=N, simulation_dt=sdt, simulation_duration=duration)
        s0 = g.init_stats()
        g.set_eta_synthesis(1)
        g.add_waves_irregular(**kw)
        waves["c3_rocfft"] = _wave_view(_diff_stats(g.init_stats(), s0))
        orc_mod.set_num_threads(1)
        a = time.perf_counter()
        orc.add_waves_irregular(**{k: v for k, v in kw.items() if k != "num_bodies"})
        waves["c3_direct_sum"]["cpu_oracle_seconds_one_thread"] = time.perf_counter() - a
        g.close()
        del orc
        # ---- (c) TaperedDirect build at C3 (the first step under the mode runs taper_kernel over all of K) ----
        ref.set_convolution_mode(1)
        st = motion.state(0.01 * 26)
        ref.step(0.01 * 26, *st)
        sr = ref.init_stats()
        gb = _rate(sr["taper_bytes"], sr["taper_seconds"])
        out["tapered_direct_build_c3"] = {"taper_kernel_seconds": sr["taper_seconds"], "bytes_read_and_written": sr["taper_bytes"], "GBps": gb,
                                          "frac_of_hbm_peak": (gb or 0.0) / HBM_PEAK_GBS, "bound": "hbm",
                                          "note": "SG-5 smoothing + half-cosine taper of every (row, column) series (src/hydro_forces.cpp:385-535), K read once and written once"}
        ref.close()
    except Exception as e:  # noqa: BLE001
        out["bemio_ingest_c3"] = {**ing, "error": str(e)}
        waves = {"c3_direct_sum": _wave_view(c3_wave_stats)}
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    # ---- (b') the sphere-irregular size: one body, nt 56 668, nf 1 000 (src/wave_types.cpp:717-774; demos/sphere irregular) ----
    try:
        c1 = many_body_case(1, S=401, dt_rirf=0.05, n_exc=401, dt_exc=0.05, seed=3)
        kw1 = dict(simulation_dt=0.015, simulation_duration=850.0, ramp_duration=60.0, wave_height=2.0, wave_period=12.0, frequency_min=0.001,
                   frequency_max=1.0, nfrequencies=1000, peak_enhancement_factor=1.0, seed=1)
        for mode, key in ((0, "sphere_irregular_size_direct_sum"), (1, "sphere_irregular_size_rocfft")):
            h = HydroForces.from_case(c1)
            h.set_eta_synthesis(mode)
            s0 = h.init_stats()
            h.add_waves_irregular(**kw1)
            waves[key] = _wave_view(_diff_stats(h.init_stats(), s0))
            h.close()
        o1 = load_into_oracle(c1)
        orc_mod.set_num_threads(1)
        a = time.perf_counter()
        o1.add_waves_irregular(**kw1)
        waves["sphere_irregular_size_direct_sum"]["cpu_oracle_seconds_one_thread"] = time.perf_counter() - a
    except Exception as e:  # noqa: BLE001
        waves["sphere_irregular_size_error"] = str(e)
    out["irregular_waves"] = waves
    return out


C4_ONE_GPU_FILES = ("profiles/r06/bench_c4_1gpu.json", "profiles/r05/bench_c4_1gpu.json", "profiles/r04/bench_c4_1gpu.json", "profiles/r03/bench_c4_1gpu.json")


def c4_one_gpu_reference():
    """The C4 workload (ONE coupled 512-body array) on ONE MI355X, from the committed measurement: the denominator of a strong-scaling
    ratio for the N > 1 lines (the N = 1 line of this benchmark is C3, BASELINE.json's metric, and carries the same figure live as
    `c4_one_gpu`)."""
    for rel in C4_ONE_GPU_FILES:
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            try:
                rj = json.load(open(path))
                sec = rj.get("c4_one_gpu") or rj
                v = sec.get("evals_per_s", sec.get("value"))
                if v:
                    return {"evals_per_s": float(v), "ms_per_step": float(sec["ms_per_step"]), "workload": "C4",
                            "measured": f"{rel} (one MI355X, the same generator and step loop; committed, not measured in this run)"}
            except Exception:
                continue
    return None


def c4_one_gpu(sdt, lookahead):
    """C4 -- the whole coupled 512-body array (K = 77.3 GB, generated in HBM) -- on THIS GPU: synchronous hc_step, so that the N > 1
    lines of this benchmark (the same array row-sharded over N GPUs) have a one-GPU figure of the SAME workload to be divided by."""
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    N, warm, steps = N_BODIES_C4, 104, 256  # (steady state: a 96-step window after 40 caught less than its share of the passes, 0.43 against 0.51 ms per step)
    motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
    nhist = int(np.ceil(S_RIRF * DT / sdt)) + 5
    t_hist = T0 - sdt * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    n_all = warm + steps
    gpu = make_shard(N, 0, N, 0, sdt, T0 + (n_all + 8) * sdt + 5.0, lookahead, t_hist, v_hist)
    times = [T0 + k * sdt for k in range(n_all)]
    states = [motion.state(t) for t in times]
    gpu.enable_profiling(1)
    for k in range(warm):
        gpu.step(times[k], *states[k])
    gpu.reset_profile()
    per = np.zeros(steps)
    t0 = time.perf_counter()
    for i in range(steps):
        a = time.perf_counter()
        gpu.step(times[warm + i], *states[warm + i])
        per[i] = time.perf_counter() - a
    elapsed = time.perf_counter() - t0
    p = gpu.profile()
    launches = max(1, p["block_kernel_launches"])
    # a pass one block ahead goes out in slices: per block = seconds of all its launches
    pass_s_per_block = p["block_kernel_seconds"] / max(1.0, steps / float(lookahead))
    once_per_block = 8.0 * (6.0 * N) * (6.0 * N) * S_RIRF
    out = {"workload": f"C4: ONE coupled synthetic {N}-body array (D = {6 * N}, K = {once_per_block / 1e9:.1f} GB FP64) on one GPU, synchronous "
                       f"hc_step through the Python wrapper, {steps} steps after {warm}",
           "evals_per_s": steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "median_ms_per_step": float(np.median(per)) * 1e3,
           "max_ms_per_step": float(per.max()) * 1e3, "steps": steps, "lookahead": lookahead,
           "pass_schedule": "one block ahead (slices on the pass lane)" if p["ahead_blocks"] > 0 else "at block start",
           "pass_launches": int(p["block_kernel_launches"]), "pass_us_per_launch": 1e6 * p["block_kernel_seconds"] / launches,
           "pass_ms_per_block": pass_s_per_block * 1e3,
           "pass_frac_of_hbm_peak": (p["block_kernel_bytes_once"] / (p["block_kernel_seconds"] / launches) / 1e9 / HBM_PEAK_GBS) if p["block_kernel_seconds"] > 0 else None,
           "per_step_us": {"pass": p["block_kernel_seconds"] / steps * 1e6, "short_passes": p["mini_pass_seconds"] / steps * 1e6,
                           "scatter": p["scatter_kernel_seconds"] / steps * 1e6, "step_kernels": p["step_kernel_seconds"] / steps * 1e6},
           "note": "the one-GPU figure of the workload the --gpus N > 1 lines shard (their c4_one_gpu_reference / speedup_vs_c4_one_gpu)"}
    out.update(dispatch_info([gpu]))
    out["synth_fill"] = synth_view(gpu)
    gpu.close()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    single = world == 1 and (args.gpus > 1 or args.single_process)
    if world != args.gpus and not single:
        args.gpus = world
    if args.scaling is None:
        args.scaling = "strong" if (world > 1 or single) else "weak"
    strong = args.scaling == "strong"
    if args.bodies is None:
        args.bodies = N_BODIES_C4 if strong else N_BODIES

    rccl_child = None
    if single and not args.no_secondary and not args.single_process:
        # (before this process touches the GPU: the child gets the devices to itself; counting devices does not initialise them)
        import torch as _t
        rccl_child = spawn_rccl_ranks(args, args.gpus, args.gpus if args.stub_context else _t.cuda.device_count())
    import torch
    import torch.distributed as dist
    from hydrochrono_amd import capi
    from hydrochrono_amd.hydro import HydroForces
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions

    stub = bool(args.stub_context)
    if stub:
        # the N > 1 control flow without a GPU: CPU tensors, gloo, tests/stub_context.StubShard instead of the library's contexts
        if (world < 2 and not single) or not strong:
            sys.exit("--stub-context is the CPU rehearsal of the multi-rank (--scaling strong) flow only")
        from stub_context import StubShard
        HydroForces = StubShard  # noqa: N806
        args.no_secondary = False
    elif not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hydro-force path has no CPU fallback")
    sdt = args.step_dt
    ndev = 0 if stub else torch.cuda.device_count()
    cuda_sync = (lambda: None) if stub else torch.cuda.synchronize

    if single:
        # ---- ONE process, G contexts, hc_step_multi: the multi-GPU path of a Chrono host through the C ABI ----
        G = args.gpus
        devices = [g % max(1, ndev) for g in range(G)]
        if not stub and not args.no_pin:
            capi.load().hc_bind_thread_to_device(devices[0])  # (the calling thread runs context 0; the workers of hc_step_multi bind themselves)
        info, _ = run_group_sync(args.bodies, G, devices, sdt, args.lookahead, args.warmup, args.steps, check_steps=0 if args.no_secondary else 40,
                                 stub_cls=HydroForces if stub else None)
        out = {"metric": "hydro-force evals/sec (all bodies)", "value": info["evals_per_s"], "unit": "evals/s", "n_gpus": G, "steps": args.steps,
               "warmup": args.warmup, "alignment_steps": info["alignment_steps"], "passes_in_timed_region": info["passes_in_timed_region"],
               "ms_per_step": info["ms_per_step"], "median_ms_per_step": info["median_ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"C4-style: ONE coupled synthetic {args.bodies}-body array row-sharded over {G} contexts of ONE process "
                                      f"(devices {devices}; {ndev} visible), one hc_step_multi per step: state stored into every context, "
                                      "all step kernels dispatched, host-side gather of the force rows; no collective",
                          "bodies": args.bodies, "irf_samples": S_RIRF, "wave_components": WAVES["nfrequencies"], "lookahead": info["lookahead"],
                          "sharding": "body-row shards, single process, host gather (SURVEY 8e drop-in variant)"},
               "exchange": "host gather inside hc_step_multi (no collective)",
               "roofline": info.pop("roofline"), "per_shard": info.pop("per_shard"), "exchange_check": info.pop("exchange_check", None),
               "cpu_baseline": None,
               "cpu_baseline_note": "not timed for the sharded workload: the CPU oracle at C4 size (512 bodies, 77 GB of K) does not fit a bounded sample; the N = 1 "
                                    "line times it on the C3 workload (BASELINE.json's metric)",
               "rccl_ranks": rccl_child,
               "single_process": info}
        out.update({k: info[k] for k in ("dispatch_mode", "dispatch_mode_reason")})
        if args.bodies == N_BODIES_C4:
            out["workload"] = "C4"
            ref1 = c4_one_gpu_reference()
            if ref1:
                out["c4_one_gpu_reference"] = ref1
                out["speedup_vs_c4_one_gpu"] = out["value"] / ref1["evals_per_s"]
                out["speedup_note"] = "the reference is a 256-step steady-state run; this line's timed region is aligned to hold its share of passes"
        if stub:
            out["stub_context"] = "tests/stub_context.StubShard: the control flow on CPU, no GPU, no physics -- NOT a measurement"
        elif ndev < G:
            out["note"] = f"only {ndev} GPU(s) visible: contexts share devices (functional run, not a scaling figure)"
        print(json.dumps(out), flush=True)
        return

    # HC_BENCH_SHARE_GPU=1 (functional test of the N > 1 code path on a one-GPU box): all ranks use device 0 and the
    # force all-gather goes over gloo instead of RCCL.  Never used for reported numbers.
    share_gpu = os.environ.get("HC_BENCH_SHARE_GPU") == "1" or stub
    if share_gpu and not stub:
        local_rank = 0
        os.environ["HC_DEVICE_SHARED"] = "1"  # the library: other processes hold contexts on this device too (no queue parking, no pass lane)
    if not stub:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from hydrochrono_amd.parallel import ForceExchange, body_shard
    # The stepping thread next to its GPU (a synchronous step is a round trip between one host thread and one GPU;
    # from the other socket it costs 1.4 us more, and an unpinned thread lands on either socket).  hc_bind_thread_to_device restricts
    # THIS thread to the CPUs local to the device's PCIe root; the CPU baseline below gets the original mask back.
    host_thread = {"bound_to_device_local_cpus": False}
    affinity_before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    if not stub and not args.no_pin and rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (the CPU oracle's OpenMP workers are created HERE, with the mask of every core, before the stepping thread narrows its own:
        # threads inherit the mask of the thread that creates them)
        try:
            import oracle as orc_mod
            from cases import load_into_oracle
            orc_mod.set_num_threads(os.cpu_count() or 1)
            tiny = many_body_case(1, S=16, dt_rirf=0.05, n_exc=9, dt_exc=0.1, seed=1)
            o_ = load_into_oracle(tiny)
            o_.add_waves_none()
            z3 = np.zeros(3)
            for k_ in range(3):
                o_.step(0.05 * k_, z3, z3, z3 + 0.1 * k_, z3)
            del o_
        except Exception:  # noqa: BLE001
            pass
    if not stub and not args.no_pin:
        buf = C.create_string_buffer(512)
        if capi.load().hc_device_local_cpus(local_rank, buf, 512) == 0 and capi.load().hc_bind_thread_to_device(local_rank) == 0:
            host_thread = {"bound_to_device_local_cpus": True, "cpus": buf.value.decode(), "how": "hc_bind_thread_to_device (sched_setaffinity of the stepping thread)"}
    N = args.bodies
    case = None
    if strong:
        # one coupled N-body array, this rank owns the output rows of bodies [b0, b1); inputs generated in HBM
        b0, b1 = body_shard(N, world, rank)
        gpu = HydroForces(N, device=local_rank, body_range=(b0, b1))
        gpu.synth_fill(20251031, S_RIRF, DT, N_EXC, DT)
        gpu.finalize()
        motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
        exchange = ForceExchange(N, world, rank, device="cpu" if share_gpu else "cuda") if world > 1 else None
    else:
        case = many_body_case(N, S=S_RIRF, dt_rirf=DT, n_exc=N_EXC, dt_exc=DT, seed=20251031 + rank)
        gpu = HydroForces.from_case(case, device=local_rank)
        motion = PrescribedMotion(N, rest_positions(case), seed=20251031 + rank)
        exchange = None  # independent farms: nothing to exchange
    gpu.set_lookahead(args.lookahead)
    eff_lookahead = gpu.schedule()["lookahead"]  # (what this build of the library made of --lookahead: the release library holds 16 and 32)
    if eff_lookahead != args.lookahead and args.lookahead > 0:
        if rank == 0:
            print(f"bench.py: --lookahead {args.lookahead} is not in this build of the library (HYDROCHRONO_AMD_FLAVOR=tuning holds the depth-64 pass): "
                  f"running depth {eff_lookahead}", file=sys.stderr)
        args.lookahead = eff_lookahead
    pinned_schedule = None
    if world > 1 and strong:
        # ONE pass schedule for the row shards of one array: under the adaptive default every process would judge its own caller's gaps, the
        # ranks could answer differently block by block (different summation grouping: rows no longer bitwise those of the unsharded array)
        # and pay their passes on different steps, while every step waits for the slowest rank.  Rank 0's answer, pinned on all.
        box = [gpu.schedule()["ahead_now"] if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        pinned_schedule = int(box[0])
        gpu.set_pass_schedule(pinned_schedule)
    n_steady = args.steady_steps if (world == 1 and not args.no_secondary) else 0
    n_pipe = 0 if (args.no_secondary or world > 1) else args.steps
    n_plain = 0 if (args.no_secondary or world > 1 or args.lookahead == 0) else max(20, args.steps // 8)
    # Phase alignment (untimed, before the W warm-up steps): one look-ahead pass serves a block of `lookahead` steps and is paid
    # by the first step of the next block, so a short timed region could fall between two passes and report the median step as
    # if it were the mean.  The region is placed so that a block boundary lies in its middle: it then carries
    # max(1, ~K/lookahead) passes -- pessimistic for K < lookahead, neutral for K >> lookahead.
    align = 0
    if args.lookahead > 0:
        Lb = 16 if args.lookahead <= 16 else (32 if args.lookahead <= 32 else 64)  # (the depth the library runs, read back above)
        first = args.warmup + args.steps // 2          # earliest step index the boundary (a multiple of Lb) may have
        boundary = ((first + Lb - 1) // Lb) * Lb
        align = boundary - first
    pre = align + args.warmup
    total = pre + args.steps
    n_py = 128 if (n_steady > 0 and not args.python_loop) else 0
    n_all = total + n_steady + n_py + ((2 * (72 + 128) + (104 + 128) + 3 * 128 + 3) if n_steady > 0 else 0) + n_pipe + n_plain + 16 + ((16 + args.steps) if world > 1 else 0)
    # the wave model is built for the caller's step size; the free-surface table must cover every step of this run
    duration = max(WAVES["simulation_duration"], T0 + n_all * sdt + 5.0)
    gpu.add_waves_irregular(**dict(WAVES, num_bodies=N, simulation_dt=sdt, simulation_duration=duration))
    c3_wave_stats = gpu.init_stats() if (world == 1 and not stub and not strong) else None  # (the `init` block: this call's stages at C3)
    D_local = gpu.D_local

    nhist = int(np.ceil(S_RIRF * DT / sdt)) + 5
    t_hist = T0 - sdt * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    gpu.set_history(t_hist, v_hist)

    times = [T0 + k * sdt for k in range(n_all)]
    times_np = np.array(times, dtype=np.float64)
    states = np.ascontiguousarray(np.stack([motion.packed(t) for t in times]))  # [n_all][12N] = pos | rpy | linvel | angvel
    forces = np.zeros((n_all, D_local))
    n3 = 3 * N
    # raw entry point: integer addresses straight through (the timed loop is the C-ABI call and nothing else)
    hc_step = None if stub else C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(
        ("hc_step", capi.load()))
    hc_step_many_raw = None if stub else C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(
        ("hc_step_many", capi.load()))
    ctx = gpu.ctx
    sp = [states.ctypes.data + k * states.strides[0] for k in range(n_all)]
    fp = [forces.ctypes.data + k * forces.strides[0] for k in range(n_all)]
    per_step = np.zeros(n_all)

    # An explicit stream for everything torch enqueues (copies, the RCCL all-gather) and for hc_step_device: the legacy default
    # stream has handle 0, which hc_step_device reads as "the context's own non-blocking stream" -- torch work would not be
    # ordered against the step kernels there.
    stream = None if stub else torch.cuda.Stream()
    if stream is not None:
        torch.cuda.set_stream(stream)
    stream_handle = stream.cuda_stream if stream is not None else 0
    hx = None
    n_other = 0  # steps of the exchange mode that is NOT `value`, run after the timed region as a secondary
    if exchange is not None:
        from hydrochrono_amd.host_exchange import HostExchange
        n_other = 0 if args.no_secondary else 16 + args.steps
        n_dev = total if args.exchange == "rccl" else n_other           # steps that go through the device path
        k_dev0 = 0 if args.exchange == "rccl" else total
        d_states = torch.tensor(states[k_dev0:k_dev0 + n_dev], device="cpu" if stub else "cuda")
        state_ptrs = {k_dev0 + k: d_states.data_ptr() + k * d_states.stride(0) * 8 for k in range(n_dev)}
        # host-gather steps land in host memory (where the integrator wants them); the RCCL path gathers on the device
        gathered = torch.zeros(total + n_other, 6 * N, dtype=torch.float64, device="cpu" if (share_gpu or args.exchange == "host") else "cuda")
        gathered_np = gathered.numpy() if gathered.device.type == "cpu" else np.zeros((total + n_other, 6 * N))
        g_dev = torch.zeros(n_dev, 6 * N, dtype=torch.float64, device="cpu" if share_gpu else "cuda")  # rows [k - k_dev0] of the device path
        # this rank's rows as the kernel left them: the step kernel writes them here and the all-gather sends them from here (no
        # staging copy on the stream between the two)
        own_rows = torch.zeros(total + n_other, exchange.max_rows, dtype=torch.float64, device="cpu" if stub else "cuda")
        own_ptrs = [own_rows.data_ptr() + k * own_rows.stride(0) * 8 for k in range(total + n_other)]
        own_host = np.zeros((total + n_other, D_local))
        direct_gather = exchange.even and not share_gpu  # equal shards: the collective writes the step's row of g_dev itself
        hx = HostExchange(gpu, N, world, rank, tag=f"hc_bench_{os.environ.get('MASTER_PORT', '0')}")
        dist.barrier()
        hx.attach()
        if stub:
            hc_begin, hc_end = gpu.begin_raw, gpu.end_raw
        else:
            hc_begin = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(("hc_step_begin", capi.load()))
            hc_end = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(("hc_step_end", capi.load()))
        cuda_sync()

    def run_sync_host(k0, k1):
        """Coupled array over several ranks, host gather: every rank hands its shard's step to its GPU (hc_step_begin: state through
        the BAR, direct dispatch), collects the rows of ALL shards from the shared-memory result buffers the GPUs write, completes
        its own step (hc_step_end).  The next step starts when this rank holds the full 6N vector on the host."""
        pc = time.perf_counter
        seq = hx.sequence()
        for k in range(k0, k1):
            a = pc()
            rc = hc_begin(ctx, times[k], sp[k], sp[k] + 8 * n3, sp[k] + 16 * n3, sp[k] + 24 * n3)
            if rc:
                gpu._chk(rc)
            seq += 1
            hx.gather_into(seq, gathered_np[k].ctypes.data)
            rc = hc_end(ctx, own_host[k].ctypes.data)
            per_step[k] = pc() - a
            if rc:
                gpu._chk(rc)

    def run_sync_rccl(k0, k1):
        # the step kernel writes this rank's rows, the RCCL all-gather leaves the full 6N force vector on every rank's GPU (the
        # device-side form of the one exchange step of the path), and the stream is synchronised: the next step
        # starts only when every rank holds all forces of this one
        pc = time.perf_counter
        for k in range(k0, k1):
            a = pc()
            gpu.step_device(times[k], state_ptrs[k], own_ptrs[k], stream_handle)
            row = g_dev[k - k_dev0]
            if share_gpu:  # functional mode on one GPU (or none): gloo moves the rows
                if stream is not None:
                    stream.synchronize()
                row.copy_(exchange.gather(own_rows[k, : exchange.rows].cpu()))
            elif direct_gather:
                dist.all_gather_into_tensor(row, own_rows[k], group=exchange.group)
                stream.synchronize()
            else:
                row.copy_(exchange.gather(own_rows[k, : exchange.rows]), non_blocking=True)
                stream.synchronize()
            per_step[k] = pc() - a

    def run_sync(k0, k1):
        """K synchronous evaluations: state from host memory in, forces in host memory out, one call after the other."""
        if exchange is None and not args.python_loop:
            # the library's own prescribed-motion loop: hc_step k0 .. k1 - 1 one after the other, no interpreter between the calls --
            # entered through a raw prototype with the addresses worked out beforehand (the arrays are C-contiguous float64 by
            # construction), so that a 20-step timed region holds the 20 steps and one foreign call, not the wrapper's array checks
            rc = hc_step_many_raw(ctx, k1 - k0, times_np.ctypes.data + 8 * k0, sp[k0], fp[k0], per_step.ctypes.data + 8 * k0, None)
            if rc:
                gpu._chk(rc)
        elif exchange is None:
            pc = time.perf_counter
            for k in range(k0, k1):
                a = pc()
                rc = hc_step(ctx, times[k], sp[k], sp[k] + 8 * n3, sp[k] + 16 * n3, sp[k] + 24 * n3, fp[k])
                per_step[k] = pc() - a
                if rc:
                    gpu._chk(rc)
        elif args.exchange == "host":
            run_sync_host(k0, k1)
        else:
            run_sync_rccl(k0, k1)

    def prof_diff(p1, p0):
        return {k: (p1[k] - p0[k]) if not k.endswith("_bytes") and not k.endswith("_bytes_once") else p1[k] for k in p1}

    # profiling runs through warm-up, timed region and the steady-state block: every look-ahead pass of the run is timed (the
    # per-step launches every `profile_stride`-th step), so the roofline rests on more than the one pass a 20-step region holds
    gpu.enable_profiling(args.profile_stride)
    gpu.reset_profile()
    run_sync(0, pre)
    cuda_sync()
    prof_pre = gpu.profile()
    if world > 1:
        dist.barrier()
    cuda_sync()
    t_start = time.perf_counter()
    run_sync(pre, total)
    cuda_sync()
    elapsed_own = time.perf_counter() - t_start  # this rank's own time, before it waits for the slowest
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    prof_end = gpu.profile()
    prof = prof_diff(prof_end, prof_pre)  # the timed region alone
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    exchange_ok = None
    if exchange is not None and args.exchange == "rccl":
        gathered[:total].copy_(g_dev[:total])
    if exchange is not None:
        # the one exchange step of the path, checked after the run: every rank finds its own rows, bit for bit, at its place in
        # the gathered vector of every step, and all ranks hold the same gathered vectors (checksums of the raw bits)
        b0_, b1_ = exchange.shards[rank]
        if args.exchange == "host":
            own_ok = bool(np.array_equal(gathered_np[:total, 6 * b0_:6 * b1_], own_host[:total]))
        else:
            own_ok = bool(torch.equal(gathered[:total, 6 * b0_:6 * b1_].to(own_rows.device), own_rows[:total, : exchange.rows]))
        dev_ = "cpu" if share_gpu else "cuda"
        bits = gathered[:total].contiguous().view(torch.int64).to(dev_)
        csum = (bits & 0xFFFFFFFF).sum(dim=1) + (bits >> 32).sum(dim=1)  # per-step checksum, exact in int64
        lo, hi = csum.clone().to(dev_), csum.clone().to(dev_)
        flag = torch.tensor([1 if own_ok else 0], dtype=torch.int64, device=dev_)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        exchange_ok = {"own_rows_bitwise_on_every_rank": bool(flag.item() == 1), "all_ranks_hold_the_same_vectors": bool(torch.equal(lo, hi)),
                       "steps_checked": int(total)}

    # ---- N > 1: what every rank saw, so that ONE run on a multi-GPU node yields the diagnosis (which rank is slow, and in which kernel) ----
    per_rank = None
    if world > 1:
        mine = per_step[pre:total] * 1e3
        nst = max(1, args.steps)
        me = {"rank": rank, "device": (-1 if stub else local_rank), "rows": int(D_local), "ms_per_step_own_loop": elapsed_own / nst * 1e3,
              "median_ms_per_step": float(np.median(mine)), "p90_ms_per_step": float(np.percentile(mine, 90)), "max_ms_per_step": float(mine.max()),
              "kernel_us_per_step": {"pass": prof["block_kernel_seconds"] / nst * 1e6, "short_passes": prof["mini_pass_seconds"] / nst * 1e6,
                                     "scatter": prof["scatter_kernel_seconds"] / nst * 1e6, "step_kernels": prof["step_kernel_seconds"] / nst * 1e6,
                                     "note": "the per-step launches are sampled (--profile-stride), every pass is timed"},
              "passes_in_timed_region": int(prof["block_kernel_launches"]), "blocks_with_rows_made_ahead": int(prof["ahead_blocks"]),
              "aql_dispatches": int(prof["direct_dispatches"]), "hip_launches": int(prof["hip_launches"])}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, me)

    # ---- secondary figures (not `value`) ----
    steady = pipelined = plain = None
    k_next = total
    if n_steady > 0:
        # steady state, median of >= 200 synchronous steps (the fixed driver command times 20)
        t_s = time.perf_counter()
        run_sync(k_next, k_next + n_steady)
        t_s = time.perf_counter() - t_s
        ps = per_step[k_next:k_next + n_steady] * 1e3
        steady = {"steps": n_steady, "evals_per_s": n_steady / t_s, "mean_ms_per_step": t_s / n_steady * 1e3, "median_ms_per_step": float(np.median(ps)),
                  "p10_ms_per_step": float(np.percentile(ps, 10)), "p90_ms_per_step": float(np.percentile(ps, 90)),
                  "note": "synchronous hc_step calls right after the timed region, passes included"}
        k_next += n_steady
    python_loop = None
    if n_py > 0:
        # the same synchronous calls issued one by one from the interpreter (how rounds 1-4a timed `value`): 4 blocks, passes included
        pc = time.perf_counter
        t_p = pc()
        for k in range(k_next, k_next + n_py):
            a = pc()
            rc = hc_step(ctx, times[k], sp[k], sp[k] + 8 * n3, sp[k] + 16 * n3, sp[k] + 24 * n3, fp[k])
            per_step[k] = pc() - a
            if rc:
                gpu._chk(rc)
        t_p = pc() - t_p
        pp = per_step[k_next:k_next + n_py] * 1e3
        python_loop = {"steps": n_py, "evals_per_s": n_py / t_p, "mean_ms_per_step": t_p / n_py * 1e3, "median_ms_per_step": float(np.median(pp)),
                       "note": "hc_step through ctypes, one call per interpreter iteration"}
        k_next += n_py
    prof_all = gpu.profile()  # warm-up + timed region + steady-state block
    gpu.enable_profiling(False)
    chrono_like = None
    if n_steady > 0:
        # What a Chrono loop sees: the host integrates between two force evaluations, and what a step leaves for later steps (scatter,
        # the pass of the next block) runs meanwhile.  100 us of host work between calls (a busy wait standing in for DoStepDynamics'
        # own share; the reference's whole RM3 step takes 360 us), only the calls are timed.
        n_cl, n_in = 128, 72   # timed steps per loop; run-in after a change of schedule (plain boundary step, a block with its own pass, the first block made ahead)
        n_in_adaptive = 104    # ... and one more block for the adaptive rule: it counts a block of gaps before it answers
        pc = time.perf_counter

        def gap_loop(k0, skip, work):
            lat_ = np.zeros(n_cl)
            for i in range(skip + n_cl):
                k = k0 + i
                a = pc()
                rc = hc_step(ctx, times[k], sp[k], sp[k] + 8 * n3, sp[k] + 16 * n3, sp[k] + 24 * n3, fp[k])
                b_ = pc()
                if i >= skip:
                    lat_[i - skip] = b_ - a
                if rc:
                    gpu._chk(rc)
                while pc() - b_ < work:
                    pass
            return {"mean_hc_step_us": float(lat_.mean()) * 1e6, "median_hc_step_us": float(np.median(lat_)) * 1e6,
                    "p90_hc_step_us": float(np.percentile(lat_, 90)) * 1e6, "max_hc_step_us": float(lat_.max()) * 1e6}
        # Under the library's DEFAULT schedule -- adaptive: the rule sees the gaps of a block and puts the next pass one block ahead
        # (run-in: the boundary step, the block whose gaps are counted, the first block made ahead) ...
        p0 = gpu.profile()
        chrono_like = {"steps": n_cl, "host_work_between_calls_us": 100.0, "pass_schedule": "adaptive (the library's default, hc_set_pass_schedule(ctx, -1, 0))",
                       **gap_loop(k_next, n_in_adaptive, 100e-6),
                       "note": "100 us of host work (a busy wait) between synchronous hc_step calls; only the calls are timed"}
        k_next += n_in_adaptive + n_cl
        short_gap = {"host_work_between_calls_us": 30.0, "adaptive_default": gap_loop(k_next, 0, 30e-6)}
        k_next += n_cl
        p1 = gpu.profile()
        chrono_like["schedule_answers"] = {"one_block_ahead": int(p1["schedule_blocks_ahead"] - p0["schedule_blocks_ahead"]),
                                           "at_block_start": int(p1["schedule_blocks_at_start"] - p0["schedule_blocks_at_start"]),
                                           "blocks_that_started_with_rows_made_ahead": int(p1["ahead_blocks"] - p0["ahead_blocks"])}
        # ... and with each schedule pinned
        gpu.set_pass_schedule(0)
        chrono_like["pass_at_block_start"] = {"steps": n_cl, **gap_loop(k_next, n_in, 100e-6), "note": "hc_set_pass_schedule(ctx, 0, 0)"}
        k_next += n_in + n_cl
        short_gap["pass_at_block_start"] = gap_loop(k_next, 0, 30e-6)
        k_next += n_cl
        gpu.set_pass_schedule(1)
        chrono_like["pass_one_block_ahead"] = {"steps": n_cl, **gap_loop(k_next, n_in, 100e-6),
                                               "note": "hc_set_pass_schedule(ctx, 1, 0)"}
        k_next += n_in + n_cl
        short_gap["pass_one_block_ahead"] = gap_loop(k_next, 0, 30e-6)
        k_next += n_cl
        chrono_like["pass_one_block_ahead"]["blocks_without_a_pass_of_their_own"] = int(gpu.profile()["ahead_blocks"] - p1["ahead_blocks"])
        chrono_like["with_30us_of_host_work"] = short_gap
        gpu.set_pass_schedule(-1)
    if n_pipe > 0:
        # hc_step_device, states resident in HBM, every step enqueued without waiting for the previous one's forces.
        # Run-in (untimed): the change of pass schedule above dropped the look-ahead plan, so the next step is a PLAIN one (all of K
        # streamed: 190 us) followed by a pass -- 380 us that a 20-step loop would carry as 19 us per step (BENCH_r03: 26 k evals/s
        # against 60 k in BENCH_r02, which had no schedule change in front of this loop).
        if n_steady > 0:
            run_sync(k_next, k_next + 3)
            k_next += 3
        d_states = torch.tensor(states[k_next:k_next + n_pipe], device="cuda")
        d_out = torch.zeros(n_pipe, D_local, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        s_ptr, o_ptr, sb, ob = d_states.data_ptr(), d_out.data_ptr(), d_states.stride(0) * 8, d_out.stride(0) * 8
        tp = time.perf_counter()
        for k in range(n_pipe):
            gpu.step_device(times[k_next + k], s_ptr + k * sb, o_ptr + k * ob, stream.cuda_stream)
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp
        forces[k_next:k_next + n_pipe] = d_out.cpu().numpy()
        pipelined = {"evals_per_s": n_pipe / tp, "ms_per_step": tp / n_pipe * 1e3, "steps": n_pipe,
                     "note": "hc_step_device enqueued ahead of the GPU (not usable by a Chrono loop)"}
        k_next += n_pipe
    if n_plain > 0:
        gpu.set_lookahead(0)
        for k in range(k_next, k_next + 4):
            hc_step(ctx, times[k], sp[k], sp[k] + 8 * n3, sp[k] + 16 * n3, sp[k] + 24 * n3, fp[k])
        gpu.enable_profiling(args.profile_stride)
        gpu.reset_profile()
        tp = time.perf_counter()
        for k in range(k_next + 4, k_next + 4 + n_plain):
            hc_step(ctx, times[k], sp[k], sp[k] + 8 * n3, sp[k] + 16 * n3, sp[k] + 24 * n3, fp[k])
        tp = time.perf_counter() - tp
        pp = gpu.profile()
        gpu.enable_profiling(False)
        kus = 1e6 * pp["conv_kernel_seconds"] / max(1, pp["conv_kernel_launches"])
        plain = {"evals_per_s": n_plain / tp, "ms_per_step": tp / n_plain * 1e3, "steps": n_plain, "kernel": "hc::conv_step_kernel",
                 "mean_kernel_us": kus, "algorithmic_bytes_per_launch": pp["conv_kernel_bytes"],
                 "achieved_GBps": pp["conv_kernel_bytes"] / (kus * 1e-6) / 1e9 if kus > 0 else 0.0,
                 "frac_of_hbm_peak": (pp["conv_kernel_bytes"] / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS) if kus > 0 else 0.0,
                 "note": "synchronous hc_step with look-ahead off: one launch streams all of K per step"}
        k_next += 4 + n_plain
    dinfo = dispatch_info([gpu])
    dinfo["aql_dispatches"], dinfo["hip_launches"] = int(prof_all["direct_dispatches"]), int(prof_all["hip_launches"])

    # ---- N > 1 under a launcher: the exchange mode that is not `value`, all ranks, right after the timed region ----
    other_exchange = None
    if exchange is not None and n_other > 0:
        other = "rccl" if args.exchange == "host" else "host"
        run_o = run_sync_rccl if other == "rccl" else run_sync_host
        try:  # a secondary must not cost the run its line (the RCCL variant has never run with more than one rank: no multi-GPU node so far)
            run_o(total, total + 16)
            cuda_sync()
            dist.barrier()
            t_o = time.perf_counter()
            run_o(total + 16, total + n_other)
            cuda_sync()
            dist.barrier()
            t_o = time.perf_counter() - t_o
            tt = torch.tensor([t_o], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            n_o = n_other - 16
            other_exchange = {"exchange": other, "evals_per_s": n_o / float(tt.item()), "ms_per_step": float(tt.item()) / n_o * 1e3,
                              "median_ms_per_step": float(np.median(per_step[total + 16:total + n_other])) * 1e3, "steps": n_o,
                              "collective_backend": dist.get_backend(exchange.group), "collective_world_size": dist.get_world_size(exchange.group),
                              "rccl_world_size": dist.get_world_size(exchange.group) if dist.get_backend(exchange.group) == "nccl" else None,
                              "note": ("hc_step_device (HIP launches on the rank's stream) + RCCL all-gather of the rows on the device + stream synchronise"
                                       if other == "rccl" else "hc_step on every rank + host gather through shared-memory result buffers")}
        except Exception as e:  # noqa: BLE001
            other_exchange = {"exchange": other, "error": str(e)}

    # ---- N > 1 under a launcher: the single-process C-ABI mode as a secondary, run by rank 0 while the others wait ----
    single_sec = None
    if world > 1 and strong and not args.no_secondary and not stub:
        dist.barrier()
        if rank == 0:
            try:
                devs = [0] * world if share_gpu else [g % ndev for g in range(world)]
                single_sec, _ = run_group_sync(N, world, devs, sdt, args.lookahead, 64, max(args.steps, 64), motion)
                single_sec["note"] = ("ONE process (this rank) drives all shards through hc_step_multi: state store per GPU, all step kernels "
                                      "dispatched before any wait, host-side gather -- no collective, no torch on the path")
            except Exception as e:  # a secondary must not cost the run its line
                single_sec = {"error": str(e)}
            torch.cuda.set_device(local_rank)  # the shard contexts made other devices current
        dist.barrier()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        timed = per_step[pre:total] * 1e3
        # dominant kernel: the look-ahead pass when blocking is on (one launch covers `lookahead` steps), else the per-step kernel.
        # `achieved` = the bytes a launch must move once / its mean duration over EVERY launch of the synchronous phases of this run
        # (warm-up, timed region, steady-state block: the same kernel with the same arguments); the launches of the timed region alone
        # are listed next to it.
        def pass_stats(pr):
            if pr["block_kernel_launches"] > 0:
                return pr["block_kernel_seconds"] / pr["block_kernel_launches"], int(pr["block_kernel_launches"])
            return pr["conv_kernel_seconds"] / max(1, pr["conv_kernel_launches"]), int(pr["conv_kernel_launches"])
        blocked = prof_all["block_kernel_launches"] > 0
        kname, units = ("hc::conv_block_kernel", int(args.lookahead)) if blocked else ("hc::conv_step_kernel", 1)
        conv_s, n_timed = pass_stats(prof_all)
        conv_s_region, n_region = pass_stats(prof)
        bytes_once = prof_all["block_kernel_bytes_once"] if blocked else prof_all["conv_kernel_bytes"]
        bytes_units = prof_all["block_kernel_bytes"] if blocked else prof_all["conv_kernel_bytes"]
        achieved = bytes_once / conv_s / 1e9 if conv_s > 0 else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")
        if os.path.exists(tpath) and N == N_BODIES and not strong:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f"block{units}_hbm_bytes_per_launch" if units > 1 else "hbm_bytes_per_launch")
                traffic_src = "profiles/conv_traffic.json (committed --pmc FETCH_SIZE x2 + WRITE_SIZE passes; not collected in this run)"
            except Exception:
                traffic = None
        us = lambda sec, n: 1e6 * sec / max(1, n)  # noqa: E731
        # ---- the WHOLE step against the memory (SURVEY 8d's per-step byte model is superseded: K leaves HBM once per `units` steps) ----
        step_fig = None
        if blocked and world == 1 and not strong:
            per_sample = 8.0 * D_local * 6 * N  # one IRF sample's block of K for this context's rows
            ratio = sdt / DT
            reach_mean = (units - 1) / 2.0 * ratio + (0.0 if abs(ratio - 1.0) < 1e-9 else 1.0)  # IRF samples a scatter launch streams, block average
            n_own = 1 if abs(ratio - 1.0) < 1e-9 else 2
            parts = {"pass_share": bytes_once / units, "scatter": per_sample * reach_mean, "own_samples_in_the_step_kernel": per_sample * n_own,
                     "terms_and_rows": 8.0 * (kterm_mean(units) + 4) * D_local}
            step_bytes = float(sum(parts.values()))
            mean_s, med_s = ms * 1e-3, float(np.median(timed)) * 1e-3
            pmc = None
            ppath = os.path.join(ROOT, "profiles", "r06", "pmc_step_traffic.json")
            if os.path.exists(ppath):
                try:
                    pmc = json.load(open(ppath)).get("hbm_bytes_per_steady_state_step")
                except Exception:  # noqa: BLE001
                    pmc = None
            step_fig = {"bound": "PCIe round trip + the step kernel's dependent chain (profiles/r06/step_stage_clock.txt), not bandwidth",
                        "step_traffic_bytes": step_bytes, "step_traffic_parts": parts,
                        "step_traffic_bytes_pmc": pmc, "step_traffic_pmc_source": "profiles/r06/pmc_step_traffic.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes)" if pmc else None,
                        "step_GBps_at_mean": step_bytes / mean_s / 1e9, "step_frac_of_hbm_peak_at_mean": step_bytes / mean_s / 1e9 / HBM_PEAK_GBS,
                        # the median step holds no pass launch: its own bytes (scatter + own samples + terms) over its own time
                        "median_step_traffic_bytes": step_bytes - parts["pass_share"],
                        "median_step_GBps": (step_bytes - parts["pass_share"]) / med_s / 1e9,
                        "median_step_frac_of_hbm_peak": (step_bytes - parts["pass_share"]) / med_s / 1e9 / HBM_PEAK_GBS,
                        "survey_8d_bytes_per_step": prof_all["conv_kernel_bytes"],
                        "reuse_over_survey_model": prof_all["conv_kernel_bytes"] / step_bytes,
                        "note": "algorithmic bytes of ONE steady-state step: the pass's K-once bytes / steps per pass + what the scatter behind the step streams "
                                "(block average) + the step kernel's own IRF samples + term slots; SURVEY 8d's model (all of K every step) is what "
                                "plain_per_step_mode runs"}
        out = {
            "metric": "hydro-force evals/sec (all bodies)",
            "value": (1 if strong else world) * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "alignment_steps": align,  # untimed, before the warm-up: puts a look-ahead block boundary in the middle of the timed region
            "passes_in_timed_region": int(prof["block_kernel_launches"]),
            "ms_per_step": ms,
            "median_ms_per_step": float(np.median(timed)),
            "p10_ms_per_step": float(np.percentile(timed, 10)),
            "p90_ms_per_step": float(np.percentile(timed, 90)),
            "higher_is_better": True,
            "scaling": args.scaling if (world > 1 or strong) else "n/a (C3 headline: one GPU, BASELINE.json's metric; the multi-GPU lines shard C4, see c4_one_gpu)",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "workload": "C4" if (strong and N == N_BODIES_C4) else ("C3" if N == N_BODIES else f"{N} bodies"),
            "config": {
                "workload": (f"C4-style: ONE coupled synthetic {N}-body array row-sharded over {world} GPU(s), " if strong else
                             f"C3: synthetic {N}-body array per GPU, ") +
                            f"{S_RIRF} radiation-IRF samples, irregular JONSWAP "
                            f"waves with {WAVES['nfrequencies']} components (excitation-IRF convolution, L={gpu.sizes()['L']}), "
                            f"prescribed motion, step dt = {sdt} s, dt_rirf = {DT} s, steady-state history; "
                            "synchronous steps (forces of step n are on the host before step n+1 is issued)",
                "bodies": N, "bodies_per_gpu": (N / world if strong else N), "irf_samples": S_RIRF,
                "wave_components": WAVES["nfrequencies"], "lookahead": args.lookahead,
                "sharding": (("body-row shards of one coupled array; every rank collects all rows on the host from shared-memory result buffers "
                              "(no collective)" if args.exchange == "host" else
                              "body-row shards of one coupled array + RCCL all-gather of forces every step") if strong else
                             "one independent farm per GPU, no data-path collective") if world > 1 else "single GPU",
            },
            "host_thread": host_thread,
            "caller": ("hc_step_many (the C ABI's own loop of K synchronous hc_step calls)"
                       if (world == 1 and exchange is None and not args.python_loop) else "one hc_step (or begin / gather / end) per Python call"),
            "dispatch_mode": dinfo["dispatch_mode"] if (world == 1 or args.exchange == "host") else "HIP launches on the rank's stream (hc_step_device) + RCCL all-gather",
            "exchange": (args.exchange if world > 1 else None),
            "pass_schedule_pinned_for_all_ranks": ({0: "at block start", 1: "one block ahead"}.get(pinned_schedule) if pinned_schedule is not None else None),
            "collective_backend": (dist.get_backend(exchange.group) if exchange is not None else None),
            "collective_world_size": (dist.get_world_size(exchange.group) if exchange is not None else None),
            "rccl_world_size": (dist.get_world_size(exchange.group) if (exchange is not None and dist.get_backend(exchange.group) == "nccl") else None),
            "dispatch_mode_reason": dinfo["dispatch_mode_reason"],
            "aql_dispatches": dinfo["aql_dispatches"], "hip_launches": dinfo["hip_launches"],
            "roofline": {
                "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_stream_ceiling": achieved / HBM_STREAM_CEILING_GBS, "stream_ceiling": HBM_STREAM_CEILING_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_once, "units_per_launch": units,
                "reuse_factor": bytes_units / bytes_once if bytes_once else None,
                "algorithmic_bytes_of_the_units": bytes_units,
                "profile_file": ROOFLINE_PROFILE if (N == N_BODIES and not strong) else None,
                "mean_kernel_us": conv_s * 1e6, "launches_timed": n_timed,
                "timed_region_kernel_us": conv_s_region * 1e6, "timed_region_launches": n_region,
                "step_kernel_us": us(prof_all["step_kernel_seconds"], prof_all["step_kernel_launches"]),
                "scatter_kernel_us": us(prof_all["scatter_kernel_seconds"], prof_all["scatter_kernel_launches"]),
                "fp64_TFLOPs": (2.0 * (bytes_units / 8.0) / conv_s / 1e12) if conv_s > 0 else None,
                "fp64_frac_of_mfma_peak": (2.0 * (bytes_units / 8.0) / conv_s / 1e12 / FP64_MFMA_PEAK_TF) if conv_s > 0 else None,
                "whole_step": step_fig,
                "note": (f"achieved = K-once bytes of a launch / mean duration of every pass of the run (completion signals); a launch serves {units} steps"
                         if units > 1 else "one launch = one step"),
            },
            "term_seconds": {k: prof[k] for k in ("hydrostatics_seconds", "radiation_seconds", "waves_seconds")},
        }
        if per_rank is not None:
            # one line per rank: which rank was slow, and in which kernel (the collective above is over the control plane, after the timed region)
            out["per_rank"] = per_rank
            own = [r["ms_per_step_own_loop"] for r in per_rank]
            out["per_rank_ms_per_step"] = {"min": min(own), "max": max(own), "slowest_rank": int(np.argmax(own)),
                                           "note": "each rank's own loop time / steps (before it waits for the others); `ms_per_step` is the max over ranks incl. the closing barrier"}
        if stub:
            out["stub_context"] = "tests/stub_context.StubShard: the multi-rank control flow on CPU over gloo, no GPU, no physics -- NOT a measurement"
        if steady is not None:
            out["steady_state"] = steady
        if python_loop is not None:
            out["python_loop"] = python_loop
        if chrono_like is not None:
            out["chrono_like_loop"] = chrono_like
        if pipelined is not None:
            out["device_pipelined"] = pipelined
        if plain is not None:
            out["plain_per_step_mode"] = plain
        if other_exchange is not None:
            out["other_exchange_mode"] = other_exchange
        if single_sec is not None:
            out["single_process_c_abi"] = single_sec
        if world == 1 and not strong and not args.no_secondary and not args.no_c4_share:
            try:
                gpu.close()  # the shard below gets the device to itself, like a rank of a multi-GPU run (a context that shares its device keeps its passes on the step path's lane)
                out["c4_rank_share"] = c4_rank_share(sdt, args.lookahead if args.lookahead > 0 else 32)
            except Exception as e:  # a secondary must not cost the run its line
                out["c4_rank_share"] = {"error": str(e)}
        if world == 1 and not strong and not args.no_secondary and not args.no_c4_one_gpu:
            try:
                gpu.close()
                out["c4_one_gpu"] = c4_one_gpu(sdt, args.lookahead if args.lookahead > 0 else 32)
            except Exception as e:  # a secondary must not cost the run its line
                out["c4_one_gpu"] = {"error": str(e)}
        if world == 1 and not strong and not args.no_secondary and not args.no_small_configs:
            # BASELINE.json's other single-GPU configs and the added-mass product, each beside its CPU figure (the device to themselves)
            gpu.close()
            for key, fn in (("c2_two_body", c2_two_body), ("c5_one_body_2048", c5_one_body_2048), ("added_mass_mv", added_mass_product)):
                try:
                    out[key] = fn()
                except Exception as e:  # a secondary must not cost the run its line
                    out[key] = {"error": str(e)}
        if world == 1 and not strong and not args.no_secondary and not args.no_init and case is not None and c3_wave_stats is not None:
            try:
                gpu.close()
                out["init"] = init_block(case, c3_wave_stats, sdt, duration)
                for key in ("c4_one_gpu", "c4_rank_share"):
                    if isinstance(out.get(key), dict) and "synth_fill" in out[key]:
                        out["init"].setdefault("synth_fill", {})[key] = out[key]["synth_fill"]
            except Exception as e:  # a secondary must not cost the run its line
                out["init"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline and case is not None:
            if affinity_before is not None:
                os.sched_setaffinity(0, affinity_before)  # the CPU oracle's threads on every core of the box, as before
            base, f_faithful, flat_threads = cpu_baseline(case, motion, t_hist, v_hist, args.cpu_seconds, sdt, duration)
            n_chk = k_next
            f_flat = oracle_all_steps(case, motion, t_hist, v_hist, sdt, duration, n_chk, flat_threads)
            nf = min(len(f_faithful), n_chk)
            out["cpu_baseline"] = base
            out["parity"] = {
                "timed_steps_max_rel_err": max_rel_err(forces[pre:total], f_flat[pre:total]),
                "all_steps_checked": n_chk,
                "all_steps_max_rel_err": max_rel_err(forces[:n_chk], f_flat[:n_chk]),
                "oracle": "flat-array variant of the CPU oracle on every step",
                "faithful_oracle_steps": nf,
                "faithful_oracle_max_rel_err": max_rel_err(forces[:nf], np.stack(f_faithful[:nf])),
                "tolerance": 1e-6,
            }
            out["parity_max_rel_err_vs_oracle"] = out["parity"]["all_steps_max_rel_err"]
            out["speedup_vs_cpu_baseline"] = out["value"] / base["value"]
            out["speedup_vs_optimized_cpu"] = out["value"] / base["optimized_port"]["value"]
        elif exchange is not None:
            out["gathered_rows_finite"] = bool(torch.isfinite(gathered[pre:total]).all().item())
            out["exchange_check"] = exchange_ok
        if strong and N == N_BODIES_C4:
            # the same coupled array on ONE GPU, so that a strong-scaling ratio can be formed from this line alone: the N = 1 line of
            # this benchmark is the C3 case (BASELINE.json's metric) and carries this workload as its `c4_one_gpu` secondary
            ref1 = c4_one_gpu_reference()
            if ref1 and world > 1:
                out["c4_one_gpu_reference"] = ref1
                out["speedup_vs_c4_one_gpu"] = out["value"] / ref1["evals_per_s"]
            if share_gpu:
                out["note"] = "HC_BENCH_SHARE_GPU=1: all ranks on ONE device, rows moved over gloo -- a functional run, not a scaling figure"
        print(json.dumps(out), flush=True)
    if hx is not None:
        dist.barrier()
        hx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
