#!/usr/bin/env python3
"""Headline benchmark of the hydro-force path: all-body force evaluations per second.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], "C3"): synthetic 64-body array, 1024 radiation-IRF samples, irregular JONSWAP
sea state with 512 wave components, prescribed body motion with dt = dt_rirf = 0.01 s, steady state (velocity history
pre-filled over the whole 10.23 s IRF window).  A "step" = one evaluation of all 6N hydrodynamic forces
(hydrostatic - radiation + waves) through hc_step_device, body states already resident in HBM.

Multi-GPU, default (weak scaling): every rank owns one independent 64-body farm (independent simulations, e.g. the
iterations of a design exploration); the farms share nothing, so there is no data-path collective -- only the
barriers around the timed region.  value = farms * K / max-over-ranks time.

--scaling strong --bodies 512 is configuration C4: ONE coupled 512-body array (K = 77.3 GB FP64, generated in HBM by
hc_synth_fill) row-sharded over the ranks (hydrochrono_amd.parallel.body_shard), forces all-gathered every step;
value = K / time.  It also runs on one GPU (77 GB fits in 288 GB).

The JSON line also carries
  roofline      HBM roofline of the convolution kernel: algorithmic bytes per launch / mean HIP-event duration
  cpu_baseline  the CPU oracle (reference-faithful OpenMP restatement, oracle/) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N=1 only); it is also used to check the GPU forces.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WAVES = dict(simulation_dt=0.01, simulation_duration=60.0, ramp_duration=0.0, wave_height=2.0, wave_period=8.0,
             frequency_min=0.02, frequency_max=0.5, nfrequencies=512, peak_enhancement_factor=3.3, seed=1)
N_BODIES, S_RIRF, N_EXC, DT = 64, 1024, 1024, 0.01
T0 = 20.0  # start of the timed window (history covers [T0 - 10.29, T0))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)  # 2000 steps = 125 look-ahead blocks, about 50 ms of GPU time
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--bodies", type=int, default=N_BODIES, help="bodies per GPU (default: the C3 configuration)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--profile-stride", type=int, default=17, help="HIP-event sampling stride for the per-step launches (co-prime "
                    "with the 16-step look-ahead period); every look-ahead pass, the roofline kernel, is timed regardless")
    ap.add_argument("--lookahead", type=int, default=16, help="0: plain per-step evaluation (K streamed every step)")
    ap.add_argument("--step-dt", type=float, default=DT, help="caller's step size (default = the IRF grid spacing, the common "
                    "case; e.g. 0.007 makes every IRF sample a true interpolation, SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline time budget")
    return ap.parse_args()


def cpu_baseline(case, motion, t_hist, v_hist, budget_s, step_dt=DT, duration=60.0):
    """Times the CPU oracle on this box's host cores on a bounded sample of the same workload.

    Two variants: the reference-faithful restatement (OpenMP over IRF steps, per-element
    accessor, nested-vector history -- `value`, thread count chosen by a short sweep because the reference's
    `schedule(static)` over 1024 steps does not scale to every core count) and an optimised flat-array CPU variant
    (`optimized_port`) so that the GPU speed-up is not flattered by reference overheads.  Returns (dict, forces at T0)."""
    import oracle as orc_mod
    from cases import load_into_oracle
    cores = os.cpu_count() or 1
    orc = load_into_oracle(case)
    orc.add_waves_irregular(**dict(WAVES, simulation_dt=step_dt, simulation_duration=duration))
    orc.prefill_history(t_hist, v_hist)
    k = [0]

    def timed(fn, nsteps):
        d, f0 = [], None
        for _ in range(nsteps):
            t = T0 + k[0] * step_dt
            st = motion.state(t)
            a = time.perf_counter()
            f = fn(t, *st)
            d.append(time.perf_counter() - a)
            f0 = f if f0 is None else f0
            k[0] += 1
        return d, f0

    t_begin = time.perf_counter()
    sweep, first = {}, None
    for th in sorted({cores, max(1, cores // 2), min(cores, 64), min(cores, 16)}, reverse=True):
        orc_mod.set_num_threads(th)
        d, f0 = timed(orc.step, 3)
        first = f0 if first is None else first
        sweep[th] = float(np.median(d))
    best = min(sweep, key=sweep.get)
    orc_mod.set_num_threads(best)
    n_more = int(max(3, min(40, (0.5 * budget_s - (time.perf_counter() - t_begin)) / sweep[best])))
    d, _ = timed(orc.step, n_more)
    med = float(np.median(d))
    flat_sweep = {}
    for th in sorted({cores, min(cores, 64), min(cores, 16)}, reverse=True):
        orc_mod.set_num_threads(th)
        orc.flat_prepare()  # re-laid out (first touch) with this thread count
        d_flat, _ = timed(orc.flat_step, 8)
        flat_sweep[th] = float(np.median(d_flat[2:]))
    best_flat = min(flat_sweep, key=flat_sweep.get)
    med_flat = flat_sweep[best_flat]
    info = {"value": 1.0 / med, "unit": "evals/s", "cores": best, "kind": "port", "ms_per_step": med * 1e3,
            "sample": f"{n_more} consecutive steady-state steps of the same workload (median), reference-faithful oracle "
                      f"-O2 -fopenmp, OMP threads = {best} (best of sweep); box has {cores} logical cores",
            "threads_sweep_ms": {str(th): v * 1e3 for th, v in sweep.items()},
            "optimized_port": {"value": 1.0 / med_flat, "unit": "evals/s", "cores": best_flat, "ms_per_step": med_flat * 1e3,
                               "threads_sweep_ms": {str(th): v * 1e3 for th, v in flat_sweep.items()},
                               "sample": "6 steps per thread count (median), flat-array OpenMP-over-rows CPU variant of the same math"}}
    return info, first


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    from hydrochrono_amd.hydro import HydroForces
    from hydrochrono_amd.mock_chrono import PrescribedMotion
    from hydrochrono_amd.synthetic import many_body_case, rest_positions

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hydro-force path has no CPU fallback")
    # HC_BENCH_SHARE_GPU=1 (functional test of the N > 1 code path on a one-GPU box): all ranks use device 0 and the
    # force all-gather goes over gloo instead of RCCL.  Never used for reported numbers.
    share_gpu = os.environ.get("HC_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from hydrochrono_amd.parallel import ForceExchange, body_shard
    strong = args.scaling == "strong"
    N = args.bodies
    D = 6 * N
    case = None
    if strong:
        # one coupled N-body array, this rank owns the output rows of bodies [b0, b1); inputs generated in HBM
        b0, b1 = body_shard(N, world, rank)
        gpu = HydroForces(N, device=local_rank, body_range=(b0, b1))
        gpu.synth_fill(20251031, S_RIRF, DT, N_EXC, DT)
        gpu.finalize()
        motion = PrescribedMotion(N, np.zeros((N, 3)), seed=20251031)
        exchange = ForceExchange(N, world, rank, device="cuda")
    else:
        case = many_body_case(N, S=S_RIRF, dt_rirf=DT, n_exc=N_EXC, dt_exc=DT, seed=20251031 + rank)
        gpu = HydroForces.from_case(case, device=local_rank)
        motion = PrescribedMotion(N, rest_positions(case), seed=20251031 + rank)
        exchange = None  # independent farms: nothing to exchange
    # the wave model is built for the caller's step size; the free-surface table must cover every step of this run
    # (timed + warm-up + the plain-mode secondary measurement), so long runs extend the 60 s of the C3 definition
    n_all = args.warmup + args.steps + max(20, args.steps // 4) + 8
    duration = max(WAVES["simulation_duration"], T0 + n_all * args.step_dt + 5.0)
    waves = dict(WAVES, num_bodies=N, simulation_dt=args.step_dt, simulation_duration=duration)
    gpu.add_waves_irregular(**waves)
    gpu.set_lookahead(args.lookahead)
    D_local = gpu.D_local

    sdt = args.step_dt
    nhist = int(np.ceil(S_RIRF * DT / sdt)) + 5
    t_hist = T0 - sdt * np.arange(1, nhist + 1)
    v_hist = np.stack([motion.velocity6(t) for t in t_hist])
    gpu.set_history(t_hist, v_hist)

    total = args.warmup + args.steps
    states = torch.tensor(np.stack([motion.packed(T0 + k * sdt) for k in range(total)]), device="cuda")
    forces = torch.zeros(total, D_local, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    # device addresses of each step's state / force row, so that the timed loop is the C-ABI call and nothing else
    state_ptrs = [states.data_ptr() + k * states.stride(0) * 8 for k in range(total)]
    force_ptrs = [forces.data_ptr() + k * forces.stride(0) * 8 for k in range(total)]
    times = [T0 + k * sdt for k in range(total)]
    step_device = gpu.step_device

    def run(k0, k1):
        if exchange is not None and world > 1:
            # coupled array: kernels write straight into the exchange's send buffer; the RCCL all-gather leaves the
            # full 6N force vector on every rank (the one exchange step of the path)
            send_ptr = exchange.send.data_ptr()
            for k in range(k0, k1):
                step_device(times[k], state_ptrs[k], send_ptr, stream)
                exchange.gather()
        else:
            for k in range(k0, k1):
                step_device(times[k], state_ptrs[k], force_ptrs[k], stream)

    run(0, args.warmup)
    torch.cuda.synchronize()
    gpu.enable_profiling(args.profile_stride)  # HIP events around the conv kernel of every n-th timed step
    gpu.reset_profile()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    run(args.warmup, total)
    enqueue_s = time.perf_counter() - t_start  # host time to enqueue every step (the GPU runs behind it)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    prof = gpu.profile()
    gpu.enable_profiling(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Secondary figure (not `value`): the same workload with look-ahead off, i.e. K streamed from HBM every step -- the
    # conventional form of the path, whose kernel is the one the per-step HBM roofline applies to.
    plain = None
    if args.lookahead > 0 and world == 1:
        n_plain = max(20, args.steps // 4)
        gpu.set_lookahead(0)
        extra_states = torch.tensor(np.stack([motion.packed(T0 + (total + k) * sdt) for k in range(n_plain + 4)]), device="cuda")
        extra_out = torch.zeros(n_plain + 4, D_local, dtype=torch.float64, device="cuda")
        for k in range(4):
            gpu.step_device(T0 + (total + k) * sdt, extra_states[k].data_ptr(), extra_out[k].data_ptr(), stream)
        torch.cuda.synchronize()
        gpu.enable_profiling(args.profile_stride)
        gpu.reset_profile()
        tp = time.perf_counter()
        for k in range(4, n_plain + 4):
            gpu.step_device(T0 + (total + k) * sdt, extra_states[k].data_ptr(), extra_out[k].data_ptr(), stream)
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp
        pp = gpu.profile()
        gpu.enable_profiling(False)
        kus = 1e6 * pp["conv_kernel_seconds"] / max(1, pp["conv_kernel_launches"])
        plain = {"evals_per_s": n_plain / tp, "ms_per_step": tp / n_plain * 1e3, "steps": n_plain, "kernel": "hc::conv_step_kernel",
                 "mean_kernel_us": kus, "achieved_GBps": pp["conv_kernel_bytes"] / (kus * 1e-6) / 1e9 if kus > 0 else 0.0,
                 "frac_of_hbm_peak": (pp["conv_kernel_bytes"] / (kus * 1e-6) / 1e9 / HBM_PEAK_GBS) if kus > 0 else 0.0}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        # dominant kernel: the look-ahead pass when blocking is on (one launch covers 16 steps), else the per-step kernel
        if prof["block_kernel_launches"] > 0:
            kname, steps_per_launch = "hc::conv_block_kernel", 16
            conv_s = prof["block_kernel_seconds"] / prof["block_kernel_launches"]
            alg_bytes = prof["block_kernel_bytes"]
            n_timed = prof["block_kernel_launches"]
        else:
            kname, steps_per_launch = "hc::conv_step_kernel", 1
            conv_s = prof["conv_kernel_seconds"] / max(1, prof["conv_kernel_launches"])
            alg_bytes = prof["conv_kernel_bytes"]
            n_timed = prof["conv_kernel_launches"]
        achieved = alg_bytes / conv_s / 1e9 if conv_s > 0 else 0.0
        rem_us = 1e6 * prof["rem_kernel_seconds"] / max(1, prof["rem_kernel_launches"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")
        if os.path.exists(tpath) and N == N_BODIES:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("block_hbm_bytes_per_launch" if steps_per_launch == 16 else "hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "hydro-force evals/sec (all bodies)",
            "value": (1 if strong else world) * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            # wall time of the enqueue loop; in long runs the host runs ahead until the launch queue is full and then waits
            # on the GPU, so this approaches ms_per_step -- the host's own cost per step is about 0.007 ms (short runs)
            "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"C4-style: ONE coupled synthetic {N}-body array row-sharded over {world} GPU(s), " if strong else
                             f"C3: synthetic {N}-body array per GPU, ") +
                            f"{S_RIRF} radiation-IRF samples, irregular JONSWAP "
                            f"waves with {WAVES['nfrequencies']} components (excitation-IRF convolution, L={gpu.sizes()['L']}), "
                            f"prescribed motion, step dt = {sdt} s, dt_rirf = {DT} s, steady-state history",
                "bodies": N, "bodies_per_gpu": (N / world if strong else N), "irf_samples": S_RIRF,
                "wave_components": WAVES["nfrequencies"],
                "sharding": ("body-row shards of one coupled array + RCCL all-gather of forces" if strong else
                             "one independent farm per GPU, no data-path collective") if world > 1 else "single GPU",
            },
            "roofline": {
                "bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "steps_per_launch": steps_per_launch,
                "mean_kernel_us": conv_s * 1e6, "launches_timed": n_timed,
                "in_block_step_kernel_us": rem_us,
                # what the launch really moves / computes: PMC-measured HBM bytes over the live duration, and the FP64 MFMA
                # rate of the [D x F] x [F x 16] product against the 78.6 TFLOP/s dense FP64 peak
                "traffic_GBps": (traffic / conv_s / 1e9) if (traffic and conv_s > 0) else None,
                "traffic_frac_of_hbm_peak": (traffic / conv_s / 1e9 / HBM_PEAK_GBS) if (traffic and conv_s > 0) else None,
                "fp64_TFLOPs": (2.0 * (alg_bytes / 8.0) / conv_s / 1e12) if conv_s > 0 else None,
                "fp64_frac_of_mfma_peak": (2.0 * (alg_bytes / 8.0) / conv_s / 1e12 / 78.6) if conv_s > 0 else None,
                "note": ("one look-ahead launch covers 16 steps: algorithmic bytes = the SURVEY 8d per-step figure summed over those "
                         "steps (less the newest-sample share each step adds itself), while K leaves HBM once (see traffic), so "
                         "frac > 1 measures the reuse, not a faster memory")
                        if steps_per_launch == 16 else "one launch = one step",
            },
        }
        if plain is not None:
            out["plain_per_step_mode"] = plain
        if world == 1 and not args.no_cpu_baseline and case is not None:
            base, f_cpu = cpu_baseline(case, motion, t_hist, v_hist, args.cpu_seconds, sdt, duration)
            f_gpu = forces[0].cpu().numpy()  # step k = 0 is t = T0 on both sides
            out["cpu_baseline"] = base
            out["parity_max_rel_err_vs_oracle"] = float(np.max(np.abs(f_gpu - f_cpu)) / np.max(np.abs(f_cpu)))
            out["speedup_vs_cpu_baseline"] = out["value"] / base["value"]
            out["speedup_vs_optimized_cpu"] = out["value"] / base["optimized_port"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
