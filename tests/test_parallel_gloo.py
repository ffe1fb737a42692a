"""N > 1 path on CPU: body-row sharding + the force all-gather, world_size 2 over gloo (127.0.0.1 rendezvous)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hydrochrono_amd.parallel import ForceExchange, body_shard


def test_body_shard_partition():
    for N in (2, 5, 64, 512, 513):
        for world in (1, 2, 3, 8):
            if world > N:
                with pytest.raises(ValueError):
                    body_shard(N, world, 0)
                continue
            shards = [body_shard(N, world, r) for r in range(world)]
            assert shards[0][0] == 0 and shards[-1][1] == N
            assert all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
            sizes = [b - a for a, b in shards]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, steps, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = 6 * N
        rng = np.random.default_rng(7)  # same stream on every rank: the "full state every rank receives"
        A = rng.normal(size=(D, D))
        b0, b1 = body_shard(N, world, rank)
        ex = ForceExchange(N, world, rank, device="cpu")
        ok = True
        for _ in range(steps):
            x = rng.normal(size=D)
            local = torch.from_numpy(A[6 * b0:6 * b1] @ x)  # this rank's force rows
            full = ex.gather(local).numpy()
            ok &= bool(np.array_equal(full, np.concatenate([A[6 * s0:6 * s1] @ x for s0, s1 in ex.shards])))
            ok &= full.shape == (D,)
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [4, 5])  # even and uneven shards
def test_force_all_gather_world2_gloo(N):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), N, 5, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


# ------------------------------------------------------------------------------------------------
# real contexts: two processes, each with a row-sharded hc_ctx (hc_create_sharded) on the one GPU of the box, forces
# all-gathered over gloo (on an 8-GPU node the same code runs one rank per GPU over RCCL, bench.py --gpus N)
# ------------------------------------------------------------------------------------------------
def _ctx_worker(rank, world, port, N, steps, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HC_DEVICE_SHARED="1")  # the ranks share the one GPU of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        from cases import load_into_oracle
        from hydrochrono_amd.hydro import HydroForces
        from hydrochrono_amd.mock_chrono import PrescribedMotion
        from hydrochrono_amd.synthetic import many_body_case, rest_positions
        case = many_body_case(N, S=96, dt_rirf=0.01, n_exc=65, dt_exc=0.02, seed=4100 + N)
        kw = dict(simulation_dt=0.01, simulation_duration=4.0, ramp_duration=0.5, wave_height=2.0, wave_period=6.0,
                  frequency_min=0.05, frequency_max=0.6, nfrequencies=48, peak_enhancement_factor=3.3)
        b0, b1 = body_shard(N, world, rank)
        mine = HydroForces.from_case(case, device=0, body_range=(b0, b1))
        mine.add_waves_irregular(**kw)
        ex = ForceExchange(N, world, rank, device="cpu")
        motion = PrescribedMotion(N, rest_positions(case), seed=5)
        full = orc = None
        if rank == 0:  # the unsharded context and the CPU oracle as checkers
            full = HydroForces.from_case(case, device=0)
            full.add_waves_irregular(**kw)
            orc = load_into_oracle(case)
            orc.add_waves_irregular(**kw)
        ok, worst = True, 0.0
        for n in range(steps):
            t = 0.01 * n
            st = motion.state(t)
            gathered = ex.gather(torch.from_numpy(mine.step(t, *st))).numpy().copy()
            ok &= gathered.shape == (6 * N,)
            if rank == 0:
                ok &= bool(np.array_equal(gathered, full.step(t, *st)))  # row shards add in the same order: bitwise
                fo = orc.step(t, *st)
                worst = max(worst, float(np.max(np.abs(gathered - fo)) / np.max(np.abs(fo))))
        ret[rank] = (bool(ok), worst)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [6, 5])  # even and uneven shards
def test_row_sharded_contexts_two_ranks_one_gpu(N):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ctx_worker, args=(world, _free_port(), N, 70, ret), nprocs=world, join=True)
    res = dict(ret)
    assert res[0][0] and res[1][0]
    assert res[0][1] <= 1e-10  # gathered forces vs the CPU oracle (contract 1e-6)


# ------------------------------------------------------------------------------------------------
# RCCL on the one GPU of the box: a world of one rank, the device path of bench.py --gpus N (hc_step_device writes the
# rank's force rows into the exchange buffer, the all-gather follows on the same stream).  With one rank the collective
# moves nothing between GPUs, but the communicator, the device buffers and the stream ordering are the real ones.
# ------------------------------------------------------------------------------------------------
def _rccl_worker(rank, world, port, N, steps, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from hydrochrono_amd.hydro import HydroForces
        from hydrochrono_amd.mock_chrono import PrescribedMotion
        from hydrochrono_amd.synthetic import many_body_case, rest_positions
        case = many_body_case(N, S=64, dt_rirf=0.01, n_exc=33, dt_exc=0.02, seed=4200)
        dev, host = HydroForces.from_case(case, device=0), HydroForces.from_case(case, device=0)
        ex = ForceExchange(N, world, rank, device="cuda")
        motion = PrescribedMotion(N, rest_positions(case), seed=9)
        states = torch.tensor(np.stack([motion.packed(0.01 * n) for n in range(steps)]), device="cuda")
        out = torch.zeros(steps, 6 * N, dtype=torch.float64, device="cuda")
        stream = torch.cuda.Stream()  # an explicit stream: handle 0 would mean the context's own stream, see step_device
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for n in range(steps):
                dev.step_device(0.01 * n, states[n].data_ptr(), ex.send.data_ptr(), stream.cuda_stream)
                dist.all_gather_into_tensor(ex.recv, ex.send)  # what gather() issues for world > 1
                out[n].copy_(ex.recv[: 6 * N])
        torch.cuda.synchronize()
        want = np.stack([host.step(0.01 * n, *motion.state(0.01 * n)) for n in range(steps)])
        got = out.cpu().numpy()
        ret[rank] = (bool(np.array_equal(got, want)), dist.get_backend(), float(np.max(np.abs(got - want))))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_device_path_all_gather_over_rccl_one_rank():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(1, _free_port(), 3, 80, ret), nprocs=1, join=True)
    assert dict(ret) == {0: (True, "nccl", 0.0)}


# ------------------------------------------------------------------------------------------------
# host gather through shared-memory result buffers (hc_set_result_buffer / hc_wait_result_buffer): two processes, each with a
# row-sharded context on the one GPU of the box; every process collects both shards' rows straight from the buffers the step
# kernels write -- no collective on the data path (gloo only for the barrier and the final verdict)
# ------------------------------------------------------------------------------------------------
def _shm_worker(rank, world, port, N, steps, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HC_DEVICE_SHARED="1")  # the ranks share the one GPU of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        from cases import load_into_oracle
        from hydrochrono_amd import capi
        from hydrochrono_amd.host_exchange import HostExchange
        from hydrochrono_amd.hydro import HydroForces
        from hydrochrono_amd.mock_chrono import PrescribedMotion
        from hydrochrono_amd.synthetic import many_body_case, rest_positions
        case = many_body_case(N, S=96, dt_rirf=0.01, n_exc=65, dt_exc=0.02, seed=4300 + N)
        kw = dict(simulation_dt=0.01, simulation_duration=4.0, ramp_duration=0.5, wave_height=2.0, wave_period=6.0,
                  frequency_min=0.05, frequency_max=0.6, nfrequencies=48, peak_enhancement_factor=3.3)
        b0, b1 = body_shard(N, world, rank)
        mine = HydroForces.from_case(case, device=0, body_range=(b0, b1))
        mine.add_waves_irregular(**kw)
        ex = HostExchange(mine, N, world, rank, tag=f"hc_test_{port}")
        dist.barrier()
        ex.attach()
        motion = PrescribedMotion(N, rest_positions(case), seed=5)
        full = orc = None
        if rank == 0:
            full = HydroForces.from_case(case, device=0)
            full.add_waves_irregular(**kw)
            orc = load_into_oracle(case)
            orc.add_waves_irregular(**kw)
        lib = capi.load()
        dp = lambda x: x.ctypes.data_as(capi.c_double_p)  # noqa: E731
        own = np.empty(6 * (b1 - b0))
        ok, worst = True, 0.0
        for n in range(steps):
            t = 0.01 * n
            st = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in motion.state(t)]
            ok &= lib.hc_step_begin(mine.ctx, C.c_double(t), *[dp(x) for x in st]) == 0
            gathered = ex.gather(ex.sequence())      # both shards' rows, read where the GPUs wrote them
            ok &= lib.hc_step_end(mine.ctx, dp(own)) == 0
            ok &= bool(np.array_equal(gathered[6 * b0:6 * b1], own))
            if rank == 0:
                ok &= bool(np.array_equal(gathered, full.step(t, *st)))
                fo = orc.step(t, *st)
                worst = max(worst, float(np.max(np.abs(gathered - fo)) / np.max(np.abs(fo))))
            if n % 16 == 3 and rank == 1:
                import time
                time.sleep(0.002)  # a slow rank: the others are a step ahead at most (the buffer's two halves)
        dist.barrier()
        ex.close()
        ret[rank] = (bool(ok), worst)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [6, 5])  # even and uneven shards
def test_host_gather_through_shared_memory_two_ranks_one_gpu(N):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shm_worker, args=(world, _free_port(), N, 150, ret), nprocs=world, join=True)
    res = dict(ret)
    assert res[0][0] and res[1][0]
    assert res[0][1] <= 1e-10


# ------------------------------------------------------------------------------------------------
# The device path of `bench.py --exchange rccl` with TWO ranks.  RCCL refuses two ranks on one device, and the box has one, so the
# collective itself runs over gloo here; everything in front of it is the real thing: hc_step_device of a row-sharded context on a
# CALLER'S stream, the consumer of the rows (what the all-gather is on an 8-GPU node) enqueued on the SAME stream right behind the
# step kernel, and the next step enqueued BEFORE the host has waited for this one -- the order of bench.py's run_sync_rccl plus one
# step of run-ahead, which is what would expose a step kernel, a scatter or a pass (they move to the context's own stream behind an
# event, hc_step.cpp: enqueue_tail / to_background) that reads a ring slot or a result row another stream still writes.
# RCCL with more than one rank has not run anywhere yet.
# ------------------------------------------------------------------------------------------------
def _dev_order_worker(rank, world, port, N, steps, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HC_DEVICE_SHARED="1")  # the ranks share the one GPU of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:
            sys.path.insert(0, here)
        from cases import load_into_oracle
        from hydrochrono_amd.hydro import HydroForces
        from hydrochrono_amd.mock_chrono import PrescribedMotion
        from hydrochrono_amd.synthetic import many_body_case, rest_positions
        torch.cuda.set_device(0)
        case = many_body_case(N, S=96, dt_rirf=0.01, n_exc=65, dt_exc=0.02, seed=4400 + N)
        kw = dict(simulation_dt=0.01, simulation_duration=4.0, ramp_duration=0.5, wave_height=2.0, wave_period=6.0,
                  frequency_min=0.05, frequency_max=0.6, nfrequencies=48, peak_enhancement_factor=3.3)
        b0, b1 = body_shard(N, world, rank)
        mine = HydroForces.from_case(case, device=0, body_range=(b0, b1))
        mine.add_waves_irregular(**kw)
        ex = ForceExchange(N, world, rank, device="cpu")
        motion = PrescribedMotion(N, rest_positions(case), seed=5)
        states = torch.tensor(np.stack([motion.packed(0.01 * n) for n in range(steps)]), device="cuda")
        rows = torch.zeros(steps, ex.max_rows, dtype=torch.float64, device="cuda")       # where the step kernels write
        landed = torch.zeros(steps, ex.max_rows, dtype=torch.float64).pin_memory()      # where the stream-ordered consumer leaves them
        events = [torch.cuda.Event() for _ in range(steps)]
        stream = torch.cuda.Stream()
        torch.cuda.synchronize()

        def enqueue(n):
            mine.step_device(0.01 * n, states[n].data_ptr(), rows[n].data_ptr(), stream.cuda_stream)
            with torch.cuda.stream(stream):
                landed[n].copy_(rows[n], non_blocking=True)  # same stream, right behind the step kernel: the all-gather's place
                events[n].record(stream)

        full = orc = None
        if rank == 0:
            full = HydroForces.from_case(case, device=0)
            full.add_waves_irregular(**kw)
            orc = load_into_oracle(case)
            orc.add_waves_irregular(**kw)
        ok, worst = True, 0.0
        enqueue(0)
        for n in range(steps):
            if n + 1 < steps:
                enqueue(n + 1)          # one step of run-ahead: step n + 1 is in flight while step n's rows are collected
            events[n].synchronize()
            gathered = ex.gather(landed[n, : ex.rows]).numpy().copy()
            if rank == 0:
                st = motion.state(0.01 * n)
                ok &= bool(np.array_equal(gathered, full.step(0.01 * n, *st)))
                fo = orc.step(0.01 * n, *st)
                worst = max(worst, float(np.max(np.abs(gathered - fo)) / np.max(np.abs(fo))))
        p = mine.profile()
        ok &= p["hip_launches"] > steps and p["direct_dispatches"] == 0  # the device path: HIP launches on the caller's stream
        ret[rank] = (bool(ok), worst)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [6, 5])  # even and uneven shards
def test_device_path_ordering_two_ranks_one_gpu(N):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dev_order_worker, args=(world, _free_port(), N, 140, ret), nprocs=world, join=True)
    res = dict(ret)
    assert res[0][0] and res[1][0]
    assert res[0][1] <= 1e-10


# ------------------------------------------------------------------------------------------------
# bench.py --gpus 8 under torch.distributed.run has never met eight ranks (the pool's boxes have one GPU): its whole N > 1 control flow
# runs here on CPU over gloo with a stub context (tests/stub_context.py: no physics) -- rendezvous, uneven body-row shards, the
# shared-memory result buffers and their tagged granules, host gather, the device-path all-gather as the other exchange mode,
# the after-run exchange check, the per-rank diagnostics and the JSON line -- so that argument and buffer-shape bugs are found here.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("exchange,bodies", [("host", 44), ("rccl", 40)])
def test_bench_multi_rank_control_flow_world8_stub(exchange, bodies):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    world = 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "24", "--warmup", "5",
           "--stub-context", "--bodies", str(bodies), "--exchange", exchange]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 24 and d["scaling"] == "strong" and d["exchange"] == exchange and "stub_context" in d
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert d["exchange_check"] == {"own_rows_bitwise_on_every_rank": True, "all_ranks_hold_the_same_vectors": True, "steps_checked": d["exchange_check"]["steps_checked"]}
    # every rank reported, with the rows of ITS shard (44 bodies: four shards of 6 and four of 5 bodies)
    from hydrochrono_amd.parallel_split import body_shard
    assert [p["rank"] for p in d["per_rank"]] == list(range(world))
    assert [p["rows"] for p in d["per_rank"]] == [6 * (b1 - b0) for b0, b1 in (body_shard(bodies, world, r) for r in range(world))]
    for p in d["per_rank"]:
        assert set(p["kernel_us_per_step"]) >= {"pass", "short_passes", "scatter", "step_kernels"} and p["ms_per_step_own_loop"] > 0
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"] and 0 <= d["per_rank_ms_per_step"]["slowest_rank"] < world
    # the exchange mode that is not `value` ran on all ranks too, and says how many ranks its collective spanned
    o = d["other_exchange_mode"]
    assert "error" not in o, o
    assert o["exchange"] == ("rccl" if exchange == "host" else "host") and o["collective_world_size"] == world and o["collective_backend"] == "gloo"
    assert o["rccl_world_size"] is None  # (gloo here; on GPUs this is the RCCL communicator's size)


def test_bench_no_launcher_gpus8_line_is_first_contact_proof_stub():
    """`python bench.py --gpus 8` WITHOUT a launcher -- what a driver may run on an 8-GPU node -- is one process (hc_step_multi over eight
    shard contexts, host gather).  Its line must stand on its own at first contact: the timed region aligned to a look-ahead block
    boundary, `roofline` and `per_shard` kernel splits, `passes_in_timed_region`, an `exchange_check` (rows of hc_step_multi against each
    shard's own hc_step), `cpu_baseline: null` with the reason -- and `rccl_ranks`: the launcher form started as a CHILD process before
    this one touches a GPU (one rank per GPU, RCCL all-gather of the rows every step; gloo here), so that the one command also says
    whether RCCL saw eight ranks.  Rehearsed on CPU with the stub context (no physics)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    world, bodies = 8, 44
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "24", "--warmup", "5", "--stub-context", "--bodies", str(bodies)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(env, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 24 and d["warmup"] == 5 and d["scaling"] == "strong" and d["value"] > 0
    # the timed region: 5 warm-up steps behind an alignment stretch that puts a block boundary into its middle
    assert (d["alignment_steps"] + 5 + 12) % 32 == 0 and "passes_in_timed_region" in d
    from hydrochrono_amd.parallel_split import body_shard
    assert [p["rows"] for p in d["per_shard"]] == [6 * (b1 - b0) for b0, b1 in (body_shard(bodies, world, g) for g in range(world))]
    for p in d["per_shard"]:
        assert set(p["kernel_us_per_step"]) >= {"pass", "short_passes", "scatter", "step_kernels"} and "pass_frac_of_hbm_peak" in p
    assert d["roofline"]["bound"] == "hbm" and {"frac_max_over_shards", "frac_min_over_shards", "units_per_launch"} <= set(d["roofline"])
    ec = d["exchange_check"]
    assert ec["rows_of_hc_step_multi_equal_each_shards_own_hc_step"] is True and ec["steps_checked"] > 0 and ec["finite"] is True
    assert d["cpu_baseline"] is None and "C4" in d["cpu_baseline_note"]
    rr = d["rccl_ranks"]
    assert "error" not in rr, rr
    assert rr["n_gpus"] == world and rr["exchange"] == "rccl" and rr["collective_world_size"] == world and rr["collective_backend"] == "gloo"
    assert rr["exchange_check"]["own_rows_bitwise_on_every_rank"] is True and rr["exchange_check"]["all_ranks_hold_the_same_vectors"] is True
    assert [p["rank"] for p in rr["per_rank"]] == list(range(world)) and rr["ms_per_step"] > 0
    assert rr["pass_schedule_pinned_for_all_ranks"] in ("at block start", "one block ahead")
