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
