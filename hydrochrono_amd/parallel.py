"""Multi-GPU plumbing of the hydro-force path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The path shards by OUTPUT BODY ROWS: rank r owns bodies [b0, b1) -> its slice of the radiation kernel
K[6*(b1-b0)][S*6N], of the excitation IRF, hydrostatics and added-mass rows.  Every rank receives the full body
state each step (the host integrator already holds it), so the only exchange is an all-gather of the per-rank force
rows (6*(b1-b0) doubles, tens of KB at most): latency-bound, no reduction, deterministic.
"""
import torch
import torch.distributed as dist


from .parallel_split import body_shard  # noqa: F401  (re-exported)


class ForceExchange:
    """All-gathers row-sharded force vectors into the full 6N vector on every rank.

    `send` is the buffer kernels should write into (6*max_local doubles, first 6*n_local valid); `gather()` returns
    the assembled [6N] tensor in global body order."""

    def __init__(self, num_bodies, world, rank, device, group=None, dtype=torch.float64):
        self.N, self.world, self.rank, self.group = num_bodies, world, rank, group
        self.shards = [body_shard(num_bodies, world, r) for r in range(world)]
        self.max_rows = 6 * max(b1 - b0 for b0, b1 in self.shards)
        self.rows = 6 * (self.shards[rank][1] - self.shards[rank][0])
        self.send = torch.zeros(self.max_rows, dtype=dtype, device=device)
        self.recv = torch.zeros(world * self.max_rows, dtype=dtype, device=device)
        self.even = all(6 * (b1 - b0) == self.max_rows for b0, b1 in self.shards)
        if not self.even:
            idx = []
            for r, (b0, b1) in enumerate(self.shards):
                idx.extend(range(r * self.max_rows, r * self.max_rows + 6 * (b1 - b0)))
            self.index = torch.tensor(idx, dtype=torch.long, device=device)

    def gather(self, local=None):
        if local is not None:
            self.send[: self.rows].copy_(local)
        if self.world == 1:
            return self.send[: self.rows]
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.recv if self.even else self.recv.index_select(0, self.index)
