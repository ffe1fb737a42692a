"""Host gather of row-sharded force vectors between the processes of ONE node without a collective (tests / bench only).

Every process owns one row shard (hc_create_sharded) and points its context's result buffer (hc_set_result_buffer) at a shared
file under /dev/shm that all processes map: the step kernel of each GPU writes its tagged {value, step sequence} granules there, and
every process collects every shard's rows straight from those buffers (hc_wait_result_buffer) -- SURVEY.md 8e's "outputs -> host
gather", for an MPI-style host.  No torch, no RCCL, no copy.  The C entry points do the work; this module only creates the segments."""
import ctypes as C
import mmap
import os

import numpy as np

from . import capi
from .parallel_split import body_shard

PAGE = 4096


class HostExchange:
    def __init__(self, gpu, num_bodies, world, rank, tag):
        """gpu: this process's HydroForces shard (rows of body_shard(num_bodies, world, rank)); tag: a name all ranks agree on."""
        self.lib, self.gpu, self.N, self.world, self.rank = capi.load(), gpu, num_bodies, world, rank
        self.shards = [body_shard(num_bodies, world, r) for r in range(world)]
        self.rows = [6 * (b1 - b0) for b0, b1 in self.shards]
        self.sizes = [-(-(2 * 16 * r) // PAGE) * PAGE for r in self.rows]
        self.paths = [f"/dev/shm/{tag}_{r}" for r in range(world)]
        self.maps, self.addrs = [None] * world, [0] * world
        fd = os.open(self.paths[rank], os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
        os.ftruncate(fd, self.sizes[rank])
        self._map(rank, fd)
        gpu.set_result_buffer(self.addrs[rank], self.sizes[rank])
        self._wait = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_ulonglong, C.c_void_p, C.c_double)(("hc_wait_result_buffer", self.lib))

    def _map(self, r, fd):
        self.maps[r] = mmap.mmap(fd, self.sizes[r])
        os.close(fd)
        self.addrs[r] = C.addressof(C.c_char.from_buffer(self.maps[r]))

    def attach(self):
        """After a barrier (every rank has created its segment): map the other ranks' buffers."""
        for r in range(self.world):
            if r != self.rank:
                self._map(r, os.open(self.paths[r], os.O_RDWR))

    def sequence(self):
        return self.gpu.step_sequence()

    def gather_into(self, seq, out_addr, timeout=20.0):
        """Rows of every shard of step `seq` into the 6N doubles at out_addr (own shard first: it is the one to arrive first)."""
        for r in [self.rank] + [x for x in range(self.world) if x != self.rank]:
            rc = self._wait(self.addrs[r], self.rows[r], seq, out_addr + 8 * 6 * self.shards[r][0], timeout)
            if rc:
                raise RuntimeError(f"rank {self.rank}: the rows of rank {r} for step sequence {seq} did not arrive")

    def gather(self, seq):
        out = np.empty(6 * self.N)
        self.gather_into(seq, out.ctypes.data)
        return out

    def close(self):
        self.gpu.set_result_buffer(None, 0)
        self.addrs = [0] * self.world
        for m in self.maps:
            if m is not None:
                try:
                    m.close()
                except BufferError:
                    pass
        try:
            os.unlink(self.paths[self.rank])
        except OSError:
            pass
