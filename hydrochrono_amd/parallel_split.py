"""Body-row partition of a coupled array over shards (no torch dependency; hydrochrono_amd.parallel re-exports it)."""


def body_shard(num_bodies, world, rank):
    """Contiguous balanced partition of bodies over ranks: the first (num_bodies % world) ranks get one extra."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if world > num_bodies:
        raise ValueError("more ranks than bodies: use replicas instead of sharding (SURVEY.md 8e)")
    base, extra = divmod(num_bodies, world)
    b0 = rank * base + min(rank, extra)
    return b0, b0 + base + (1 if rank < extra else 0)
