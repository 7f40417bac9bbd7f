"""Sector sharding across GPUs: sectors are independent, so a volume scan is split by
sector index with NO data-path collective.  torch.distributed (gloo on CPU tensors, on the GPU box
and in the CPU tests alike: RCCL is never initialised) is used only for the barrier, the MAX of the
elapsed time and -- optionally -- gathering the per-rank result tables on rank 0 (control plane,
4 KiB per sector)."""


def sectors_for_rank(n_sectors, rank, world):
    """Sector s of a sweep runs on GPU s mod G."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_sectors, world))


def owner_of(sector, world):
    return sector % world


def volume_plan(n_elevations, n_sectors, rank, world):
    """(elevation, sector) pairs of this rank for a volume scan, in acquisition order."""
    return [(e, s) for e in range(n_elevations) for s in sectors_for_rank(n_sectors, rank, world)]


def max_over_ranks(dist, seconds, device=None):
    """Whole-job time = slowest rank (bench.py contract)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_results(dist, local, n_elevations, n_sectors, gates, rank, world):
    """local: {(elev, sector): float32[gates][2]} of this rank -> full table on rank 0 (else None)."""
    import numpy as np
    parts = [None] * world
    dist.all_gather_object(parts, {k: v.tobytes() for k, v in local.items()})
    if rank != 0:
        return None
    table = np.full((n_elevations, n_sectors, gates, 2), np.nan, np.float32)
    for p in parts:
        for (e, s), b in p.items():
            table[e, s] = np.frombuffer(b, np.float32).reshape(gates, 2)
    return table
