"""Per-clip sharding across the GPUs of one node.

The path has no exchange step: every frame (Whisper) / every clip (fbank, because of CMN) is
independent, so ranks take contiguous blocks of clips and no collective
touches the data.  torch.distributed (RCCL) is used by bench.py only for the timing barrier
and the max-over-ranks of the elapsed time.
"""
from __future__ import annotations


def shard_range(n_items: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [lo, hi) block of `n_items` owned by `rank`; sizes differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_by_samples(lengths, world_size: int) -> list[tuple[int, int]]:
    """Contiguous clip ranges balanced by total samples (ragged batches)."""
    total = float(sum(lengths))
    bounds, acc, lo, r = [], 0.0, 0, 1
    for i, n in enumerate(lengths):
        acc += n
        while r < world_size and acc >= total * r / world_size:
            bounds.append((lo, i + 1))
            lo = i + 1
            r += 1
    bounds.append((lo, len(lengths)))
    while len(bounds) < world_size:
        bounds.append((len(lengths), len(lengths)))
    return bounds[:world_size]


def timed_steps(step, synchronize, steps: int, warmup: int, dist=None, device=None):
    """The bench protocol: `warmup` untimed steps, then exactly `steps` timed steps bracketed by a
    barrier + synchronize on both sides; returns the MAX elapsed seconds over ranks.

    `dist` is torch.distributed (initialised) or None for a single process; `device` is where the
    reduction tensor lives (cuda device for nccl, None/cpu for gloo)."""
    import time
    for _ in range(warmup):
        step()
    synchronize()
    if dist is not None:
        dist.barrier()
    synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
