"""Voxel sharding across one-process-per-GPU ranks (torch.distributed; nccl = RCCL on the GPUs, gloo on CPU).

The path has no exchange step: voxels are independent (reference solvers/curvefit.py:204-212), so each rank
fits a contiguous row range of the (n_vox, n_b) signal matrix.  The only communication is the optional
re-assembly of the parameter maps on every rank (all_gather of the result rows) and the max-reduce of the
timed region in bench.py.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [start, stop) of `n` rows owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_rows(local, dist=None, device=None):
    """Concatenate per-rank row blocks (axis 0) on every rank.  Row counts may differ between ranks."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    if device is not None:
        t = t.to(device)
    counts = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    return out.cpu().numpy() if is_np else out


def max_over_ranks(value: float, dist=None, device=None) -> float:
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
