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


# ---- host threads next to their GPU ----------------------------------------------------------------------------------
# The host feed of N devices (uploads, downloads, the first-touch of result pages) wants to run on the NUMA node each GPU hangs
# off; everything here reads sysfs only -- it never touches the GPU, so it can run before HIP is initialised (bench.py pins a
# whole rank that way) and inside a thread that is about to make a blocking ABI call (the plugin's n_gpus threads).
def cpulist(text: str) -> set:
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def fmt_cpus(cpus) -> str:
    c = sorted(cpus)
    runs, i = [], 0
    while i < len(c):
        j = i
        while j + 1 < len(c) and c[j + 1] == c[j] + 1:
            j += 1
        runs.append(str(c[i]) if i == j else f"{c[i]}-{c[j]}")
        i = j + 1
    return ",".join(runs)


def gpu_numa_cpus(index: int):
    """(numa node, set of CPUs) local to the `index`-th visible GPU -- KFD topology -> PCI device -> local_cpulist -- or
    (None, None) where sysfs does not tell."""
    import glob
    import os

    gpus = []
    for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda q: int(os.path.basename(q))):
        try:
            props = dict(l.split(None, 1) for l in open(os.path.join(d, "properties")).read().splitlines() if " " in l)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            gpus.append(props)
    vis = os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis:
        try:
            gpus = [gpus[int(v)] for v in vis.split(",") if v.strip() != ""]
        except (ValueError, IndexError):
            return None, None
    if index < 0 or index >= len(gpus):
        return None, None
    loc, dom = int(gpus[index].get("location_id", "0")), int(gpus[index].get("domain", "0"))
    bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
    base = f"/sys/bus/pci/devices/{bdf}"
    try:
        node = int(open(base + "/numa_node").read())
        cpus = cpulist(open(base + "/local_cpulist").read())
    except (OSError, ValueError):
        return None, None
    return node, cpus


class pinned_to_gpu:
    """Context manager: the CALLING THREAD (and the helper threads the library starts from it, which inherit the mask) runs on
    the CPUs of `device`'s NUMA node while inside; the previous mask comes back on exit.  A no-op where sysfs does not tell,
    where the node's CPUs are not in the current mask, or with PNX_NO_PIN=1.  Never raises."""

    def __init__(self, device: int):
        self.device, self.prev, self.cpus = int(device), None, None

    def __enter__(self):
        import os

        try:
            if os.environ.get("PNX_NO_PIN") == "1":
                return self
            have = os.sched_getaffinity(0)
            _, local = gpu_numa_cpus(self.device)
            mine = (local or set()) & have
            if mine and mine != have:
                os.sched_setaffinity(0, mine)
                self.prev, self.cpus = have, mine
        except Exception:  # noqa: BLE001 -- an affinity nicety must not fail a fit
            self.prev = None
        return self

    def __exit__(self, *exc):
        import os

        if self.prev is not None:
            try:
                os.sched_setaffinity(0, self.prev)
            except Exception:  # noqa: BLE001
                pass
        return False
