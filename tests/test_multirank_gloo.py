"""world_size-2 rehearsal of the N>1 path on CPU (gloo): shard -> fit -> re-assemble == single process.

The per-rank fit here is the oracle (no GPU in this container); what is under test is the product's
sharding / re-assembly / max-reduce code in pyneapple_amd/sharding.py, which bench.py uses unchanged
with the nccl (RCCL) backend."""
from __future__ import annotations

import os
import socket

import numpy as np
import pytest

from pyneapple_amd.sharding import shard_range


def test_shard_range_partitions():
    for n, w in ((10, 3), (4194304, 8), (7, 8), (0, 2)):
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pnx_oracle as O
    from pyneapple_amd import synth
    from pyneapple_amd.sharding import gather_rows, max_over_ranks, shard_range

    b, y, _ = synth.make_numpy("bi_reduced", 301, 24, sigma=0.01, seed=11)  # odd count: ragged shards
    _, p0, lo, hi = synth.shared_arrays("bi_reduced")
    a, e = shard_range(len(y), rank, world)
    r = O.curvefit("bi_reduced", b, y[a:e], p0, lo, hi)
    popt = gather_rows(np.ascontiguousarray(r["popt"].T), dist)
    status = gather_rows(r["status"].astype(np.int64), dist)
    t = max_over_ranks(1.0 + rank, dist)
    dist.barrier()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), popt=popt, status=status, t=t)
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    import torch.multiprocessing as mp

    from oracle import pnx_oracle as O
    from pyneapple_amd import synth

    O.lib()  # build before forking workers
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    b, y, _ = synth.make_numpy("bi_reduced", 301, 24, sigma=0.01, seed=11)
    _, p0, lo, hi = synth.shared_arrays("bi_reduced")
    ref = O.curvefit("bi_reduced", b, y, p0, lo, hi)
    for rank in range(2):
        d = np.load(tmp_path / f"r{rank}.npz")
        np.testing.assert_array_equal(d["popt"], ref["popt"].T)   # bit-exact: same code, same rows
        np.testing.assert_array_equal(d["status"], ref["status"])
        assert float(d["t"]) == 2.0                               # max over ranks
