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


def _bench_worker(rank, world, port, out_dir, fail_rank):
    """bench.py's barrier-aligned timing of a host-array call on every rank (all_ranks_timed), with a stand-in for the call."""
    import json
    import sys
    import time

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    calls = [0]

    def call():
        calls[0] += 1
        if rank == fail_rank and calls[0] == 2:  # the warm-up works, the first timed call fails
            raise MemoryError("no room for the result")
        time.sleep(0.01 * (rank + 1))
        return {"popt": np.zeros((3, 100 * (rank + 1)))}

    out = bench.all_ranks_timed(call, lambda r: (True, int(r["popt"].nbytes)), 3, 1000, world, dist, "cpu")
    with open(os.path.join(out_dir, f"b{rank}.json"), "w") as fh:
        json.dump({"out": out, "calls": calls[0]}, fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_host_leg_of_the_bench_on_two_ranks(tmp_path, fail_rank):
    """The PCIe-inclusive leg of an N > 1 bench line: the call time of a repetition is the slowest rank's, per-rank rates come
    back on every rank -- and a rank whose call fails is reported in the line instead of leaving its peer at a barrier."""
    import json

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_bench_worker, args=(2, port, str(tmp_path), fail_rank), nprocs=2, join=True)
    res = [json.load(open(tmp_path / f"b{k}.json")) for k in range(2)]
    if fail_rank < 0:
        for r in res:
            o = r["out"]
            assert o["n_gpus"] == 2 and o["steps"] == 3 and o["equals_device_resident_result"] is True
            assert 18.0 < o["ms_per_step"] < 200.0           # the slower rank (20 ms) sets the time
            assert o["value"] == pytest.approx(1000 / (o["ms_per_step"] * 1e-3))
            assert len(o["per_rank_pcie_GBps"]) == 2 and min(o["per_rank_pcie_GBps"]) > 0
            assert o["per_rank_ms"][0][0] < o["per_rank_ms"][1][0]
        assert res[0]["out"]["ms_reps"] == res[1]["out"]["ms_reps"]   # both ranks report the same line
    else:
        for k, r in enumerate(res):
            o = r["out"]
            assert o["value"] is None and "rank(s) [1]" in o["error"]
            assert ("MemoryError" in o["error"]) == (k == 1)
        assert res[1]["calls"] == 2 and res[0]["calls"] == 4          # the healthy rank ran all its calls and did not hang
