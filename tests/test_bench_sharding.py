"""bench.py's multi-GPU plumbing on CPU: the launcher spawns N fresh ranks, the ranks shard ONE seed-fixed volume,
and fitting the shards equals fitting the whole volume bit for bit (the per-rank fit here is the oracle -- there is
no GPU in CI; on the MI355X node the same row ranges go through the C ABI)."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True,
                          timeout=120)


def test_launcher_spawns_n_ranks_that_partition_one_volume():
    r = _run(["--gpus", "3", "--launch-check"])
    assert r.returncode == 0, r.stderr
    ranks = sorted((json.loads(l) for l in r.stdout.splitlines() if l.strip()), key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == [0, 1, 2] and all(d["world"] == 3 for d in ranks)
    assert [d["local_rank"] for d in ranks] == [0, 1, 2]
    n = ranks[0]["n_vox_total"]
    assert n == 256 * 256 * 64
    assert ranks[0]["rows"][0] == 0 and ranks[-1]["rows"][1] == n
    assert all(a["rows"][1] == b["rows"][0] for a, b in zip(ranks[:-1], ranks[1:]))


def test_eight_ranks_rehearsal_launcher_row_split_and_cpu_slices():
    """The N = 8 the driver runs, without a GPU (a GPU box admits at most six processes on its card, so eight ranks cannot be
    rehearsed there; tests/test_gpu_multi.py runs four on one card): eight fresh ranks, contiguous row ranges that partition the
    C3 volume exactly, and -- as on a card shared by all ranks -- eight CPU slices that are pairwise disjoint wherever this
    process may use at least eight CPUs."""
    r = _run(["--gpus", "8", "--launch-check", "--voxels", "524288"], env={"PNX_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, r.stderr
    ranks = sorted((json.loads(l) for l in r.stdout.splitlines() if l.strip()), key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == list(range(8)) and all(d["world"] == 8 for d in ranks)
    assert sum(d["rows"][1] - d["rows"][0] for d in ranks) == 524288 == ranks[0]["n_vox_total"]
    assert ranks[0]["rows"][0] == 0 and all(a["rows"][1] == b["rows"][0] for a, b in zip(ranks[:-1], ranks[1:]))
    import bench

    sets = [bench._cpulist(d["affinity"]["cpus"]) for d in ranks]
    assert all(sets)
    if len(os.sched_getaffinity(0)) >= 8:
        assert all(not (sets[i] & sets[j]) for i in range(8) for j in range(i + 1, 8))


def test_launcher_stops_the_other_ranks_when_one_dies():
    """A rank that exits non-zero at start-up must not leave its peers waiting in the rendezvous until the backend's
    timeout: the launcher polls all ranks, terminates the rest and returns the failing rank's code."""
    import time

    t = time.time()
    r = _run(["--gpus", "3", "--launch-check", "--fail-rank", "2"], env={"PNX_BENCH_LAUNCH_HOLD": "60"})
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "rank 2 exited with code 3" in r.stderr
    assert time.time() - t < 30  # not the 60 s the surviving ranks would have waited


def test_gpus_flag_must_match_world_size():
    r = _run(["--gpus", "4", "--launch-check"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_single_process_default_is_one_rank():
    r = _run(["--launch-check"])
    assert r.returncode == 0
    d = json.loads(r.stdout)
    assert d["world"] == 1 and d["rows"] == [0, 256 * 256 * 64]


def test_shards_of_the_volume_are_the_volume_and_fit_identically(oracle):
    import torch

    from pyneapple_amd import synth
    from pyneapple_amd.sharding import shard_range

    n = synth.ROW_CHUNK + 1234  # crosses a generator block boundary
    _, whole = synth.make_torch_rows("tri_reduced", 0, n, 32, "cpu")
    parts = [synth.make_torch_rows("tri_reduced", *shard_range(n, r, 3), 32, "cpu")[1] for r in range(3)]
    assert torch.equal(torch.cat(parts), whole)
    # fit a window around the first shard boundary whole and sharded: identical rows
    b = synth.bvalues(32)
    _, p0, lo, hi = synth.shared_arrays("tri_reduced")
    cut = shard_range(n, 0, 3)[1]
    win = whole[cut - 40:cut + 40].numpy()
    ref = oracle.curvefit("tri_reduced", b, win, p0, lo, hi)
    left = oracle.curvefit("tri_reduced", b, parts[0][-40:].numpy(), p0, lo, hi)
    right = oracle.curvefit("tri_reduced", b, parts[1][:40].numpy(), p0, lo, hi)
    np.testing.assert_array_equal(np.concatenate([left["popt"], right["popt"]], axis=1), ref["popt"])


def test_pinned_to_gpu_is_a_harmless_no_op_without_numa_information():
    """The plugin's per-device threads pin themselves to their GPU's NUMA node (sysfs only).  In this container there is no KFD
    topology: the context manager must leave the mask alone, and restore it where it did change it."""
    import os

    from pyneapple_amd.sharding import cpulist, fmt_cpus, gpu_numa_cpus, pinned_to_gpu

    before = os.sched_getaffinity(0)
    node, cpus = gpu_numa_cpus(0)
    with pinned_to_gpu(0) as pin:
        inside = os.sched_getaffinity(0)
        assert inside <= before and inside
        if cpus is None or not (cpus & before):
            assert inside == before and pin.prev is None
    assert os.sched_getaffinity(0) == before
    with pinned_to_gpu(63):  # no such GPU
        assert os.sched_getaffinity(0) == before
    assert cpulist("0-2,5\n") == {0, 1, 2, 5} and fmt_cpus({0, 1, 2, 5}) == "0-2,5"
