"""The multi-device code paths on ONE device (what a one-GPU box can put on hardware): concurrent calls through the C ABI
from several host threads (include/pnx.h: "every entry point may be called from several host threads at once"), the
plugin's n_gpus sharding (solvers.py: one host thread per device through the blocking ABI) with every shard mapped to
device 0, and bench.py's N-rank flow as fresh child processes that share the card.  Mirrors the reference's own
two-worker test (tests/test_solver_curvefit.py:931-952: n_pools=2 equals the serial result)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _triexp(n_vox, seed=7):
    from pyneapple_amd import synth

    b, y, _ = synth.make_numpy("tri_reduced", n_vox, 32, sigma=0.01, seed=seed)
    _, p0, lo, hi = synth.shared_arrays("tri_reduced")
    return b, y, p0, lo, hi


def test_curvefit_abi_from_four_threads_equals_one_call(gpu):
    n = 40000 + 37
    b, y, p0, lo, hi = _triexp(n)
    whole = gpu.curvefit("tri_reduced", b, y, p0, lo, hi)
    cuts = [0, 9000, 20011, 31000, n]
    with ThreadPoolExecutor(4) as ex:
        parts = list(ex.map(lambda k: gpu.curvefit("tri_reduced", b, y[cuts[k]:cuts[k + 1]], p0, lo, hi), range(4)))
    np.testing.assert_array_equal(np.concatenate([r["popt"] for r in parts], axis=1), whole["popt"])
    np.testing.assert_array_equal(np.concatenate([r["pcov"] for r in parts], axis=0), whole["pcov"])
    for k in ("status", "nfev", "cost"):
        np.testing.assert_array_equal(np.concatenate([r[k] for r in parts]), whole[k])


def test_streamed_curvefit_calls_from_three_threads_at_once(gpu, monkeypatch, capfd):
    """Three host-array calls at the same time on one device, each as its own streamed persistent kernel (granules of 4 Ki
    voxels so that the batches take that path): one works on the device's kept staging set, the others on private ones; a
    kernel that cannot become resident while another occupies the registers starts when that one ends -- nobody waits for
    anybody else's data, so the calls complete without running into the watermark poll limit, bit-equal to the single call."""
    n = 250000 + 37  # each batch alone asks for more workgroups than the chip has CUs
    b, y, p0, lo, hi = _triexp(n)
    whole = gpu.curvefit("tri_reduced", b, y, p0, lo, hi)
    monkeypatch.setenv("PNX_STREAM_GRANULE_SHIFT", "13")
    monkeypatch.setenv("PNX_STREAM_IN_CHUNK", "8192")
    monkeypatch.setenv("PNX_HOST_TRACE", "1")
    cuts = [0, 80000, 165011, n]
    capfd.readouterr()
    with ThreadPoolExecutor(3) as ex:
        parts = list(ex.map(lambda k: gpu.curvefit("tri_reduced", b, y[cuts[k]:cuts[k + 1]], p0, lo, hi), range(3)))
    err = capfd.readouterr().err
    assert err.count("granules of") == 3 and "TIMED OUT" not in err and "timed out" not in err
    np.testing.assert_array_equal(np.concatenate([r["popt"] for r in parts], axis=1), whole["popt"])
    np.testing.assert_array_equal(np.concatenate([r["pcov"] for r in parts], axis=0), whole["pcov"])
    for k in ("status", "nfev", "cost"):
        np.testing.assert_array_equal(np.concatenate([r[k] for r in parts]), whole[k])


def test_nnls_abi_from_four_threads_equals_one_call(gpu):
    from pyneapple_amd import synth

    n = 6000 + 5
    bins, basis, reg = synth.nnls_matrices(32)
    _, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01, scale=1000.0, seed=3)
    whole = gpu.nnls(basis, reg, y, 250)
    cuts = [0, 1500, 3001, 4700, n]
    # a plan per thread (four plans, four kernels in flight on the card) ...
    with ThreadPoolExecutor(4) as ex:
        parts = list(ex.map(lambda k: gpu.nnls(basis, reg, y[cuts[k]:cuts[k + 1]], 250), range(4)))
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(np.concatenate([r[k] for r in parts], axis=0), whole[k])
    # ... and ONE plan shared by four threads (its scratch serves one solve at a time: the plan's lock)
    plan = gpu.NnlsPlan(basis, reg, 0)
    with ThreadPoolExecutor(4) as ex:
        parts = list(ex.map(lambda k: plan.solve(y[cuts[k]:cuts[k + 1]], 250), range(4)))
    for k in ("coefficients", "residual", "status", "iters"):
        np.testing.assert_array_equal(np.concatenate([r[k] for r in parts], axis=0), whole[k])


def test_plugin_n_gpus_shards_equal_the_single_device_fit(gpu, monkeypatch):
    from pyneapple_amd.models import NNLSModel, TriExpModel
    from pyneapple_amd.solvers import HipCurveFitSolver, HipNNLSSolver
    from pyneapple_amd import synth

    monkeypatch.setenv("PNX_SHARE_DEVICE", "1")  # both shards on device 0 (the box has one)
    b, y, p0, lo, hi = _triexp(12000 + 3)
    names = list(synth.P0["tri_reduced"])
    mk = lambda n: HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(synth.P0["tri_reduced"]),
                                     bounds=dict(synth.BOUNDS["tri_reduced"]), n_gpus=n)
    one, two = mk(1).fit(b, y), mk(2).fit(b, y)
    for nme in names:
        np.testing.assert_array_equal(one.params_[nme], two.params_[nme])
    np.testing.assert_array_equal(one.diagnostics_["pcov"], two.diagnostics_["pcov"])
    np.testing.assert_array_equal(one.diagnostics_["status"], two.diagnostics_["status"])
    # per-voxel start values and a per-voxel fixed parameter are split with the rows
    rng = np.random.default_rng(0)
    p0v = np.tile(p0[:, None], (1, y.shape[0])) * rng.uniform(0.95, 1.05, (5, y.shape[0]))
    one, two = mk(1).fit(b, y, p0=p0v), mk(3).fit(b, y, p0=p0v)
    for nme in names:
        np.testing.assert_array_equal(one.params_[nme], two.params_[nme])

    cfg = synth.NNLS_CFG
    _, ys, _ = synth.make_numpy("tri_reduced", 3000 + 1, 32, sigma=0.01, scale=1000.0, seed=11)
    mkn = lambda n: HipNNLSSolver(model=NNLSModel(d_range=cfg["d_range"], n_bins=cfg["n_bins"]), reg_order=cfg["reg_order"],
                                  mu=cfg["mu"], max_iter=cfg["max_iter"], n_gpus=n)
    one, two = mkn(1).fit(b, ys), mkn(2).fit(b, ys)
    np.testing.assert_array_equal(one.params_["coefficients"], two.params_["coefficients"])
    np.testing.assert_array_equal(one.diagnostics_["residual"], two.diagnostics_["residual"])
    np.testing.assert_array_equal(one.diagnostics_["iters"], two.diagnostics_["iters"])


def test_bench_two_ranks_as_child_processes_on_one_card():
    """bench.py --gpus 2 end to end: the launcher spawns two fresh ranks before anything touches the GPU, both put their
    shard on device 0 (PNX_BENCH_SHARE_GPU), gloo carries the barrier and the max-reduce (no second card for RCCL).  Since
    round 4 the N > 1 line carries the PCIe-inclusive leg too: both ranks hand their shard to the ABI as numpy arrays at the
    same moment (barrier-aligned), pinned to disjoint CPU sets, and get the device-resident results back bit for bit."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PNX_BENCH_SHARE_GPU="1", PNX_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--voxels", "262144", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 2
    assert d["config"]["voxels_per_rank"] == [131072, 131072] and d["config"]["voxels_total"] == 262144
    assert d["check"]["converged_frac"] > 0.995
    assert d["value"] > 0 and d["throughput"]["n_gpus"] == 2 and d["throughput"]["value"] > 0
    assert d["pipelined"]["results_identical_across_buffers"] is True
    # the PCIe-inclusive leg of both workloads, every rank streaming its shard at once
    for hm in (d["host_mode"], d["secondary"]["host_mode"]):
        assert hm["n_gpus"] == 2 and hm["value"] > 0 and len(hm["per_rank_pcie_GBps"]) == 2 and min(hm["per_rank_pcie_GBps"]) > 0
        assert hm["equals_device_resident_result"] is True
    assert d["c3_host_voxels_per_s"] == d["host_mode"]["value"] and d["c4_host_voxels_per_s"] == d["secondary"]["host_mode"]["value"]
    assert d["nnls_voxels_per_s"] == d["secondary"]["value"] > 0
    # each rank reports the CPUs it pinned itself to; two ranks on one card take disjoint sets
    aff = d["affinity"]
    assert len(aff) == 2 and all(a and a.get("cpus") for a in aff)
    assert aff[0]["cpus"] != aff[1]["cpus"]
    head = lines[0][:1500]
    for k in ("nnls_voxels_per_s", "nnls_ms_per_step", "c3_host_voxels_per_s", "c4_host_voxels_per_s", "throughput_voxels_per_s"):
        assert f'"{k}": ' in head


def test_bench_four_ranks_on_one_card():
    """ The largest rank count one card admits beside this test process (the pool allows six GPU
    processes): four ranks, four disjoint CPU slices, four staging sets streaming at once through the PCIe-inclusive leg,
    voxels_per_rank summing to the volume, one JSON line.  gloo carries the barrier (RCCL needs a device per rank and has
    never run with more than one: DESIGN.md section 6); the N = 8 launcher / row split / CPU slicing is rehearsed without a GPU
    in tests/test_bench_sharding.py."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PNX_BENCH_SHARE_GPU="1", PNX_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--voxels", "524288", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["voxels_per_rank"] == [131072] * 4 and sum(d["config"]["voxels_per_rank"]) == 524288
    assert d["value"] > 0 and d["nnls_voxels_per_s"] > 0 and d["check"]["converged_frac"] > 0.995
    for hm in (d["host_mode"], d["secondary"]["host_mode"]):
        assert hm["n_gpus"] == 4 and len(hm["per_rank_pcie_GBps"]) == 4 and min(hm["per_rank_pcie_GBps"]) > 0
        assert hm["equals_device_resident_result"] is True
    aff = d["affinity"]
    assert len(aff) == 4 and all(a and a.get("cpus") for a in aff)
    import bench

    sets = [bench._cpulist(a["cpus"]) for a in aff]
    assert all(not (sets[i] & sets[j]) for i in range(4) for j in range(i + 1, 4))
    assert "noise_sweep" not in d  # N = 1 only


def test_bench_two_ranks_under_torch_distributed_run_on_one_card():
    """The driver's own N > 1 command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with two ranks on this box's one card (PNX_BENCH_SHARE_GPU, gloo for the barrier
    and the max-reduce): RANK / LOCAL_RANK / WORLD_SIZE come from the launcher's environment instead of bench.py's own
    spawner, and rank 0 alone prints the line."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PNX_BENCH_SHARE_GPU="1", PNX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--voxels", "262144", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--no-host-mode"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["voxels_per_rank"] == [131072, 131072]
    assert d["value"] > 0 and d["nnls_voxels_per_s"] > 0 and d["check"]["converged_frac"] > 0.995
    assert len(d["affinity"]) == 2
