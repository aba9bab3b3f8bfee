"""The bench line's layout (half of the metric was cut off in the driver's record).  `finalize` is a
pure function, so it is exercised here on a committed line; the NUMA pinning helper runs on whatever CPUs this container has."""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _committed_line():
    with open(os.path.join(ROOT, "profiles", "r03_i_bench.json")) as fh:
        return json.load(fh)


def test_the_five_scalars_sit_in_the_first_1500_bytes():
    raw = _committed_line()
    line = json.dumps(bench.finalize(raw))
    head = line[:1500]
    for k in bench.SCALAR_KEYS:
        assert f'"{k}": ' in head, k
    d = json.loads(line)
    assert d["nnls_voxels_per_s"] == raw["secondary"]["value"]
    assert d["nnls_ms_per_step"] == raw["secondary"]["ms_per_step"]
    assert d["c3_host_voxels_per_s"] == raw["host_mode"]["value"]
    assert d["c4_host_voxels_per_s"] == raw["secondary"]["host_mode"]["value"]
    assert d["throughput_voxels_per_s"] == raw["throughput"]["value"]
    # roofline and cpu_baseline follow the scalars, still inside what a 1500-byte tail keeps of their leading fields
    assert head.index('"roofline"') > head.index('"throughput_voxels_per_s"')
    assert '"frac": ' in head and '"cpu_baseline"' in head


def test_noise_sweep_and_nnls_spread_sit_in_the_first_2000_bytes():
    """ The data dependence of both rates travels with them: noise_sweep (c3 / c4 voxels/s at sigma 0 / 1 /
    5 %) and the min / median of the three NNLS passes follow the scalars, inside what a 2 000-byte reader keeps."""
    raw = _committed_line()
    raw["noise_sweep"] = {"sigma": [0.0, 0.01, 0.05], "c3_voxels": 1 << 22, "c4_voxels": 1 << 20, "c3_voxels_per_s": [160.1e6, 113.2e6, 75.3e6],
                          "c4_voxels_per_s": [4.7e6, 9.1e6, 14.6e6]}
    raw["secondary"].update(ms_per_step_min=455.1, ms_per_step_median=456.2)
    line = json.dumps(bench.finalize(raw))
    head = line[:2000]
    for k in ('"noise_sweep": ', '"c3_voxels_per_s": ', '"c4_voxels_per_s": ', '"nnls_ms_per_step_min": ', '"nnls_ms_per_step_median": '):
        assert k in head, k
    d = json.loads(line)
    assert d["noise_sweep"] == raw["noise_sweep"] and d["nnls_ms_per_step_min"] == 455.1
    keys = list(d.keys())
    assert keys.index("noise_sweep") == len(bench.HEAD_KEYS) + len(bench.SCALAR_KEYS)
    assert keys.index("roofline") > keys.index("nnls_ms_per_step_median")
    assert tuple(bench.NOISE_SWEEP_SIGMAS) == (0.0, 0.01, 0.05) and bench.NOISE_SWEEP_VOXELS == 1 << 20 and bench.NOISE_SWEEP_VOXELS_C3 == 1 << 22


def test_contract_keys_come_first_and_prose_comes_last():
    d = bench.finalize(_committed_line())
    keys = list(d.keys())
    assert tuple(keys[: len(bench.HEAD_KEYS)]) == bench.HEAD_KEYS
    assert keys[-1] == "notes" and isinstance(d["notes"], dict) and d["notes"]
    assert isinstance(d["config"]["workload"], str)  # the contract's workload name stays where the contract puts it

    def walk(o, path=()):
        for k, v in o.items():
            if isinstance(v, dict) and k != "notes":
                walk(v, path + (k,))
            elif path and path != ("config",):
                assert k not in ("note", "workload", "mode"), path + (k,)

    walk(d)
    assert "secondary.workload" in d["notes"] and "roofline.note" in d["notes"]
    # nothing is lost: every prose string of the raw line is in notes
    assert d["notes"]["secondary.workload"] == _committed_line()["secondary"]["workload"]


def test_nnls_workload_line_reports_its_own_rate_as_the_nnls_scalar():
    raw = {"metric": "m", "value": 7.0e6, "unit": "voxels/s", "ms_per_step": 600.0, "host_mode": {"value": 6.5e6, "workload": "w"}}
    d = bench.finalize(raw, "nnls")
    assert d["nnls_voxels_per_s"] == 7.0e6 and d["c4_host_voxels_per_s"] == 6.5e6 and d["c3_host_voxels_per_s"] is None


def test_cpu_list_round_trip():
    assert bench._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert bench._fmt_cpus({0, 1, 2, 3, 8, 10, 11}) == "0-3,8,10-11"


def test_pinning_two_ranks_without_numa_information_splits_the_affinity_mask():
    """Two ranks that share a card (the one-GPU rehearsal) take disjoint halves of the CPUs this process may use; the call never
    raises and restores nothing -- it runs in a child so that the test process keeps its own mask."""
    import subprocess

    code = ("import json, os, sys; sys.path.insert(0, %r); import bench\n"
            "before = sorted(os.sched_getaffinity(0))\n"
            "r = bench.pin_rank_to_gpu_numa(int(sys.argv[1]), 2, True)\n"
            "print(json.dumps({'before': before, 'after': sorted(os.sched_getaffinity(0)), 'r': r}))\n" % ROOT)
    outs = [json.loads(subprocess.run([sys.executable, "-c", code, str(k)], capture_output=True, text=True, check=True).stdout) for k in (0, 1)]
    a, b = set(outs[0]["after"]), set(outs[1]["after"])
    assert a and b and a <= set(outs[0]["before"]) and b <= set(outs[1]["before"])
    if len(outs[0]["before"]) >= 2:
        assert not (a & b)
    assert outs[0]["r"]["cpus"] == bench._fmt_cpus(a)
