#!/usr/bin/env python3
"""The streamed host path (one persistent kernel per host-array call, pnx_api.hip curvefit_streamed) against the chunk ring on
random configurations: model, number of b-values (odd: no LDS-DMA refill), volume size, float32 / float64 arrays, covariance
on / off, Jacobian mode, a shared or per-voxel fixed parameter, per-voxel start values and bounds, the T1 factor, granule size,
upload piece size, download and page-touch thread counts, a small evaluation budget (failure sentinels).  Every output must be
bit-identical, and the streamed call must neither time out nor fall back.
    python tests/fuzz_stream_vs_ring.py [n_cases] [seed]          (on a GPU box; 40 fixed-seed cases run in the GPU suite)"""
from __future__ import annotations
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")

import contextlib
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import api, synth  # noqa: E402

KNOBS = ("PNX_HOST_STREAM", "PNX_STREAM_GRANULE_SHIFT", "PNX_STREAM_IN_CHUNK", "PNX_STREAM_OUT_THREADS", "PNX_HOST_TOUCHERS",
         "PNX_HOST_CHUNK")


def draw_case(rng):
    model = str(rng.choice(["mono", "bi_reduced", "tri_reduced"]))
    n_b = int(rng.integers(6, 41))
    n_vox = int(rng.integers(1100, 30000))
    dtype = np.float32 if rng.random() < 0.4 else np.float64
    b, y, P = synth.make_numpy(model, n_vox, n_b, sigma=float(rng.choice([0.0, 0.01, 0.05])), seed=int(rng.integers(1 << 30)))
    names, p0, lo, hi = synth.shared_arrays(model)
    kw = dict(want_pcov=bool(rng.random() < 0.7), jac=str(rng.choice(["fd", "analytic"])), max_nfev=int(rng.choice([250, 250, 12])))
    kind = str(rng.choice(["plain", "plain", "fixed_shared", "fixed_map", "per_voxel", "t1"]))
    if kind in ("fixed_shared", "fixed_map") and model != "mono":
        j = int(rng.integers(len(names)))
        free = [i for i in range(len(names)) if i != j]
        truth = P[names[j]]
        kw.update(fixed_idx=[j], fixed_vals=(np.array([truth.mean()]) if kind == "fixed_shared" else truth[None, :].astype(dtype)),
                  jac="analytic")
        p0, lo, hi = p0[free], lo[free], hi[free]
    elif kind == "per_voxel":
        n = len(names)
        p0 = np.tile(p0[:, None], (1, n_vox)) * rng.uniform(0.9, 1.1, (n, n_vox))
        lo = np.tile(lo[:, None], (1, n_vox)) * rng.uniform(0.8, 1.0, (n, n_vox))
        hi = np.tile(hi[:, None], (1, n_vox)) * rng.uniform(1.0, 1.2, (n, n_vox))
    elif kind == "t1" and model == "mono":  # three free parameters with the factor: streamed; more would be register tight
        tr, tm = 3000.0, 25.0
        T1 = rng.uniform(800, 1600, n_vox)
        y = y * ((1 - np.exp(-tr / T1)) * np.exp(-tm / T1))[:, None]
        p0, lo, hi = np.append(p0, 1000.0), np.append(lo, 100.0), np.append(hi, 5000.0)
        kw.update(t1_mode=2, tr=tr, tm=tm)
    else:
        kind = "plain"
    env = {"PNX_STREAM_GRANULE_SHIFT": str(int(rng.integers(10, 13))), "PNX_STREAM_IN_CHUNK": str(int(rng.integers(1024, 20000))),
           "PNX_STREAM_OUT_THREADS": str(int(rng.integers(1, 4))), "PNX_HOST_TOUCHERS": str(int(rng.integers(0, 4))),
           "PNX_HOST_CHUNK": str(int(rng.integers(1024, 1 << 15)))}
    desc = f"{model} {kind} n_b={n_b} n_vox={n_vox} {np.dtype(dtype).name} {kw.get('jac')} pcov={kw['want_pcov']} max_nfev={kw['max_nfev']} {env}"
    return desc, model, b, y.astype(dtype), p0, lo, hi, kw, env


def run(n_cases=100, seed=0, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    rng = np.random.default_rng(seed)
    saved = {k: os.environ.get(k) for k in KNOBS + ("PNX_HOST_TRACE",)}
    bad, streamed, voxels = 0, 0, 0
    try:
        for c in range(n_cases):
            desc, model, b, y, p0, lo, hi, kw, env = draw_case(rng)
            os.environ.update(env)
            os.environ["PNX_HOST_STREAM"] = "0"
            os.environ.pop("PNX_HOST_TRACE", None)
            ring = api.curvefit(model, b, y, p0, lo, hi, **kw)
            os.environ["PNX_HOST_STREAM"] = "1"
            os.environ["PNX_HOST_TRACE"] = "1"
            err = _stderr_of(lambda: api.curvefit(model, b, y, p0, lo, hi, **kw))
            st, text = err
            voxels += y.shape[0]
            n_gran = -(-y.shape[0] >> int(env["PNX_STREAM_GRANULE_SHIFT"]))
            want_stream = n_gran >= 2
            is_stream = "[pnx stream]" in text and "granules of" in text
            streamed += is_stream
            problems = []
            if "timed out" in text or "TIMED OUT" in text or "STALLED" in text or "could not be used" in text:
                problems.append("watermark time-out / fall-back")
            if is_stream != want_stream:
                problems.append(f"streamed={is_stream}, expected {want_stream}")
            for k in ("popt", "pcov", "status", "nfev", "cost"):
                if (ring[k] is None) != (st[k] is None) or (ring[k] is not None and not np.array_equal(ring[k], st[k], equal_nan=True)):
                    problems.append(f"{k} differs")
            if problems:
                bad += 1
                say(f"[case {c}] FAIL {problems}: {desc}")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    say(f"{n_cases} cases, {voxels} voxels, {streamed} streamed: {bad} failing")
    return {"fuzzer": "stream_vs_ring", "n_cases": n_cases, "seed": seed, "voxels": voxels, "streamed_cases": streamed, "failing_cases": bad}


def _stderr_of(fn):
    """Run fn with the process's stderr (fd 2: the library writes its trace there) captured; returns (result, text)."""
    import tempfile
    sys.stderr.flush()
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        old = os.dup(2)
        os.dup2(tmp.fileno(), 2)
        try:
            res = fn()
        finally:
            os.dup2(old, 2)
            os.close(old)
        tmp.seek(0)
        return res, tmp.read().decode(errors="replace")


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    out = run(int(a[0]) if a else 100, int(a[1]) if len(a) > 1 else 0)
    if "--json" in sys.argv:
        import json
        from pyneapple_amd import _build
        out["source_ids"] = _build.source_ids()
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    sys.exit(1 if out["failing_cases"] else 0)
