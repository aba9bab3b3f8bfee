#!/usr/bin/env python3
"""NNLS from host arrays on random chunkings: the chunk ring with the deferred hand-over (block-kernel plans), with one hand-over
pass per chunk, with a side buffer that is too small (overflow -> second run), float32 storage, and the peak-table variant --
all against the single-chunk call of the same rows, bit for bit.  Rows come from the seed-fixed synthetic volume whose first 2^15
rows contain voxels that the block kernel hands over (passive set beyond 128 positions).
    python tests/fuzz_nnls_host.py [n_cases] [seed]      (on a GPU box; 12 fixed-seed cases run in the GPU suite)"""
from __future__ import annotations
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import api, synth  # noqa: E402

KNOBS = ("PNX_NNLS_HOST_CHUNK", "PNX_NNLS_DEFER_CAP", "PNX_NNLS_PEAKS_CHUNK", "PNX_NNLS_PEAKS_RING", "PNX_HOST_TOUCHERS", "PNX_HOST_RAMP")


def run(n_cases=40, seed=0, verbose=True):
    import torch

    say = print if verbose else (lambda *a, **k: None)
    rng = np.random.default_rng(seed)
    bins, basis, reg = synth.nnls_matrices(32)
    _, yt = synth.make_torch_rows("tri_reduced", 0, 1 << 15, 32, torch.device("cuda", 0), sigma=0.01, scale=1000.0)
    y_all = yt.cpu().numpy()
    plan = api.NnlsPlan(basis, reg, 0)
    cuts = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
    pk = dict(max_iter=250, height=0.1, regularized=True, max_peaks=8, cutoffs=cuts)
    saved = {k: os.environ.get(k) for k in KNOBS}
    bad = handed = voxels = 0
    try:
        for c in range(n_cases):
            a = int(rng.integers(0, (1 << 15) - 3000))
            n = int(rng.integers(2500, min(14000, (1 << 15) - a)))
            y = y_all[a:a + n]
            f32 = rng.random() < 0.3
            peaks = (not f32) and rng.random() < 0.4
            yy = y.astype(np.float32) if f32 else y
            os.environ["PNX_NNLS_HOST_CHUNK"] = os.environ["PNX_NNLS_PEAKS_CHUNK"] = str(1 << 20)
            one = plan.solve_peaks(yy, bins, **pk) if peaks else plan.solve(yy, 250)
            n_hand = int(((plan.solve(y, 250)["coefficients"] > 0).sum(axis=1) > 128).sum()) if (peaks or f32) else \
                int(((one["coefficients"] > 0).sum(axis=1) > 128).sum())
            handed += n_hand
            chunk = int(rng.integers(1024, 5000))
            cap = str(rng.choice(["16384", "0", str(max(1, n_hand - 1)), "1"]))
            env = {"PNX_NNLS_HOST_CHUNK": str(chunk), "PNX_NNLS_PEAKS_CHUNK": str(chunk), "PNX_NNLS_DEFER_CAP": cap,
                   "PNX_NNLS_PEAKS_RING": str(int(rng.random() < 0.85)), "PNX_HOST_TOUCHERS": str(int(rng.integers(0, 4))),
                   "PNX_HOST_RAMP": str(int(rng.random() < 0.7))}
            os.environ.update(env)
            many = plan.solve_peaks(yy, bins, **pk) if peaks else plan.solve(yy, 250)
            voxels += n
            diff = [k for k in one if (one[k] is None) != (many[k] is None) or
                    (one[k] is not None and not np.array_equal(one[k], many[k], equal_nan=True))]
            if diff:
                bad += 1
                say(f"[case {c}] FAIL {diff}: rows [{a}, {a + n}) handed over {n_hand} f32={f32} peaks={peaks} {env}")
    finally:
        plan.close()
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    say(f"{n_cases} cases, {voxels} voxels, {handed} handed-over voxels among them: {bad} failing")
    return {"fuzzer": "nnls_host", "n_cases": n_cases, "seed": seed, "voxels": voxels, "handed_over_voxels": handed, "failing_cases": bad}


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    out = run(int(a[0]) if a else 40, int(a[1]) if len(a) > 1 else 0)
    if "--json" in sys.argv:
        import json
        from pyneapple_amd import _build
        out["source_ids"] = _build.source_ids()
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    sys.exit(1 if out["failing_cases"] else 0)
