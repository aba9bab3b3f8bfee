#!/usr/bin/env python3
"""C4 from numpy arrays: the deferred hand-over pass behind the last chunk's solve on a stream of its own (overlapping that
chunk's download) against the pass after the ring has drained (PNX_NNLS_DEFER_OVERLAP=0), alternating in one process."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
bins, basis, reg = synth.nnls_matrices(32)
_, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01, scale=1000.0)
plan = api.NnlsPlan(basis, reg, 0)
r = plan.solve(y[: 1 << 16], 250)
ref = None
for rep in range(4):
    for flag in ("1", "0"):
        os.environ["PNX_NNLS_DEFER_OVERLAP"] = flag
        del r
        t = time.perf_counter(); r = plan.solve(y, 250); dt = time.perf_counter() - t
        if ref is None: ref = r["coefficients"][:1 << 18].copy()
        print(f"overlap={flag}: {dt * 1e3:.1f} ms = {n / dt / 1e6:.3f} M voxels/s, identical {bool(np.array_equal(r['coefficients'][:1 << 18], ref))}", flush=True)
