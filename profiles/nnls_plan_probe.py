#!/usr/bin/env python3
"""Cost of creating (and destroying) an NNLS plan: the plugin makes one per fit() call, so a fit of a few voxels pays it in
full.  Since the last part of round 4 a block-kernel plan holds 1 GB of slabs for the Gram-form kernel (the hand-over pass runs on
the full grid).   usage: python3 profiles/nnls_plan_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth, _lib
_lib.load()
for order in (2, 0):
    cfg = dict(synth.NNLS_CFG, reg_order=order)
    bins, basis, reg = synth.nnls_matrices(32, cfg)
    _, y, _ = synth.make_numpy("tri_reduced", 64, 32, sigma=0.01, scale=1000.0)
    api.NnlsPlan(basis, reg, 0).close()
    tc, td, ts = [], [], []
    for _ in range(5):
        t0 = time.perf_counter(); plan = api.NnlsPlan(basis, reg, 0); t1 = time.perf_counter()
        plan.solve(y, 250); t2 = time.perf_counter()
        plan.close(); t3 = time.perf_counter()
        tc.append(t1 - t0); ts.append(t2 - t1); td.append(t3 - t2)
    print(f"reg_order={order}: create {np.median(tc) * 1e3:.2f} ms, solve of 64 voxels from numpy {np.median(ts) * 1e3:.2f} ms, destroy {np.median(td) * 1e3:.2f} ms", flush=True)
