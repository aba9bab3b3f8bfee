#!/usr/bin/env python3
"""HipSegmentationWiseFitter on a 256x256x64x32 volume with 8 labels: wall time of fit() and of its label reduction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
from pyneapple_amd.models import TriExpModel
from pyneapple_amd.segmented import HipSegmentationWiseFitter, _label_positions
from pyneapple_amd.solvers import HipCurveFitSolver

shape = (256, 256, 64)
n = int(np.prod(shape))
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
img = y.reshape(*shape, 32)
seg = (np.arange(n).reshape(shape) // (n // 8)).astype(np.int64)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
solver = HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)),
                           bounds={k: (a, c) for k, a, c in zip(names, lo, hi)})
f = HipSegmentationWiseFitter(solver)
for rep in range(3):
    t = time.perf_counter(); f.fit(b, img, segmentation=seg); dt = time.perf_counter() - t
    print(f"fit: {dt * 1e3:.0f} ms, labels {f.segment_labels.tolist()}, D3 {np.round(f.fitted_params_['D3'], 6).tolist()}", flush=True)
labels, inv = _label_positions(seg)
t = time.perf_counter(); s, c = api.label_sums(y, inv, labels.size); dt = time.perf_counter() - t
t2 = time.perf_counter(); ref = np.stack([np.bincount(inv, weights=y[:, k], minlength=labels.size) for k in range(32)], axis=1); dt2 = time.perf_counter() - t2
print(f"label sums: device (host arrays in/out) {dt * 1e3:.0f} ms, numpy {dt2 * 1e3:.0f} ms, max rel diff {np.abs(s / ref - 1).max():.2e}")
