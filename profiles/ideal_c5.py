#!/usr/bin/env python3
"""BASELINE config 5: IDEAL multi-resolution triexp (3 levels) on a 256x256x64x32 volume through HipIDEALFitter +
HipCurveFitSolver (host arrays in and out, like the reference's IDEALFitter).  Prints the wall time per stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import synth
from pyneapple_amd.ideal import HipIDEALFitter
from pyneapple_amd.models import TriExpModel
from pyneapple_amd.solvers import HipCurveFitSolver

shape = (256, 256, 64) if len(sys.argv) < 2 else tuple(int(a) for a in sys.argv[1].split("x"))
n = int(np.prod(shape))
t0 = time.perf_counter()
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
img = y.reshape(*shape, 32)
print(f"synthetic volume {shape} x 32: {time.perf_counter() - t0:.1f} s", flush=True)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
solver = HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)),
                           bounds={k: (a, c) for k, a, c in zip(names, lo, hi)})
steps = np.array([[shape[0] // 4, shape[1] // 4], [shape[0] // 2, shape[1] // 2], [shape[0], shape[1]]])
fit = HipIDEALFitter(solver, steps, {k: 0.5 for k in names}, interpolation_method="cubic")
for rep in range(2):
    t0 = time.perf_counter()
    fit.fit(b, img)
    dt = time.perf_counter() - t0
    conv = float(np.mean(solver.diagnostics_["status"] > 0))
    print(f"IDEAL 3 levels {steps.tolist()}: {dt:.2f} s total, {n / dt / 1e6:.2f} M final-level voxels/s, converged {conv:.4f}; stages {getattr(fit, 'stage_times_', None)}", flush=True)
