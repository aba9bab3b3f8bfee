#!/usr/bin/env python3
"""End-to-end time of the plugin fitter on the C3 volume (numpy image in, FitResult out): HipPixelWiseFitter +
HipCurveFitSolver, stages timed separately."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import synth
from pyneapple_amd.fitters import HipPixelWiseFitter
from pyneapple_amd.models import TriExpModel
from pyneapple_amd.solvers import HipCurveFitSolver

shape = (256, 256, 64)
n = int(np.prod(shape))
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
img = y.reshape(*shape, 32)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
solver = HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)),
                           bounds={k: (a, c) for k, a, c in zip(names, lo, hi)})
for rep in range(3):
    t0 = time.perf_counter(); solver.fit(b, y); t1 = time.perf_counter()
    f = HipPixelWiseFitter(solver); f.fit(b, img); t2 = time.perf_counter()
    maps = f.parameter_maps(); t3 = time.perf_counter()
    r = f.results_
    print(f"solver.fit {t1 - t0:.3f} s | fitter.fit (solver + R^2 + indices) {t2 - t1:.3f} s | float32 maps {t3 - t2:.3f} s | "
          f"converged {r.convergence_rate:.4f} mean R^2 {r.mean_r_squared:.5f}", flush=True)
