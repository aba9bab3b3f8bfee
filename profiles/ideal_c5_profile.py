#!/usr/bin/env python3
"""cProfile of the second HipIDEALFitter.fit call on BASELINE config 5 (where the host-side time goes)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import synth
from pyneapple_amd.ideal import HipIDEALFitter
from pyneapple_amd.models import TriExpModel
from pyneapple_amd.solvers import HipCurveFitSolver

shape = (256, 256, 64)
n = int(np.prod(shape))
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
img = y.reshape(*shape, 32)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
solver = HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)),
                           bounds={k: (a, c) for k, a, c in zip(names, lo, hi)})
steps = np.array([[64, 64], [128, 128], [256, 256]])
fit = HipIDEALFitter(solver, steps, {k: 0.5 for k in names}, interpolation_method="cubic")
fit.fit(b, img)
pr = cProfile.Profile()
pr.enable()
fit.fit(b, img)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
import time
t = time.perf_counter(); fit.fit(b, img); dt = time.perf_counter() - t
print(f"third call {dt * 1e3:.1f} ms; per level (prepare, fit + statistics, map download) seconds: {fit.stage_times_}")
