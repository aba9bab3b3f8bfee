#!/usr/bin/env python3
"""cProfile of HipPixelWiseFitter.fit on the C3 volume (second call): where the host-side time goes."""
import cProfile, os, pstats, sys
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
f = HipPixelWiseFitter(solver)
f.fit(b, img)
pr = cProfile.Profile()
pr.enable()
f.fit(b, img)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
