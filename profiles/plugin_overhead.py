"""What the Pyneapple plugin classes add on top of the C-ABI call: HipCurveFitSolver.fit / HipNNLSSolver.fit / fit_peaks on
the C3 / C4 volumes (numpy in, solver state out) next to api.curvefit / NnlsPlan.solve on the same arrays."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
from pyneapple_amd.models import NNLSModel, TriExpModel
from pyneapple_amd.solvers import HipCurveFitSolver, HipNNLSSolver

n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t); del r
    return [round(t * 1e3, 1) for t in ts]


api.curvefit("tri_reduced", b, y, p0, lo, hi)
print("C3 api.curvefit              ", best(lambda: api.curvefit("tri_reduced", b, y, p0, lo, hi)), "ms", flush=True)
solver = HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)),
                           bounds={k: (l, h) for k, l, h in zip(names, lo, hi)})
print("C3 HipCurveFitSolver.fit     ", best(lambda: solver.fit(b, y)), "ms", flush=True)
del solver
if "--nnls" in sys.argv:
    cfg = synth.NNLS_CFG
    bins, basis, reg = synth.nnls_matrices(32)
    _, yn, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01, scale=1000.0)
    plan = api.NnlsPlan(basis, reg, 0)
    plan.solve(yn[:65536], 250)
    print("C4 NnlsPlan.solve            ", best(lambda: plan.solve(yn, 250), 2), "ms", flush=True)
    plan.close()
    s = HipNNLSSolver(model=NNLSModel(d_range=cfg["d_range"], n_bins=cfg["n_bins"]), reg_order=cfg["reg_order"], mu=cfg["mu"], max_iter=250)
    bv = synth.bvalues(32)
    print("C4 HipNNLSSolver.fit         ", best(lambda: s.fit(bv, yn), 2), "ms", flush=True)
    print("C4 HipNNLSSolver.fit_peaks   ", best(lambda: s.fit_peaks(bv, yn, height=0.1, cutoffs=[(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]), 2), "ms", flush=True)
