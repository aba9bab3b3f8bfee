"""C4 from numpy arrays with the chunk ring's timeline (PNX_HOST_TRACE=1) next to the wall time of the call: what the call
spends outside the ring (slot allocation, deferred hand-over pass, patching, frees), and whether the number of page-touch
helpers matters (it does not: 649-656 ms with 1-4; the second full-size call of a process starts 10 ms late, later ones after
2 ms; first launch to last kernel end is the resident 628 ms)."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
bins, basis, reg = synth.nnls_matrices(32)
_, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01, scale=1000.0)
plan = api.NnlsPlan(basis, reg, 0)
r = plan.solve(y[:65536], 250); del r
r = plan.solve(y, 250); del r
os.environ["PNX_HOST_TRACE"] = "1"
t = time.perf_counter(); r = plan.solve(y, 250); dt = time.perf_counter() - t
print(f"wall {dt * 1e3:.1f} ms", file=sys.stderr, flush=True)
del r
for touchers in ("2", "1", "4", "2", "3", "1", "4", "2"):
    os.environ["PNX_HOST_TOUCHERS"] = touchers
    t = time.perf_counter(); r = plan.solve(y, 250); dt = time.perf_counter() - t; del r
    print(f"touchers {touchers}: wall {dt * 1e3:.1f} ms", file=sys.stderr, flush=True)
