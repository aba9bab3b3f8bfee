"""Experiment, not kept: result arrays with transparent huge pages (madvise MADV_HUGEPAGE on np.empty, tried as api._result
with PNX_HUGEPAGES; the hosts run THP in `madvise` mode).  C3 from numpy arrays 38-42 ms and C4 648-658 ms either way, and
dropping the 1 GB / 8.4 GB result costs 63-85 ms / 340-540 ms either way: neither the first-touch faults (helper threads take
them in parallel already) nor the page size is what makes releasing a downloaded result slow.  The script needs that
experimental api._result to show a difference; it is kept for the numbers above (gpurun_out/st/huge.log of round 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth

n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
bins, basis, reg = synth.nnls_matrices(32)
_, yn, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01, scale=1000.0)
plan = api.NnlsPlan(basis, reg, 0)
plan.solve(yn[:65536], 250)
api.curvefit("tri_reduced", b, y, p0, lo, hi)
for hp in ("0", "1", "0", "1"):
    os.environ["PNX_HUGEPAGES"] = hp
    ts, td = [], []
    for _ in range(4):
        t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi); ts.append(time.perf_counter() - t)
        t = time.perf_counter(); del r; td.append(time.perf_counter() - t)
    print(f"huge pages {hp}: C3 call {[round(t * 1e3, 1) for t in ts]} ms, dropping the result {[round(t * 1e3, 1) for t in td]} ms", flush=True)
    ts, td = [], []
    for _ in range(2):
        t = time.perf_counter(); r = plan.solve(yn, 250); ts.append(time.perf_counter() - t)
        t = time.perf_counter(); del r; td.append(time.perf_counter() - t)
    print(f"huge pages {hp}: C4 call {[round(t * 1e3) for t in ts]} ms, dropping the result {[round(t * 1e3) for t in td]} ms", flush=True)
