"""SegmentedFitter's second step from numpy arrays: biexp with D1 fixed per voxel, 4 Mi voxels x 32 b-values, streamed host path
against the chunk ring."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
b, y, P = synth.make_numpy("bi_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("bi_reduced")
free = [0, 2]
kw = dict(fixed_idx=[1], fixed_vals=P["D1"][None, :].copy(), jac="analytic")
for mode in ("0", "1", "0", "1"):
    os.environ["PNX_HOST_STREAM"] = mode
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = api.curvefit("bi_reduced", b, y, p0[free], lo[free], hi[free], **kw); ts.append(time.perf_counter() - t)
        ok = float((r["status"] > 0).mean()); del r
    print(f"{'streamed' if mode == '1' else 'chunk ring'}: {[round(t * 1e3, 1) for t in ts]} ms  best {n / min(ts) / 1e6:.1f} M voxels/s  converged {ok:.5f}", flush=True)
