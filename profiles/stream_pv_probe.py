"""Per-voxel start values and bounds from numpy arrays (what the reference's IDEAL fitter hands a solver plugin on its last
level): triexp, 4 Mi voxels x 32 b-values, p0 / lo / hi each (5, n_vox); streamed host path against the chunk ring."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0s, los, his = synth.shared_arrays("tri_reduced")
rng = np.random.default_rng(0)
p0 = np.ascontiguousarray(np.tile(p0s[:, None], (1, n)) * rng.uniform(0.95, 1.05, (5, n)))
lo = np.ascontiguousarray(np.tile(los[:, None], (1, n)))
hi = np.ascontiguousarray(np.tile(his[:, None], (1, n)))
for mode in ("0", "1", "0", "1"):
    os.environ["PNX_HOST_STREAM"] = mode
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi); ts.append(time.perf_counter() - t)
        ok = float((r["status"] > 0).mean()); del r
    print(f"{'streamed' if mode == '1' else 'chunk ring'}: {[round(t * 1e3, 1) for t in ts]} ms  best {n / min(ts) / 1e6:.1f} M voxels/s  converged {ok:.5f}", flush=True)
os.environ["PNX_HOST_TRACE"] = "1"
r = api.curvefit("tri_reduced", b, y, p0, lo, hi); del r
for piece in (1 << 18, 1 << 19):
    os.environ["PNX_STREAM_IN_CHUNK"] = str(piece)
    os.environ.pop("PNX_HOST_TRACE", None)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi); ts.append(time.perf_counter() - t); del r
    print(f"streamed, upload pieces of {piece >> 10} Ki voxels: {[round(t * 1e3, 1) for t in ts]} ms", flush=True)
