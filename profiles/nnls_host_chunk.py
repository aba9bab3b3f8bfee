"""C4 through the host entry point (numpy in, numpy (n_vox, 250) spectra out) for a range of host chunk sizes: every chunk is one
launch of the block kernel plus one hand-over pass, so few large chunks beat many small ones until the last chunk's
download (2 KB per voxel) is left exposed."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = int(os.environ.get("PNX_RUN_VOXELS", 256 * 256 * 64))
bins, basis, reg = synth.nnls_matrices(32)
_, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01, scale=1000.0)
plan = api.NnlsPlan(basis, reg, 0)
r = plan.solve(y[:65536], 250); del r
for chunk, cap in ((3 << 18, 0), (1 << 17, 16384), (1 << 18, 16384), (1 << 19, 16384), (3 << 18, 16384), (1 << 20, 16384)):
    os.environ["PNX_NNLS_HOST_CHUNK"] = str(chunk)
    os.environ["PNX_NNLS_DEFER_CAP"] = str(cap)  # 0: one hand-over pass per chunk (the behaviour before the deferral)
    ts = []
    for _ in range(2):
        t = time.perf_counter(); r = plan.solve(y, 250); ts.append(time.perf_counter() - t); del r
    print(f"chunk {chunk >> 10}k, hand-over {'deferred to the end of the call' if cap else 'per chunk'}: {[round(t * 1e3) for t in ts]} ms -> {n / min(ts) / 1e6:.2f} M voxels/s", flush=True)
