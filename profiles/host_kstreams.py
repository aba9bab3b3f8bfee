import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
r = api.curvefit("tri_reduced", b, y, p0, lo, hi); del r
for ks, slots, chunk in ((2, 3, 3 << 18), (3, 4, 3 << 18), (3, 5, 1 << 19), (4, 6, 1 << 19), (3, 4, 1 << 19), (2, 4, 1 << 19)):
    os.environ["PNX_HOST_KSTREAMS"] = str(ks); os.environ["PNX_HOST_SLOTS"] = str(slots); os.environ["PNX_HOST_CHUNK"] = str(chunk)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi); ts.append(time.perf_counter() - t); del r
    print(f"kstreams {ks} slots {slots} chunk {chunk >> 10}k: {[round(t * 1e3, 1) for t in ts]} ms", flush=True)
