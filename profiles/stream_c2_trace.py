"""C2 (biexp 128x128x32 x 24 b-values) from numpy arrays, call by call with the streamed path's trace."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 128 * 128 * 32
b, y, _ = synth.make_numpy("bi_reduced", n, 24, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("bi_reduced")
os.environ["PNX_HOST_TRACE"] = "1"
r = api.curvefit("bi_reduced", b, y, p0, lo, hi)
for i in range(8):
    del r
    t = time.perf_counter(); r = api.curvefit("bi_reduced", b, y, p0, lo, hi); dt = time.perf_counter() - t
    print(f"call {i}: {dt * 1e3:.2f} ms", file=sys.stderr, flush=True)
