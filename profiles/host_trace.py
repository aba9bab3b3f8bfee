import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
r = api.curvefit("tri_reduced", b, y, p0, lo, hi); del r
os.environ["PNX_HOST_TRACE"] = "1"
t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi); print("ms", (time.perf_counter() - t) * 1e3)
