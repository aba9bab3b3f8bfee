"""C3 through the float32 host entry point (pnx_curvefit_batch_f32, numpy float32 in and out) for a range of chunk sizes /
slot counts of the host pipeline -- the float32 transfers are half as long as the float64 ones, so the chunk can grow (fewer
persistent-kernel drain tails) at the same exposed transfer latency.  Prints PNX_HOST_TRACE timing lines when set."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
y32 = y.astype(np.float32); del y
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
api.curvefit("tri_reduced", b, y32[:4096], p0, lo, hi)
for chunk in (3 << 18, 1 << 20, 3 << 19, 1 << 21, 1 << 22):
    for w in (3, 4):
        os.environ["PNX_HOST_CHUNK"] = str(chunk); os.environ["PNX_HOST_SLOTS"] = str(w)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y32, p0, lo, hi); ts.append(time.perf_counter() - t); del r
        print(f"f32 chunk {chunk>>10}k slots {w}: best {1e3*min(ts):.1f} ms median {1e3*sorted(ts)[1]:.1f} ms -> {n/min(ts)/1e6:.1f} M voxels/s", flush=True)
