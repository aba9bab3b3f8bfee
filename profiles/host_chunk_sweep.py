import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import numpy as np
from pyneapple_amd import api, synth, _lib
from pyneapple_amd._lib import load, ptr, check, MEM_HOST
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
api.curvefit("tri_reduced", b, y[:4096], p0, lo, hi)
o = api.make_opts("tri_reduced", 32, [], False, False, 250, 1e-8, 1e-8, 1e-8, "fd", 0, 0.0, 0.0)
nn = 5
def run(want_pcov=True):
    popt = np.empty((nn, n)); pcov = np.empty((n, nn, nn)) if want_pcov else None
    status = np.empty(n, np.int8); nfev = np.empty(n, np.int32); cost = np.empty(n)
    t1 = time.perf_counter()
    check(load().pnx_curvefit_batch_f64(C.byref(o), n, ptr(b), ptr(y), ptr(p0), ptr(lo), ptr(hi), None, ptr(popt), ptr(pcov), ptr(status), ptr(nfev), ptr(cost), MEM_HOST, 0, None))
    return time.perf_counter() - t1
for chunk in (1 << 18, 3 << 17, 1 << 19, 3 << 18, 1 << 20, 1 << 21):
    for w in (3, 4, 6):
        os.environ["PNX_HOST_CHUNK"] = str(chunk); os.environ["PNX_HOST_SLOTS"] = str(w)
        ts = [run() for _ in range(3)]
        print(f"chunk {chunk>>10}k slots {w}: best {1e3*min(ts):.1f} ms  median {1e3*sorted(ts)[1]:.1f} ms -> {n/min(ts)/1e6:.1f} M voxels/s", flush=True)
