"""Experiment behind the rule \"no streamed launch for the instantiations that need (almost) all 512 registers\": tri_full (six free
parameters) streamed with 8 / 64 / 200 of the 256 CUs left free for the runtime's copy kernels, the conversions and the epilogue
(needed the knob PNX_STREAM_RESERVE_CUS of that build; profiles/r03_i_stream_tight_probe.txt: with 8 free CUs the upload sat
behind the kernel until its poll limit, with 64 it ran).  The product keeps such fits on the chunk ring."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import numpy as np
from pyneapple_amd import api, synth
n_vox = 150007
b, y, _ = synth.make_numpy("tri_reduced", n_vox, 32, sigma=0.01, seed=21)
p0 = np.array([200.0, 0.05, 300.0, 0.005, 500.0, 0.001]); lo = np.array([0.0, 0.01, 0.0, 2e-3, 0.0, 1e-5]); hi = np.array([2000.0, 0.5, 2000.0, 0.01, 2000.0, 2e-3])
os.environ["PNX_STREAM_GRANULE_SHIFT"] = "14"
os.environ["PNX_STREAM_SPINS"] = "60000"
for dt in (np.float64, np.float32):
    yy = (y * 1000.0).astype(dt)
    for res in ("8", "64", "200"):
        os.environ["PNX_STREAM_RESERVE_CUS"] = res
        t = time.perf_counter(); r = api.curvefit("tri_full", b, yy, p0, lo, hi); dt_ = time.perf_counter() - t
        print(f"{dt.__name__} reserve {res}: {dt_ * 1e3:.1f} ms", flush=True)
