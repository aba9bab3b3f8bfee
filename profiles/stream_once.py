"""C3 from numpy arrays, three calls through the streamed host path (for a rocprofv3 --kernel-trace of the STREAM kernel):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 profiles/stream_once.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
r = api.curvefit("tri_reduced", b, y, p0, lo, hi)
for i in range(3):
    del r
    t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi); dt = time.perf_counter() - t
    print(f"call {i}: {dt * 1e3:.2f} ms, {n / dt / 1e6:.1f} M voxels/s, converged {float((r['status'] > 0).mean()):.6f}", flush=True)
