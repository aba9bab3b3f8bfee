#!/usr/bin/env python3
"""Launch-bound case (BASELINE config C1: monoexp, 32x32x1 voxels x 16 b-values): per-call time of the device-pointer entry
point called directly from Python versus the same work captured once into a HIP graph and replayed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
n_vox, n_b = 1024, 16
b, y = synth.make_torch("mono", n_vox, n_b, dev, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("mono")
popt = torch.empty((2, n_vox), dtype=torch.float64, device=dev); pcov = torch.empty((n_vox, 2, 2), dtype=torch.float64, device=dev)
st = torch.empty(n_vox, dtype=torch.int8, device=dev); nf = torch.empty(n_vox, dtype=torch.int32, device=dev); c = torch.empty(n_vox, dtype=torch.float64, device=dev)
o = api.make_opts("mono", n_b)
def enqueue(s): api.curvefit_device(o, n_vox, b, y, p0, lo, hi, None, popt, pcov, st, nf, c, 0, s)
s0 = torch.cuda.current_stream().cuda_stream
enqueue(s0); torch.cuda.synchronize()
reps = 500
t = time.perf_counter()
for _ in range(reps): enqueue(s0)
torch.cuda.synchronize(); direct = (time.perf_counter() - t) / reps
g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for _ in range(10): enqueue(torch.cuda.current_stream().cuda_stream)   # ten fits per replay
torch.cuda.current_stream().wait_stream(side)
g.replay(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps // 10): g.replay()
torch.cuda.synchronize(); graph = (time.perf_counter() - t) / reps
print(f"monoexp 1024 voxels: direct call {direct * 1e6:.1f} us per fit ({n_vox / direct / 1e6:.1f} M voxels/s), graph replay (10 fits per graph) {graph * 1e6:.1f} us per fit ({n_vox / graph / 1e6:.1f} M voxels/s)")
