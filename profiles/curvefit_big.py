#!/usr/bin/env python3
"""Kernel rate without the straggler tail: 16.8 M triexp voxels, no pcov, best of 3 (PNX_LIB selects a variant build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
n = 1 << 24
b, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01)
popt = torch.empty((5, n), dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); nf = torch.empty(n, dtype=torch.int32, device=dev); cost = torch.empty(n, dtype=torch.float64, device=dev)
opts = api.make_opts("tri_reduced", 32, [], False, False, 250, 1e-8, 1e-8, 1e-8, "fd", 0, 0.0, 0.0)
s = torch.cuda.current_stream().cuda_stream
api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, None, st, nf, cost, 0, s); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, None, st, nf, cost, 0, s); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(os.path.basename(os.environ.get("PNX_LIB", "product")), f"{best:.2f} ms  {n / best / 1e3:.1f} M voxels/s")
