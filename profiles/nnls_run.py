#!/usr/bin/env python3
"""One device-resident NNLS solve of the C4 workload (PNX_RUN_VOXELS voxels, default 2^18) after one warm-up: the process
rocprofv3 wraps for counter passes on a kernel variant (PNX_LIB).  No oracle, no timing claims."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import api, synth
n = int(os.environ.get("PNX_RUN_VOXELS", 1 << 18))
dev = torch.device("cuda", 0)
bins, basis, reg = synth.nnls_matrices(32)
plan = api.NnlsPlan(basis, reg, 0)
_, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01, scale=1000.0)
coeff = torch.empty((n, 250), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    t = time.perf_counter(); plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize()
    print("solve ms", (time.perf_counter() - t) * 1e3, "mean iters", float(it.double().mean()), flush=True)
