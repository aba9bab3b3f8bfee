#!/usr/bin/env python3
"""Throughput of the QR-form NNLS kernel (unregularised, the reference's default reg_order = 0): 250 bins, 32 measurements."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
bins, basis, reg = synth.nnls_matrices(32)
n = 1 << 19
_, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01, scale=1000.0)
for name, r in (("reg_order 0 (QR kernel)", None), ("reg_order 2 (Gram kernel)", reg)):
    plan = api.NnlsPlan(basis, r, 0)
    coeff = torch.empty((n, 250), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize()
    t = time.perf_counter(); plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{name}: {n / dt / 1e6:.2f} M voxels/s ({dt * 1e3:.0f} ms), mean iterations {it.double().mean().item():.1f}, converged {(st == 1).double().mean().item():.4f}, mean support {(coeff > 0).sum(1).double().mean().item():.1f}", flush=True)
    plan.close()
