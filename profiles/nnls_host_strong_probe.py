#!/usr/bin/env python3
"""NNLS from numpy arrays with a regulariser stronger than the reference's: several per cent of the voxels are handed over, more
than the deferred pass's side buffer holds (16 384), so the call solves them in batches (up to round 4 it ran twice).
Device-resident time of the same voxels beside it.   usage: python3 profiles/nnls_host_strong_probe.py [order mu n_vox]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
order = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mu = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 21
dev = torch.device("cuda", 0)
cfg = dict(synth.NNLS_CFG, reg_order=order, mu=mu)
bins, basis, reg = synth.nnls_matrices(32, cfg)
plan = api.NnlsPlan(basis, reg, 0)
_, yt = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01, scale=1000.0)
coeff = torch.empty((n, 250), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
plan.solve_device(n, yt, 250, coeff, rn, st, it, s); torch.cuda.synchronize()
t = time.perf_counter(); plan.solve_device(n, yt, 250, coeff, rn, st, it, s); torch.cuda.synchronize(); t_dev = time.perf_counter() - t
y = yt.cpu().numpy()
r = plan.solve(y[: 1 << 16], 250)
ts = []
for _ in range(2):
    del r
    t = time.perf_counter(); r = plan.solve(y, 250); ts.append(time.perf_counter() - t)
same = bool((torch.from_numpy(r["coefficients"][: 1 << 18]).to(dev) == coeff[: 1 << 18]).all().item())
over = int(((r["coefficients"] > 0).sum(axis=1) > 128).sum())
print(f"order {order} mu {mu}: {n} voxels, final supports beyond 128 bins on {over} ({100.0 * over / n:.2f} %); device resident {t_dev * 1e3:.1f} ms, "
      f"numpy in / out {min(ts) * 1e3:.1f} ms ({[round(x * 1e3, 1) for x in ts]}), equal to the resident result: {same}", flush=True)
