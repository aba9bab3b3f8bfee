#!/usr/bin/env python3
"""Device-resident throughput of every model's kernel (1 Mi voxels, FD Jacobian, pcov on), one line per model.
usage: python3 profiles/model_probe.py [n_vox]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
CASES = {  # model: (base generator, n_b, extra p0/lo/hi for the added parameters)
    "mono": ("mono", 16), "bi_reduced": ("bi_reduced", 24), "bi_s0": ("bi_reduced", 24), "bi_full": ("bi_reduced", 24),
    "tri_reduced": ("tri_reduced", 32), "tri_s0": ("tri_reduced", 32), "tri_full": ("tri_reduced", 32)}
for model, (base, n_b) in CASES.items():
    b, y = synth.make_torch(base, n_vox, n_b, dev, sigma=0.01, scale=1.0 if model in ("mono", "bi_reduced", "tri_reduced") else 1000.0)
    names, p0, lo, hi = synth.shared_arrays(base)
    if model.endswith("_s0"):
        p0, lo, hi = np.append(p0, 1000.0), np.append(lo, 1.0), np.append(hi, 5000.0)
    if model == "bi_full":
        p0, lo, hi = np.array([200.0, 0.01, 800.0, 0.001]), np.array([0.0, 1e-3, 0.0, 1e-5]), np.array([2000.0, 0.1, 2000.0, 5e-3])
    if model == "tri_full":
        p0 = np.array([200.0, 0.05, 300.0, 0.005, 500.0, 0.001]); lo = np.array([0.0, 0.01, 0.0, 2e-3, 0.0, 1e-5]); hi = np.array([2000.0, 0.5, 2000.0, 0.01, 2000.0, 2e-3])
    if model == "mono":
        pass
    n = len(p0)
    popt = torch.empty((n, n_vox), dtype=torch.float64, device=dev); pcov = torch.empty((n_vox, n, n), dtype=torch.float64, device=dev)
    st = torch.empty(n_vox, dtype=torch.int8, device=dev); nf = torch.empty(n_vox, dtype=torch.int32, device=dev); c = torch.empty(n_vox, dtype=torch.float64, device=dev)
    o = api.make_opts(model, n_b)
    s = torch.cuda.current_stream().cuda_stream
    def run():
        api.curvefit_device(o, n_vox, b, y, p0, lo, hi, None, popt, pcov, st, nf, c, 0, s)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); run(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{model:12s} n={n} n_b={n_b}: {n_vox / ms / 1e3:7.1f} M voxels/s ({ms:.2f} ms), converged {(st > 0).float().mean().item():.4f}, mean nfev {nf.float().mean().item():.1f}", flush=True)
