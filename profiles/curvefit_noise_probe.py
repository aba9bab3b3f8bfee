#!/usr/bin/env python3
"""The C3 curve fit against the noise level of the signal (the benchmark's is 1 %): voxels/s, evaluations per voxel, share of
the voxels that reach the limit of 250 evaluations.  2^21 voxels, device resident, FD Jacobian, pcov on.
usage: python3 profiles/curvefit_noise_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
n_vox, n_b, model = 1 << 21, 32, "tri_reduced"
names, p0, lo, hi = synth.shared_arrays(model)
n = len(p0)
popt = torch.empty((n, n_vox), dtype=torch.float64, device=dev); pcov = torch.empty((n_vox, n, n), dtype=torch.float64, device=dev)
st = torch.empty(n_vox, dtype=torch.int8, device=dev); nf = torch.empty(n_vox, dtype=torch.int32, device=dev); c = torch.empty(n_vox, dtype=torch.float64, device=dev)
o = api.make_opts(model, n_b, max_nfev=250, ftol=1e-8, jac="fd")
s = torch.cuda.current_stream().cuda_stream
for sigma in (0.0, 0.002, 0.01, 0.03, 0.05, 0.1):
    b, y = synth.make_torch(model, n_vox, n_b, dev, sigma=sigma, scale=1.0)
    def run():
        api.curvefit_device(o, n_vox, b, y, p0, lo, hi, None, popt, pcov, st, nf, c, 0, s)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); run(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    ev = nf.double()
    print(f"sigma={sigma:5.3f}: {n_vox / ms / 1e3:7.1f} M voxels/s ({ms:.2f} ms), mean nfev {ev.mean().item():.1f}, p99 {torch.quantile(ev[:1<<20], 0.99).item():.0f}, max {int(ev.max().item())}, "
          f"limit reached {(st == 0).double().mean().item():.5f}, converged {(st > 0).double().mean().item():.5f}, {ev.sum().item() / ms / 1e6:.2f} G evaluations/s", flush=True)
