#!/usr/bin/env python3
"""Odd against even numbers of b-values: the curve-fit kernel refills a lane's signal row with 16-byte global -> LDS loads, which
need rows of 16-byte pairs (an even number of b-values); an odd number takes the synchronous row copy.  Device resident, triexp
reduced, FD Jacobian, pcov on.   usage: python3 profiles/odd_nb_probe.py [n_vox]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
n_vox = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21
for model, nbs in (("tri_reduced", (32, 31, 30, 17, 16, 15)), ("bi_reduced", (24, 23, 11, 10)), ("mono", (16, 15, 7, 6))):
    for n_b in nbs:
        b, y = synth.make_torch(model, n_vox, n_b, dev, sigma=0.01, scale=1.0)
        names, p0, lo, hi = synth.shared_arrays(model)
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
        ev = nf.double().sum().item()
        print(f"{model:12s} n_b={n_b:3d}: {n_vox / ms / 1e3:7.1f} M voxels/s ({ms:.2f} ms), mean nfev {nf.float().mean().item():.1f}, {ev * n_b / ms / 1e6:.1f} G row-evaluations/s", flush=True)
