#!/usr/bin/env python3
"""Curve-fit kernel time vs batch size up to 16 M voxels (no pcov), and the nfev distribution: is the C3 step throughput- or
tail-bound?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
nn = len(names)
s = torch.cuda.current_stream().cuda_stream
opts = api.make_opts("tri_reduced", 32, [], False, False, 250, 1e-8, 1e-8, 1e-8, "fd", 0, 0.0, 0.0)
for n in (1 << 22, 1 << 23, 1 << 24):
    b, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01)
    popt = torch.empty((nn, n), dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int8, device=dev); nf = torch.empty(n, dtype=torch.int32, device=dev); cost = torch.empty(n, dtype=torch.float64, device=dev)
    for _ in range(2):
        api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, None, st, nf, cost, 0, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, None, st, nf, cost, 0, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(n, f"{ms:.3f} ms  {n / ms / 1e3:.1f} M voxels/s", flush=True)
    if n == 1 << 22:
        v = nf.cpu().numpy()
        q = np.quantile(v, [0.5, 0.9, 0.99, 0.999, 0.9999, 0.99999])
        print("nfev quantiles 50/90/99/99.9/99.99/99.999 %:", q.tolist(), "max", int(v.max()), "count >= 100:", int((v >= 100).sum()), "count >= 200:", int((v >= 200).sum()))
        idx = np.nonzero(v >= 150)[0]
        print("positions of nfev >= 150 (fraction of the volume):", np.round(idx / n, 3).tolist()[:40])
    del y, popt, st, nf, cost
