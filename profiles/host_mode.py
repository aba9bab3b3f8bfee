#!/usr/bin/env python3
"""Throughput of the host-pointer (PNX_MEM_HOST) entry points, i.e. what the plugin classes see: pageable numpy
arrays in, numpy arrays out, PCIe transfers included.  usage: python profiles/host_mode.py [n_vox]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
api.curvefit("tri_reduced", b, y[:4096], p0, lo, hi)  # warm-up
for want_pcov in (True, False):
    best = 1e9
    for _ in range(3):  # results are freed before the clock starts (unmapping 1 GB costs tens of ms)
        t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y, p0, lo, hi, want_pcov=want_pcov); dt = time.perf_counter() - t
        best = min(best, dt); conv = np.mean(r['status'] > 0); del r
    dt = best
    print(f"curvefit host mode pcov={want_pcov}: {n/dt/1e6:.1f} M voxels/s ({dt*1e3:.0f} ms, converged {conv:.4f})", flush=True)
y32 = y.astype(np.float32)
best = 1e9
for _ in range(3):
    t = time.perf_counter(); r = api.curvefit("tri_reduced", b, y32, p0, lo, hi); dt = time.perf_counter() - t
    best = min(best, dt); conv = np.mean(r['status'] > 0); del r
print(f"curvefit host mode, float32 storage (pnx_curvefit_batch_f32), pcov=True: {n/best/1e6:.1f} M voxels/s ({best*1e3:.0f} ms, converged {conv:.4f})", flush=True)
bins, basis, reg = synth.nnls_matrices(32)
m = min(n, 1 << 20)
plan = api.NnlsPlan(basis, reg, 0)
plan.solve(y[:4096] * 1000.0)
t = time.perf_counter(); r = plan.solve(y[:m] * 1000.0, 250); dt = time.perf_counter() - t
print(f"nnls host mode: {m/dt/1e6:.2f} M voxels/s ({dt*1e3:.0f} ms)", flush=True)
del r
ys = (y[:m] * 1000.0).astype(np.float32)
t = time.perf_counter(); r = plan.solve(ys, 250); dt = time.perf_counter() - t
print(f"nnls host mode, float32 storage: {m/dt/1e6:.2f} M voxels/s ({dt*1e3:.0f} ms)", flush=True)
