#!/usr/bin/env python3
"""NNLS parity against the oracle over the conditioning of the problem: measurements 32 / 16 / 8, regulariser order 1-3,
mu 0.0005 ... 0.02 (250 bins, 2000 noisy triexp voxels each).  Not collected by pytest: run it by hand on a GPU box
(it lives under tests/ because only tests may use the oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import pnx_oracle as oracle
from pyneapple_amd import api, synth
_, y, _ = synth.make_numpy("tri_reduced", 2000, 32, sigma=0.01, seed=11, scale=1000.0)
for n_meas in (32, 16, 8):
    b = np.linspace(0, 1200, n_meas)
    yy = y[:, :: 32 // n_meas][:, :n_meas] if n_meas < 32 else y
    # regenerate signals on this b grid
    _, yy, _ = synth.make_numpy("tri_reduced", 2000, n_meas, sigma=0.01, seed=11, scale=1000.0)
    for order in (1, 2, 3):
        for mu in (0.0005, 0.001, 0.002, 0.005, 0.01, 0.02):
            cfg = dict(synth.NNLS_CFG, reg_order=order, mu=mu)
            _, basis, reg = synth.nnls_matrices(n_meas, cfg)
            r = api.nnls(basis, reg, yy, 250); o = oracle.nnls(basis, reg, yy, 250, n_threads=16)
            ok = (r["status"] == 1) & (o["status"] == 1)
            peak = np.abs(o["coefficients"]).max(axis=1) + 1e-300
            ce = np.abs(r["coefficients"] - o["coefficients"]).max(axis=1) / peak
            print(f"n_meas {n_meas} order {order} mu {mu}: gpu fail {np.mean(r['status']!=1):.4f} oracle fail {np.mean(o['status']!=1):.4f} status differ {np.mean(r['status']!=o['status']):.4f} "
                  f"coeff err max {ce[ok].max() if ok.any() else float('nan'):.2e} frac>1e-6 {np.mean(ce[ok]>1e-6) if ok.any() else 0:.4f} iters equal {np.mean(r['iters'][ok]==o['iters'][ok]):.4f} mean iters {o['iters'].mean():.0f}", flush=True)
