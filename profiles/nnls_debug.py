#!/usr/bin/env python3
"""Small diagnostic solve of the C4 workload against the oracle (first PNX_DEBUG_VOXELS voxels, default 64): everything the
runtime prints goes to the caller's stdout / stderr.  PNX_LIB selects a kernel-variant library."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import api, synth
from oracle import pnx_oracle as O
n = int(os.environ.get("PNX_DEBUG_VOXELS", 64))
dev = torch.device("cuda", 0)
bins, basis, reg = synth.nnls_matrices(32)
print("plan", flush=True)
plan = api.NnlsPlan(basis, reg, 0)
_, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01, scale=1000.0)
coeff = torch.empty((n, 250), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
print("solve", flush=True)
plan.solve_device(n, y, int(os.environ.get("PNX_DEBUG_MAXITER", 250)), coeff, rn, st, it, s); torch.cuda.synchronize()
print("done", flush=True)
o = O.nnls(basis, reg, y.cpu().numpy(), int(os.environ.get("PNX_DEBUG_MAXITER", 250)), n_threads=8)
c = coeff.cpu().numpy(); cr = o["coefficients"]
err = (np.abs(c - cr).max(axis=1) / (np.abs(cr).max(axis=1) + 1e-300))
print(json.dumps({"status_equal": float((st.cpu().numpy() == o["status"]).mean()), "iters_equal": float((it.cpu().numpy() == o["iters"]).mean()),
                  "coef_err_max": float(err.max()), "rnorm_rel_max": float(np.abs(rn.cpu().numpy() / o["residual"] - 1).max()),
                  "iters_gpu": it.cpu().numpy()[:16].tolist(), "iters_ref": o["iters"][:16].tolist()}))
