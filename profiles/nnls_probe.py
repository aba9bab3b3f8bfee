#!/usr/bin/env python3
"""A/B probe for NNLS kernel variants: for every library given (PNX_LIB builds from `PNX_VARIANT=x python -m
pyneapple_amd._build`), in a fresh process, the C4 workload on 2^20 voxels device resident: voxels/s, and parity
with the oracle on the first 4096 voxels (status, iteration counts, coefficients).
    python profiles/nnls_probe.py [lib.so ...]          (no argument: the product library)"""
import json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, numpy as np, torch
sys.path.insert(0, %r)
from pyneapple_amd import api, synth
from oracle import pnx_oracle as O
n = int(%d)
dev = torch.device("cuda", 0)
nb_ = int(%d)
bins, basis, reg = synth.nnls_matrices(nb_)
plan = api.NnlsPlan(basis, reg, 0)
_, y = synth.make_torch_rows("tri_reduced", 0, n, nb_, dev, sigma=0.01, scale=1000.0)
coeff = torch.empty((n, 250), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize()
ts = []
for _ in range(2):
    t = time.perf_counter(); plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
m = 4096
o = O.nnls(basis, reg, y[:m].cpu().numpy(), 250, n_threads=16)
c = coeff[:m].cpu().numpy(); cr = o["coefficients"]
err = (np.abs(c - cr).max(axis=1) / (np.abs(cr).max(axis=1) + 1e-300))
print(json.dumps({"voxels_per_s": n / min(ts), "ms": min(ts) * 1e3, "status_equal": float((st[:m].cpu().numpy() == o["status"]).mean()),
                  "iters_equal": float((it[:m].cpu().numpy() == o["iters"]).mean()), "coef_err_max": float(err.max()),
                  "rnorm_rel_max": float(np.abs(rn[:m].cpu().numpy() / o["residual"] - 1).max()), "mean_iters": float(it.double().mean())}))
'''
libs = sys.argv[1:] or [""]
n = int(os.environ.get("PNX_PROBE_VOXELS", 1 << 20))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["PNX_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD % (HERE, n, int(os.environ.get("PNX_PROBE_NB", 32)))], env=env, capture_output=True, text=True)
    print(os.path.basename(lib) or "product", r.stdout.strip() or r.stderr[-1500:], flush=True)
