#!/usr/bin/env python3
"""A/B probe for curve-fit kernel variants (PNX_VARIANT builds): C3 volume device resident, voxels/s, plus parity of the
first 8192 voxels with the oracle.   python profiles/curvefit_probe.py [lib.so ...]"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, numpy as np, torch
sys.path.insert(0, %r)
from pyneapple_amd import api, synth
from oracle import pnx_oracle as O
n = 1 << 22
dev = torch.device("cuda", 0)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
b, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01)
opts = api.make_opts("tri_reduced", 32, max_nfev=250, ftol=1e-8, jac="fd")
popt = torch.empty((5, n), dtype=torch.float64, device=dev); pcov = torch.empty((n, 5, 5), dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); nf = torch.empty(n, dtype=torch.int32, device=dev); cost = torch.empty(n, dtype=torch.float64, device=dev)
s = torch.cuda.current_stream().cuda_stream
run = lambda: api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, pcov, st, nf, cost, 0, s)
run(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
m = 8192
o = O.curvefit("tri_reduced", b, y[:m].cpu().numpy(), p0, lo, hi, n_threads=16)
rel = np.abs(popt[:, :m].cpu().numpy() - o["popt"]) / np.abs(o["popt"])
print(json.dumps({"voxels_per_s": n / min(ts), "ms": min(ts) * 1e3, "within_1e-4": float((rel.max(axis=0) <= 1e-4).mean()),
                  "median_rel": float(np.median(rel.max(axis=0))), "status_equal": float((st[:m].cpu().numpy() == o["status"]).mean()),
                  "nfev_equal": float((nf[:m].cpu().numpy() == o["nfev"]).mean()), "mean_nfev": float(nf.double().mean())}))
'''
for lib in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    if lib:
        env["PNX_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD % HERE], env=env, capture_output=True, text=True)
    print(os.path.basename(lib) or "product", r.stdout.strip() or r.stderr[-1500:], flush=True)
