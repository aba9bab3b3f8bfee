#!/usr/bin/env python3
"""How the NNLS plans behave away from the reference's regularisation strength: the block kernel hands a voxel over to the
Gram-form kernel when its passive set wants a 129th column, which costs that voxel its first ~130 outer iterations twice.  For
each (reg_order, mu): voxels/s of the block-kernel plan and of the Gram-form kernel alone (PNX_NNLS_NO_BLK=1), 2^18 voxels of
the C4 signal, device resident, each in a fresh process.
    python profiles/nnls_mu_probe.py [order,mu ...]"""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json, numpy as np, torch
sys.path.insert(0, %r)
from pyneapple_amd import api, synth
order, mu, n = int(%d), float(%r), int(%d)
dev = torch.device("cuda", 0)
cfg = dict(synth.NNLS_CFG) if hasattr(synth, "NNLS_CFG") else {"d_range": (0.0008, 0.5), "n_bins": 250, "reg_order": 2, "mu": 0.02, "max_iter": 250}
cfg.update(reg_order=order, mu=mu)
bins, basis, reg = synth.nnls_matrices(32, cfg)
plan = api.NnlsPlan(basis, reg, 0)
import os
_, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=float(os.environ.get("PNX_PROBE_SIGMA", "0.01")), scale=1000.0)
coeff = torch.empty((n, 250), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize()
ts = []
for _ in range(2):
    t = time.perf_counter(); plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
supp = (coeff > 0).sum(dim=1)
print(json.dumps({"voxels_per_s": n / min(ts), "ms": min(ts) * 1e3, "mean_iters": float(it.double().mean()), "support_mean": float(supp.double().mean()),
                  "support_gt_120": float((supp > 120).double().mean()), "max_iter_reached": float((st == 0).double().mean()),
                  "checksum": float(coeff.sum())}))
'''
cases = [tuple(a.split(",")) for a in sys.argv[1:]] or [("2", "0.02"), ("2", "0.1"), ("2", "0.5"), ("1", "0.02"), ("1", "0.1"), ("1", "0.5"), ("3", "0.02"), ("3", "0.1")]
n = int(os.environ.get("PNX_PROBE_VOXELS", 1 << 18))
for order, mu in cases:
    for no_blk in ("", "1"):
        env = dict(os.environ)
        if no_blk:
            env["PNX_NNLS_NO_BLK"] = "1"
        r = subprocess.run([sys.executable, "-c", CHILD % (HERE, int(order), float(mu), n)], env=env, capture_output=True, text=True)
        pilot = [l for l in r.stderr.splitlines() if "pnx nnls pilot" in l]
        print(f"order={order} mu={mu} {'gram-form only' if no_blk else 'block plan   '}", r.stdout.strip() or r.stderr[-1200:], pilot[0] if pilot else "", flush=True)
