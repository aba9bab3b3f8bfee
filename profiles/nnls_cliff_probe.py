#!/usr/bin/env python3
"""Device-resident NNLS rates either side of the kernels' applicability limits (missing #4): the block kernel
takes the reference's banded regularisers with <= 32 b-values, the general (Gram-form) kernel everything else regularised, the
QR-form kernels the unregularised default (<= 64 b-values in LDS, 65..128 in a global slab).  250 bins, 2^18 voxels each.
PNX_CLIFF_BINS=300 (or 512): the same cases with that many bins -- beyond 256 the wide (eight bins per lane) instantiations."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import api, synth
dev = torch.device("cuda", 0)
n = int(os.environ.get("PNX_PROBE_VOXELS", 1 << 18))
rows = []
NB = int(os.environ.get("PNX_CLIFF_BINS", 250))
DEFAULT = (("block kernel", 32, 2), ("block kernel", 16, 2), ("general kernel", 33, 2), ("general kernel", 48, 2), ("general kernel", 64, 2),
           ("general kernel", 128, 2), ("QR kernel (LDS)", 32, 0), ("QR kernel", 33, 0), ("QR kernel", 64, 0), ("QR kernel (slab)", 65, 0),
           ("QR kernel (slab)", 96, 0), ("QR kernel (slab)", 128, 0))
CASES = os.environ.get("PNX_CLIFF_CASES")  # e.g. "33,48,64": unregularised plans with these numbers of b-values only
SPEC = os.environ.get("PNX_CLIFF_SPEC")    # e.g. "32:2,48:2,32:0,96:0": n_b:reg_order pairs
if SPEC:
    DEFAULT = tuple((("QR" if o == "0" else "Gram") + " kernel", int(nb), int(o)) for nb, o in (c.split(":") for c in SPEC.split(",")))
for label, n_b, order in ([("QR kernel", int(c), 0) for c in CASES.split(",")] if CASES else DEFAULT):
    cfg = dict(synth.NNLS_CFG, reg_order=order, n_bins=NB)
    bins, basis, reg = synth.nnls_matrices(n_b, cfg)
    plan = api.NnlsPlan(basis, reg, 0)
    _, y = synth.make_torch_rows("tri_reduced", 0, n, n_b, dev, sigma=0.01, scale=1000.0)
    coeff = torch.empty((n, NB), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        t = time.perf_counter(); plan.solve_device(n, y, 250, coeff, rn, st, it, s); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    rows.append({"kernel": label, "n_bins": NB, "n_b": n_b, "reg_order": order, "voxels_per_s": n / min(ts), "ms": min(ts) * 1e3,
                 "mean_iters": float(it.double().mean()), "mean_support": float((coeff > 0).sum(dim=1).double().mean()), "converged": float((st == 1).double().mean())})
    print(json.dumps(rows[-1]), flush=True)
    plan.close()
