#!/usr/bin/env python3
"""does starting the long fits first pay?  C3 (triexp, 4 194 304 voxels, FD Jacobian, pcov): one pass
in ascending voxel order, then refits whose queue order comes from the first pass's nfev map (longest first) -- the predictor a
refit / SegmentedFitter step 2 has for free -- and from a degraded predictor (nfev of every 8th voxel, nearest neighbour: what
an IDEAL level gets from the previous, coarser level).  Results must be bit-identical to the unordered pass."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import api, synth
dev = torch.device("cuda", 0)
model, n_b, shape = synth.WORKLOADS["triexp"]
n = int(os.environ.get("PNX_PROBE_VOXELS", int(np.prod(shape))))
names, p0, lo, hi = synth.shared_arrays(model)
b, y = synth.make_torch_rows(model, 0, n, n_b, dev, sigma=0.01)
opts = api.make_opts(model, n_b, max_nfev=250, ftol=1e-8, jac="fd")
k = len(names)
mk = lambda: (torch.empty((k, n), dtype=torch.float64, device=dev), torch.empty((n, k, k), dtype=torch.float64, device=dev),
              torch.empty(n, dtype=torch.int8, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float64, device=dev))
ref = mk()
s = torch.cuda.current_stream().cuda_stream
def run(out, order=None, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        api.curvefit_device(opts, n, b, y, p0, lo, hi, None, *out, 0, s, order=order)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return ts
t_plain = run(ref)
nfev = ref[3]
rows = [{"order": "ascending voxel index", "ms": t_plain}]
def check(out):
    return all(bool((a == b_).all().item()) or bool(((a == b_) | (a.isnan() & b_.isnan())).all().item()) for a, b_ in zip(out, ref))
# exact predictor: this volume's own nfev, longest first (stable)
order = torch.argsort(nfev, descending=True, stable=True).to(torch.int32)
out = mk(); t = run(out, order); rows.append({"order": "own nfev, longest first", "ms": t, "identical": check(out)})
# only the long tail first: voxels with nfev >= 40 in front, the rest in index order
long_first = torch.cat([torch.nonzero(nfev >= 40).flatten(), torch.nonzero(nfev < 40).flatten()]).to(torch.int32)
out = mk(); t = run(out, long_first); rows.append({"order": "voxels with nfev >= 40 first, rest ascending", "ms": t, "identical": check(out), "n_long": int((nfev >= 40).sum())})
# coarse predictor: nfev known for every 8th voxel only
coarse = nfev[::8].repeat_interleave(8)[:n]
order_c = torch.argsort(coarse, descending=True, stable=True).to(torch.int32)
out = mk(); t = run(out, order_c); rows.append({"order": "nfev of every 8th voxel (nearest neighbour), longest first", "ms": t, "identical": check(out)})
# sanity: a random permutation (costs the coalescing of nothing: each lane fetches its own row anyway)
perm = torch.randperm(n, device=dev).to(torch.int32)
out = mk(); t = run(out, perm); rows.append({"order": "random permutation", "ms": t, "identical": check(out)})
for r in rows:
    r["ms_min"] = min(r["ms"]); r["M_voxels_per_s"] = n / min(r["ms"]) / 1e3
    print(json.dumps(r), flush=True)
print("nfev: mean %.2f, max %d, >= 40: %d, >= 100: %d" % (float(nfev.double().mean()), int(nfev.max()), int((nfev >= 40).sum()), int((nfev >= 100).sum())))
