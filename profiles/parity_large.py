#!/usr/bin/env python3
"""Large-sample parity on the GPU box: HIP path vs the oracle (16 host threads) on 1 048 576 voxels of the C3 workload and
131 072 voxels of the C4 workload.  For the curve fit the voxels beyond rtol 1e-4 are characterised by their cost difference
(a flat valley gives different parameters at the same cost)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
from oracle import pnx_oracle as O
_lib.load()
dev = torch.device("cuda", 0)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
n = 1 << 20 if "--nnls-only" not in sys.argv else 4096
b, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01)
opts = api.make_opts("tri_reduced", 32, max_nfev=250, ftol=1e-8, jac="fd")
popt = torch.empty((5, n), dtype=torch.float64, device=dev); pcov = torch.empty((n, 5, 5), dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); nf = torch.empty(n, dtype=torch.int32, device=dev); cost = torch.empty(n, dtype=torch.float64, device=dev)
api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, pcov, st, nf, cost, 0, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
t = time.perf_counter()
o = O.curvefit("tri_reduced", b, y.cpu().numpy(), p0, lo, hi, n_threads=16)
print(f"oracle: {n / (time.perf_counter() - t):.0f} voxels/s", flush=True)
g = popt.cpu().numpy(); rel = (np.abs(g - o["popt"]) / np.abs(o["popt"])).max(axis=0)
gc, oc = cost.cpu().numpy(), o["cost"]
bad = rel > 1e-4
res = {"n": n, "within_1e-4": float((~bad).mean()), "within_1e-6": float((rel <= 1e-6).mean()), "median_rel": float(np.median(rel)),
       "status_equal": float((st.cpu().numpy() == o["status"]).mean()), "success_equal": float(((st.cpu().numpy() > 0) == (o["status"] > 0)).mean()),
       "nfev_equal": float((nf.cpu().numpy() == o["nfev"]).mean()), "n_beyond_1e-4": int(bad.sum()),
       "cost_rel_diff_of_those_median": float(np.median(np.abs(gc[bad] - oc[bad]) / oc[bad])) if bad.any() else 0.0,
       "cost_rel_diff_of_those_max": float(np.max(np.abs(gc[bad] - oc[bad]) / oc[bad])) if bad.any() else 0.0,
       "gpu_cost_lower_frac_of_those": float((gc[bad] <= oc[bad]).mean()) if bad.any() else 0.0,
       "cost_rel_diff_all_max": float(np.max(np.abs(gc - oc) / oc))}
pc = pcov.cpu().numpy(); ok = ~bad & np.isfinite(o["pcov"]).all(axis=(1, 2))
res["pcov_rel_max_where_params_agree_median"] = float(np.median((np.abs(pc[ok] - o["pcov"][ok]).max(axis=(1, 2)) / np.abs(o["pcov"][ok]).max(axis=(1, 2)))))
print("curvefit", json.dumps(res), flush=True)
del y, popt, pcov
m = 1 << 17
bins, basis, reg = synth.nnls_matrices(32)
plan = api.NnlsPlan(basis, reg, 0)
_, y = synth.make_torch_rows("tri_reduced", 0, m, 32, dev, sigma=0.01, scale=1000.0)
coeff = torch.empty((m, 250), dtype=torch.float64, device=dev); rn = torch.empty(m, dtype=torch.float64, device=dev)
s8 = torch.empty(m, dtype=torch.int8, device=dev); it = torch.empty(m, dtype=torch.int32, device=dev)
plan.solve_device(m, y, 250, coeff, rn, s8, it, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
t = time.perf_counter()
o = O.nnls(basis, reg, y.cpu().numpy(), 250, n_threads=16)
print(f"oracle: {m / (time.perf_counter() - t):.0f} voxels/s", flush=True)
c = coeff.cpu().numpy(); cr = o["coefficients"]
err = np.abs(c - cr).max(axis=1) / np.maximum(np.abs(cr).max(axis=1), 1e-300)
print("nnls", json.dumps({"n": m, "status_equal": float((s8.cpu().numpy() == o["status"]).mean()), "iters_equal": float((it.cpu().numpy() == o["iters"]).mean()),
                          "support_equal": float((((c > 0) == (cr > 0)).all(axis=1)).mean()), "coef_err_max": float(err.max()), "coef_err_median": float(np.median(err)),
                          "rnorm_rel_max": float(np.max(np.abs(rn.cpu().numpy() - o["residual"]) / o["residual"]))}), flush=True)
