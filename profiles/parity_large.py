#!/usr/bin/env python3
"""Large-sample parity on the GPU box: HIP path vs the oracle (host threads) on voxels of the C3 workload (default 1 048 576)
and of the C4 workload (default 131 072).  For the curve fit the voxels beyond rtol 1e-4 are characterised by their cost
difference (a flat valley gives different parameters at the same cost).

    python profiles/parity_large.py [--c3 N] [--c4 M] [--c4-first K] [--json profiles/r03_parity_large.json]

`c3()` / `c4()` are what tests/test_gpu_parity_large.py runs on a bounded sample inside the GPU suite."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def _threads():
    try:
        return min(len(os.sched_getaffinity(0)), 16)
    except AttributeError:
        return min(os.cpu_count() or 1, 16)


def c3(n=1 << 20, verbose=True):
    import torch
    from oracle import pnx_oracle as O
    from pyneapple_amd import api, synth

    dev = torch.device("cuda", 0)
    names, p0, lo, hi = synth.shared_arrays("tri_reduced")
    b, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01)
    opts = api.make_opts("tri_reduced", 32, max_nfev=250, ftol=1e-8, jac="fd")
    popt = torch.empty((5, n), dtype=torch.float64, device=dev); pcov = torch.empty((n, 5, 5), dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int8, device=dev); nf = torch.empty(n, dtype=torch.int32, device=dev); cost = torch.empty(n, dtype=torch.float64, device=dev)
    api.curvefit_device(opts, n, b, y, p0, lo, hi, None, popt, pcov, st, nf, cost, 0, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    t = time.perf_counter()
    o = O.curvefit("tri_reduced", b, y.cpu().numpy(), p0, lo, hi, n_threads=_threads())
    if verbose:
        print(f"oracle: {n / (time.perf_counter() - t):.0f} voxels/s", flush=True)
    g = popt.cpu().numpy(); rel = (np.abs(g - o["popt"]) / np.abs(o["popt"])).max(axis=0)
    gc, oc = cost.cpu().numpy(), o["cost"]
    bad = rel > 1e-4
    res = {"workload": "C3 triexp, 32 b-values, 1 % noise, FD Jacobian, seed-fixed synthetic rows [0, n)", "n": n,
           "within_1e-4": float((~bad).mean()), "within_1e-6": float((rel <= 1e-6).mean()), "median_rel": float(np.median(rel)),
           "status_equal": float((st.cpu().numpy() == o["status"]).mean()), "success_equal": float(((st.cpu().numpy() > 0) == (o["status"] > 0)).mean()),
           "nfev_equal": float((nf.cpu().numpy() == o["nfev"]).mean()), "n_beyond_1e-4": int(bad.sum()),
           "cost_rel_diff_of_those_median": float(np.median(np.abs(gc[bad] - oc[bad]) / oc[bad])) if bad.any() else 0.0,
           "cost_rel_diff_of_those_max": float(np.max(np.abs(gc[bad] - oc[bad]) / oc[bad])) if bad.any() else 0.0,
           "gpu_cost_lower_frac_of_those": float((gc[bad] <= oc[bad]).mean()) if bad.any() else 0.0,
           "cost_rel_diff_all_max": float(np.max(np.abs(gc - oc) / oc))}
    pc = pcov.cpu().numpy(); ok = ~bad & np.isfinite(o["pcov"]).all(axis=(1, 2))
    res["pcov_rel_max_where_params_agree_median"] = float(np.median((np.abs(pc[ok] - o["pcov"][ok]).max(axis=(1, 2)) / np.abs(o["pcov"][ok]).max(axis=(1, 2)))))
    return res


def c4(m=1 << 17, verbose=True, first=0):
    import torch
    from oracle import pnx_oracle as O
    from pyneapple_amd import api, synth

    dev = torch.device("cuda", 0)
    bins, basis, reg = synth.nnls_matrices(32)
    plan = api.NnlsPlan(basis, reg, 0)
    _, y = synth.make_torch_rows("tri_reduced", first, first + m, 32, dev, sigma=0.01, scale=1000.0)
    coeff = torch.empty((m, 250), dtype=torch.float64, device=dev); rn = torch.empty(m, dtype=torch.float64, device=dev)
    s8 = torch.empty(m, dtype=torch.int8, device=dev); it = torch.empty(m, dtype=torch.int32, device=dev)
    plan.solve_device(m, y, 250, coeff, rn, s8, it, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    t = time.perf_counter()
    o = O.nnls(basis, reg, y.cpu().numpy(), 250, n_threads=_threads())
    if verbose:
        print(f"oracle: {m / (time.perf_counter() - t):.0f} voxels/s", flush=True)
    c = coeff.cpu().numpy(); cr = o["coefficients"]
    err = np.abs(c - cr).max(axis=1) / np.maximum(np.abs(cr).max(axis=1), 1e-300)
    itg = it.cpu().numpy()
    differ = np.flatnonzero((itg != o["iters"]) | ~((c > 0) == (cr > 0)).all(axis=1))
    odd = [{"voxel": int(v), "iters_gpu": int(itg[v]), "iters_oracle": int(o["iters"][v]), "support_gpu": int((c[v] > 0).sum()),
            "support_oracle": int((cr[v] > 0).sum()), "bins_only_gpu": np.flatnonzero((c[v] > 0) & ~(cr[v] > 0)).tolist(),
            "bins_only_oracle": np.flatnonzero(~(c[v] > 0) & (cr[v] > 0)).tolist(), "coef_err": float(err[v]),
            "rnorm_gpu": float(rn[v]), "rnorm_oracle": float(o["residual"][v]),
            "largest_coefficient_on_a_differing_bin_rel_peak": float(max([abs(c[v, j]) for j in np.flatnonzero((c[v] > 0) != (cr[v] > 0))] + [abs(cr[v, j]) for j in np.flatnonzero((c[v] > 0) != (cr[v] > 0))] + [0.0]) / np.abs(cr[v]).max())}
           for v in differ[:16]]
    return {"voxels_with_a_different_path": odd,"workload": "C4 NNLS 250 bins, reg_order 2, mu 0.02, 32 b-values, seed-fixed synthetic rows [first, first + n)", "first": first, "n": m,
            "status_equal": float((s8.cpu().numpy() == o["status"]).mean()), "iters_equal": float((it.cpu().numpy() == o["iters"]).mean()),
            "support_equal": float((((c > 0) == (cr > 0)).all(axis=1)).mean()), "coef_err_max": float(err.max()),
            "coef_err_median": float(np.median(err)), "max_passive_set": int((c > 0).sum(axis=1).max()),
            "rnorm_rel_max": float(np.max(np.abs(rn.cpu().numpy() - o["residual"]) / o["residual"]))}


def main():
    from pyneapple_amd import _build, _lib

    _lib.load()
    a = sys.argv[1:]
    n3 = int(a[a.index("--c3") + 1]) if "--c3" in a else 1 << 20
    n4 = int(a[a.index("--c4") + 1]) if "--c4" in a else 1 << 17
    out = {"source_ids": _build.source_ids(), "oracle_threads": _threads()}
    if n3:
        out["curvefit"] = c3(n3)
        print("curvefit", json.dumps(out["curvefit"]), flush=True)
    if n4:
        out["nnls"] = c4(n4, first=int(a[a.index("--c4-first") + 1]) if "--c4-first" in a else 0)
        print("nnls", json.dumps(out["nnls"]), flush=True)
    if "--json" in a:
        with open(a[a.index("--json") + 1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
