from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The library reads its developer switches (A/B kernel selection, chunk sizes, the rejection hook of the NNLS block kernel) only
# in a process started with this variable (include/pnx.h, "Environment"); the test-suite is such a process, and so are the
# children it spawns.  tests/test_gpu_nnls.py::test_developer_switches_are_ignored_without_the_gate covers the other side.
os.environ["PNX_ENABLE_TEST_HOOKS"] = "1"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


# fixture name -> kernel model key
CURVEFIT_FIXTURES = {
    "g1_mono_b16": "mono", "g1_mono_b8": "mono",
    "g2_bi_reduced": "bi_reduced", "g2_bi_s0": "bi_s0", "g2_bi_full": "bi_full",
    "g3_tri_reduced": "tri_reduced", "g3_tri_reduced_maxiter4": "tri_reduced",
    "g3_tri_s0": "tri_s0", "g3_tri_full": "tri_full",
    "g5_bi_pervoxel": "bi_reduced", "g5_tri_pervoxel": "tri_reduced",
}
# seventh batch (oracle/gen_golden.py main_g12): curve_fit(sigma=..., absolute_sigma=...) as the reference forwards them
# (solvers/curvefit.py:33, 295-306): kernel model, fixed parameter positions (analytic Jacobian) or ()
G12_FIXTURES = {
    "g12_bi_sigma_rel": ("bi_reduced", ()), "g12_bi_sigma_abs": ("bi_reduced", ()),
    "g12_tri_sigma_abs": ("tri_reduced", ()), "g12_tri_sigma_rel": ("tri_reduced", ()),
    "g12_tri_sigma_scalar": ("tri_reduced", ()), "g12_tri_abs_nosigma": ("tri_reduced", ()),
    "g12_bi_s0_fixed_D1_sigma_abs": ("bi_s0", (1,)),
}


def g12_case(name):
    """(golden data, model, free positions, keyword arguments for oracle.curvefit / api.curvefit of a g12 fixture)."""
    model, fixed_idx = G12_FIXTURES[name]
    d = load_golden(name)
    n_all = len(d["p0_vals"])
    free = [i for i in range(n_all) if i not in fixed_idx]
    kw = dict(sigma=d["fit_sigma"] if "fit_sigma" in d.files else None, absolute_sigma=bool(d["absolute_sigma"]))
    if fixed_idx:
        names = [str(x) for x in d["all_param_names"]]
        kw.update(fixed_idx=list(fixed_idx), fixed_vals=np.stack([d["fixed_" + names[i]] for i in fixed_idx]), jac="analytic")
    else:
        kw.update(jac="fd")
    return d, model, free, kw


def _g12_weighted_cost(model, d, popt, free, kw):
    """0.5 sum ((model(popt) - y) / sigma)^2 per voxel, popt (n_vox, n_free) -- numpy, for the valley check of check_g12."""
    b, y = d["bvalues"], d["y"]
    n_all = len(d["p0_vals"])
    P = np.empty((len(y), n_all))
    P[:, free] = popt
    for k, i in enumerate(kw.get("fixed_idx", ())):
        P[:, i] = kw["fixed_vals"][k]
    e = lambda D: np.exp(-b[None, :] * D[:, None])
    if model == "bi_reduced":
        f = P[:, 0:1] * e(P[:, 1]) + (1 - P[:, 0:1]) * e(P[:, 2])
    elif model == "bi_s0":
        f = P[:, 3:4] * (P[:, 0:1] * e(P[:, 1]) + (1 - P[:, 0:1]) * e(P[:, 2]))
    elif model == "tri_reduced":
        f = P[:, 0:1] * e(P[:, 1]) + P[:, 2:3] * e(P[:, 3]) + (1 - P[:, 0:1] - P[:, 2:3]) * e(P[:, 4])
    else:
        raise ValueError(model)
    w = 1.0 if kw.get("sigma") is None else 1.0 / np.broadcast_to(np.asarray(kw["sigma"], float).reshape(-1), (len(b),))
    return 0.5 * (((f - y) * w) ** 2).sum(axis=1)


def check_g12(r, d, model, free, kw):
    """Parity with the reference's weighted fits.  Estimates: at least 97 % of the voxels within rtol 1e-4, and every voxel beyond
    it sits in the same valley -- its weighted cost equals the cost at the reference's estimate to 1e-7 (relative).  (With a
    vector sigma the trust-region walk along a flat valley is sensitive to the last bits of the SVD: SciPy itself, called
    directly, takes one evaluation more or less than this restatement on 2-4 % of the noisy voxels -- 0 % unweighted or with a
    scalar sigma -- and stops up to 3e-3 away at the same cost.)  Covariance where it means something: noisy data, well
    conditioned correlation matrix, no variance (in units of its parameter's squared bound width) 1e8 beyond the others -- a
    compartment whose D sits on its upper bound leaves a Jacobian column at the rounding level of the residuals, SciPy's own
    2-point quotient is quantised noise there (variance 1e12) and differs by factors of two between any two implementations."""
    ok = r["status"] > 0
    assert (ok == d["success"]).all()
    pe = rel_err(r["popt"].T, d["popt"]).max(axis=1)
    assert (pe <= 1e-4).mean() >= 0.97 and pe.max() < 1e-2
    far = pe > 1e-4
    if far.any():
        c_us = _g12_weighted_cost(model, d, r["popt"].T, free, kw)[far]
        c_ref = _g12_weighted_cost(model, d, d["popt"], free, kw)[far]
        assert (np.abs(c_us - c_ref) <= 1e-7 * c_ref).all()
    sel = ok & (d["sigma"] > 0) & ~far

    def corr_cond(c):  # condition of the correlation matrix: scale free (S0 ~ 1e3 beside D ~ 1e-3 in one covariance)
        if not np.isfinite(c).all() or (np.diag(c) <= 0).any():
            return np.inf
        s = np.sqrt(np.diag(c))
        return np.linalg.cond(c / s[:, None] / s[None, :])

    width = (d["hi_vals"] - d["lo_vals"])[free]

    def spread(c):  # variances in units of the squared bound widths: one of them 1e8 beyond the others = a dead Jacobian column
        v = np.abs(np.diag(c)) / width ** 2
        return v.max() / max(v.min(), 1e-300)

    cond = np.array([corr_cond(c) for c in d["pcov"][sel]])
    informative = np.array([spread(c) < 1e8 for c in d["pcov"][sel]])
    good = (cond < 1e10) & informative
    assert good.mean() > 0.5
    e = pcov_norm_err(r["pcov"][sel][good], d["pcov"][sel][good])
    assert np.median(e) < 1e-5 and (e < 1e-2).mean() > 0.97


# second batch (oracle/gen_golden.py main_g7): (kernel model, extra solver arguments, index of the amplitude that is
# degenerate with T1 or None).  With a free T1 only amplitude * relaxation factor is identifiable (a flat valley:
# SciPy's own answer along it depends on rounding), so parity is asserted on that product, on the other parameters
# and on the cost.  With infinite bounds a fast compartment's D can run away (exp(-b D) = 0 for every b > 0): those
# voxels are compared on the cost.
G7_FIXTURES = {
    "g7_mono_t1_free": ("mono", dict(t1_mode=1, tr=3000.0), 0),
    "g7_bi_s0_steam_free": ("bi_s0", dict(t1_mode=2, tr=2500.0, tm=30.0), 3),
    "g7_mono_inf_bounds": ("mono", {}, None),
    "g7_bi_half_inf_bounds": ("bi_reduced", {}, None),
}


def check_g7(r, d, kw, amp):
    """Shared assertions of the g7 fixtures for the oracle and the HIP path (r: result dict, popt (n, n_vox))."""
    assert ((r["status"] > 0) == d["success"]).all()
    got, ref = r["popt"].T, d["popt"]
    if amp is not None:
        tr, tm, steam = kw["tr"], kw.get("tm", 0.0), kw["t1_mode"] == 2
        fac = lambda T1: (1 - np.exp(-tr / T1)) * (np.exp(-tm / T1) if steam else 1.0)
        prod_g, prod_r = got[:, amp] * fac(got[:, -1]), ref[:, amp] * fac(ref[:, -1])
        assert np.abs(prod_g / prod_r - 1).max() < 1e-5
        others = [k for k in range(ref.shape[1]) if k not in (amp, ref.shape[1] - 1)]
        assert rel_err(got[:, others], ref[:, others]).max() <= 1e-4
    else:
        # a compartment whose D ran away under an infinite upper bound (exp(-b_1 D) < 1e-3: it only contributes at
        # b = 0) leaves that D undetermined -- SciPy's own answer there depends on rounding; cost only for those
        b1 = np.sort(d["bvalues"])[1]
        dcols = [k for k, nm in enumerate(d["all_param_names"]) if str(nm).startswith("D")]
        well = (np.exp(-b1 * ref[:, dcols]) >= 1e-3).all(axis=1)
        assert well.mean() > 0.9
        assert rel_err(got[well], ref[well]).max() <= 1e-4
    # every voxel ends at an equally good minimum: cost of the reference's popt, re-evaluated with the stand-in models
    from pyneapple_amd import models as M

    name = str(d["all_param_names"][-1])
    n_all = len(d["all_param_names"])
    t1kw = {}
    if amp is not None:
        t1kw = dict(fit_t1=True, repetition_time=kw["tr"])
        if kw["t1_mode"] == 2:
            t1kw.update(fit_t1_steam=True, mixing_time=kw["tm"])
    n_base = n_all - (1 if amp is not None else 0)
    model = {2: M.MonoExpModel(**t1kw), 3: M.BiExpModel(**t1kw), 4: M.BiExpModel(fit_s0=True, **t1kw)}[n_base]
    cost_ref = np.array([0.5 * np.sum((model.forward(d["bvalues"], *ref[i]) - d["y"][i]) ** 2) for i in range(len(ref))])
    scale = 0.5 * np.sum(d["y"] ** 2, axis=1)  # noise-free voxels end at cost ~ 1e-26: compare against the signal energy
    assert (np.abs(r["cost"] - cost_ref) <= 1e-6 * cost_ref + 1e-12 * scale).all()


NNLS_FIXTURES = ["g4_nnls_250_r2", "g4_nnls_250_r1", "g4_nnls_250_r3", "g4_nnls_50_r2", "g4_nnls_50_r0",
                 "g4_nnls_250_r2_maxiter20", "g9_nnls_250_r0",  # g9_nnls_250_r0: the reference's default reg_order=0
                 # round 4: more than 256 bins (the wide instantiations; 512 bins: two voxels run into max_iter = 250)
                 "g11_nnls_300_r2", "g11_nnls_512_r2", "g11_nnls_350_r1", "g11_nnls_300_r3", "g11_nnls_400_r0",
                 "g11_nnls_300_r2_maxiter20"]


def many_fixed_cases():
    """(fixture, kernel model, free idx, fixed idx, fixed maps (n_fixed, n_vox), extra kwargs): more than two per-pixel
    fixed parameters -- SegmentedFitter's second step with every diffusivity carried over (fitters/segmented.py:198-225)."""
    d = load_golden("g9_tri_s0_fixed_D1_D2_D3")
    yield d, "tri_s0", [0, 2, 5], [1, 3, 4], np.stack([d["fixed_D1"], d["fixed_D2"], d["fixed_D3"]]), {}
    d = load_golden("g9_tri_full_fixed_D1_D2_D3")
    yield d, "tri_full", [0, 2, 4], [1, 3, 5], np.stack([d["fixed_D1"], d["fixed_D2"], d["fixed_D3"]]), {}
    d = load_golden("g9_tri_fixed_4_of_5")
    yield d, "tri_reduced", [0], [1, 2, 3, 4], np.stack([d["fixed_D1"], d["fixed_f2"], d["fixed_D2"], d["fixed_D3"]]), {}


def golden_p0_bounds(d):
    if "p0_arr" in d.files:
        return d["p0_arr"], d["lo_arr"], d["hi_arr"]
    return d["p0_vals"], d["lo_vals"], d["hi_vals"]


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def pcov_norm_err(pc, pr):
    """max |dpcov_ij| / sqrt(pcov_ii pcov_jj) per voxel (scale-free)."""
    dg = np.sqrt(np.abs(np.einsum("vii->vi", pr)))
    return (np.abs(pc - pr) / (dg[:, :, None] * dg[:, None, :])).max(axis=(1, 2))


@pytest.fixture(scope="session")
def oracle():
    from oracle import pnx_oracle

    pnx_oracle.lib()
    return pnx_oracle


@pytest.fixture(scope="session")
def gpu():
    """The HIP backend; the test is an error (not a skip) if a `gpu`-marked test runs without a device."""
    from pyneapple_amd import _lib, api

    _lib.load()
    assert _lib.device_count() >= 1, "gpu-marked test running without a visible HIP device"
    return api
