"""The CPU oracle (oracle/pnx_oracle_trf.c) is pinned here: against golden vectors produced by the
reference itself (oracle/gen_golden.py) and against SciPy called the way the reference calls it
(curvefit.py:295-306)."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import (CURVEFIT_FIXTURES, G7_FIXTURES, G12_FIXTURES, check_g7, check_g12, g12_case, golden_p0_bounds, load_golden,
                      many_fixed_cases, pcov_norm_err, rel_err)

RTOL = 1e-4  # BASELINE.json north_star: rtol=1e-4 (fp64) per parameter


@pytest.mark.parametrize("name", sorted(CURVEFIT_FIXTURES))
def test_oracle_matches_reference_golden(oracle, name):
    d = load_golden(name)
    p0, lo, hi = golden_p0_bounds(d)
    r = oracle.curvefit(CURVEFIT_FIXTURES[name], d["bvalues"], d["y"], p0, lo, hi, max_nfev=int(d["max_iter"]),
                        ftol=float(d["tol"]), jac="fd")
    ok = r["status"] > 0
    assert (ok == d["success"]).all()
    assert rel_err(r["popt"].T, d["popt"]).max() <= RTOL
    # failed voxels: params == p0, cov NaN (curvefit.py:308-317)
    if (~ok).any():
        assert np.isnan(r["pcov"][~ok]).all()
    # covariance: compared where the reference's own pcov is numerically meaningful (cond < 1e10, noisy data)
    sel = ok & (d["sigma"] > 0)
    if sel.any():
        cond = np.array([np.linalg.cond(c) if np.isfinite(c).all() else np.inf for c in d["pcov"][sel]])
        good = cond < 1e10
        if good.any():
            assert np.median(pcov_norm_err(r["pcov"][sel][good], d["pcov"][sel][good])) < 1e-5
            assert (pcov_norm_err(r["pcov"][sel][good], d["pcov"][sel][good]) < 1e-2).mean() > 0.97


@pytest.mark.parametrize("name", sorted(G7_FIXTURES))
def test_oracle_matches_reference_golden_g7(oracle, name):
    """Free T1 / STEAM factor (FD Jacobian) and infinite bounds; see conftest.G7_FIXTURES for what is compared."""
    model, kw, amp = G7_FIXTURES[name]
    d = load_golden(name)
    r = oracle.curvefit(model, d["bvalues"], d["y"], d["p0_vals"], d["lo_vals"], d["hi_vals"], max_nfev=int(d["max_iter"]),
                        ftol=float(d["tol"]), jac="fd", **kw)
    check_g7(r, d, kw, amp)


@pytest.mark.parametrize("name", sorted(G12_FIXTURES))
def test_oracle_matches_reference_golden_sigma(oracle, name):
    """curve_fit(sigma=..., absolute_sigma=...): 1-D sigma (vector or scalar) scales residual and Jacobian rows by 1 / sigma
    (scipy:_minpack_py.py:958-960, 545-562); absolute_sigma leaves the covariance unscaled (:1057-1063)."""
    d, model, free, kw = g12_case(name)
    r = oracle.curvefit(model, d["bvalues"], d["y"], d["p0_vals"][free], d["lo_vals"][free], d["hi_vals"][free],
                        max_nfev=int(d["max_iter"]), ftol=float(d["tol"]), **kw)
    check_g12(r, d, model, free, kw)


def test_oracle_sigma_changes_the_fit_and_the_covariance_scale(oracle):
    """Guards the fixtures' meaning: with sigma the estimates differ from the unweighted fit, and absolute_sigma only changes
    the covariance (by the reduced chi square), not the estimates."""
    d, model, free, kw = g12_case("g12_tri_sigma_abs")
    a = (model, d["bvalues"], d["y"], d["p0_vals"], d["lo_vals"], d["hi_vals"])
    plain = oracle.curvefit(*a)
    w_abs = oracle.curvefit(*a, sigma=kw["sigma"], absolute_sigma=True)
    w_rel = oracle.curvefit(*a, sigma=kw["sigma"], absolute_sigma=False)
    noisy = d["sigma"] > 0
    assert rel_err(w_abs["popt"], plain["popt"]).max(axis=0)[noisy].min() > 1e-6
    assert np.array_equal(w_abs["popt"], w_rel["popt"]) and np.array_equal(w_abs["cost"], w_rel["cost"])
    m, n = len(d["bvalues"]), 5
    s_sq = 2 * w_rel["cost"] / (m - n)
    assert np.allclose(w_rel["pcov"][noisy], w_abs["pcov"][noisy] * s_sq[noisy, None, None], rtol=1e-12)


def test_oracle_fixed_params_golden(oracle):
    d = load_golden("g6_bi_fixed_D1")
    r = oracle.curvefit("bi_reduced", d["bvalues"], d["y"], d["p0_vals"][[0, 2]], d["lo_vals"][[0, 2]],
                        d["hi_vals"][[0, 2]], fixed_idx=[1], fixed_vals=d["fixed_D1"][None, :], jac="analytic")
    assert (r["status"] > 0).all() and d["success"].all()
    assert rel_err(r["popt"].T, d["popt"]).max() <= 1e-9
    d = load_golden("g6_mono_t1_fixed")
    r = oracle.curvefit("mono", d["bvalues"], d["y"], d["p0_vals"][:2], d["lo_vals"][:2], d["hi_vals"][:2],
                        t1_mode=1, tr=3000.0, fixed_idx=[2], fixed_vals=d["fixed_T1"][None, :], jac="analytic")
    assert rel_err(r["popt"].T, d["popt"]).max() <= 1e-9


def _two_fixed_cases():
    d = load_golden("g8_tri_fixed_D2_D3")
    yield d, "tri_reduced", [0, 1, 2], [3, 4], np.stack([d["fixed_D2"], d["fixed_D3"]]), {}
    d = load_golden("g8_mono_t1_fixed_S0_T1")
    yield d, "mono", [1], [0, 2], np.stack([d["fixed_S0"], d["fixed_T1"]]), dict(t1_mode=1, tr=3000.0)


def test_oracle_two_fixed_params_golden(oracle):
    """Two per-pixel fixed maps (SegmentedFitter with two parameters carried over from step 1)."""
    for d, model, free, fixed_idx, fv, kw in _two_fixed_cases():
        r = oracle.curvefit(model, d["bvalues"], d["y"], d["p0_vals"][free], d["lo_vals"][free], d["hi_vals"][free],
                            fixed_idx=fixed_idx, fixed_vals=fv, jac="analytic", **kw)
        assert (r["status"] > 0).all() and d["success"].all()
        assert rel_err(r["popt"].T, d["popt"]).max() <= 1e-9


def test_oracle_three_and_four_fixed_params_golden(oracle):
    for d, model, free, fixed_idx, fv, kw in many_fixed_cases():
        r = oracle.curvefit(model, d["bvalues"], d["y"], d["p0_vals"][free], d["lo_vals"][free], d["hi_vals"][free],
                            fixed_idx=fixed_idx, fixed_vals=fv, jac="analytic", **kw)
        assert (r["status"] > 0).all() and d["success"].all()
        assert list(d["free_names"]) == [str(d["all_param_names"][i]) for i in free]
        assert rel_err(r["popt"].T, d["popt"]).max() <= 1e-8


def _scipy_fit(fun, b, y, p0, lo, hi, max_nfev=250, tol=1e-8):
    from scipy.optimize import curve_fit

    try:
        popt, pcov = curve_fit(fun, b, y, p0=p0, bounds=(lo, hi), method="trf", maxfev=max_nfev, ftol=tol)
        return popt, True
    except Exception:
        return np.asarray(p0, float), False


def test_oracle_matches_scipy_fresh_data(oracle):
    """Independent of the fixtures: new seeds, SciPy called directly."""
    from pyneapple_amd import synth

    def tri(x, f1, D1, f2, D2, D3):
        return f1 * np.exp(-x * D1) + f2 * np.exp(-x * D2) + (1 - f1 - f2) * np.exp(-x * D3)

    b, y, _ = synth.make_numpy("tri_reduced", 48, 32, sigma=0.02, seed=7)
    _, p0, lo, hi = synth.shared_arrays("tri_reduced")
    r = oracle.curvefit("tri_reduced", b, y, p0, lo, hi)
    ref = np.array([_scipy_fit(tri, b, y[i], p0, lo, hi)[0] for i in range(len(y))])
    assert (rel_err(r["popt"].T, ref).max(axis=1) <= RTOL).mean() >= 0.97


def _p0_on_bounds_case():
    """biexp, clinical (non-uniform) b-values, start values sitting exactly on a lower and on an upper bound:
    least_squares moves them strictly inside first (least_squares.py:827-828, rstep = 1e-10)."""
    rng = np.random.default_rng(11)
    b = np.array([0, 5, 10, 20, 30, 40, 50, 75, 100, 150, 200, 400, 600, 800], float)
    n = 40
    f1, D1, D2 = rng.uniform(0.1, 0.4, n), rng.uniform(5e-3, 5e-2, n), rng.uniform(5e-4, 2e-3, n)
    y = (f1[:, None] * np.exp(-b * D1[:, None]) + (1 - f1[:, None]) * np.exp(-b * D2[:, None]))
    y *= 1 + 0.01 * rng.standard_normal(y.shape)
    p0 = np.array([0.0, 0.1, 0.001])   # f1 on its lower bound, D1 on its upper bound
    lo = np.array([0.0, 1e-3, 1e-5])
    hi = np.array([1.0, 0.1, 5e-3])
    return b, y, p0, lo, hi


def test_oracle_matches_scipy_p0_on_bounds(oracle):
    b, y, p0, lo, hi = _p0_on_bounds_case()

    def bi(x, f1, D1, D2):
        return f1 * np.exp(-x * D1) + (1 - f1) * np.exp(-x * D2)

    r = oracle.curvefit("bi_reduced", b, y, p0, lo, hi)
    fits = [_scipy_fit(bi, b, y[i], p0, lo, hi) for i in range(len(y))]
    ref = np.array([f[0] for f in fits])
    assert ((r["status"] > 0) == np.array([f[1] for f in fits])).all()
    assert (rel_err(r["popt"].T, ref).max(axis=1) <= RTOL).mean() >= 0.97


def test_oracle_failure_sentinels(oracle):
    from pyneapple_amd import synth

    b, y, _ = synth.make_numpy("bi_reduced", 8, 24, sigma=0.01, seed=3)
    _, p0, lo, hi = synth.shared_arrays("bi_reduced")
    y = y.copy()
    y[1, 3] = np.nan
    y[2, 0] = np.inf
    P0 = np.repeat(p0[:, None], 8, axis=1)
    LO = np.repeat(lo[:, None], 8, axis=1)
    HI = np.repeat(hi[:, None], 8, axis=1)
    P0[0, 3] = 2.0          # outside bounds
    LO[1, 4] = HI[1, 4]      # lb == ub
    r = oracle.curvefit("bi_reduced", b, y, P0, LO, HI)
    assert list(r["status"][[1, 2, 3, 4]]) == [-2, -2, -3, -1]
    for v in (1, 2, 3, 4):
        assert np.array_equal(r["popt"][:, v], P0[:, v]) and np.isnan(r["pcov"][v]).all()
    assert (r["status"][[0, 5, 6, 7]] > 0).all()
    # max_nfev exhausted -> status 0, p0 returned (SciPy raises RuntimeError, reference returns p0)
    r2 = oracle.curvefit("bi_reduced", b, y[[0]], p0, lo, hi, max_nfev=2)
    assert r2["status"][0] == 0 and np.array_equal(r2["popt"][:, 0], p0) and r2["nfev"][0] == 2


def test_scipy_vs_oracle_disagreement_rate_on_noisy_triexp(oracle):
    """The evidence behind the GPU suite's ">= 99.5 % of a large seeded set within rtol 1e-4" threshold
    (tests/test_gpu_curvefit.py): SciPy itself, called the way the reference calls it (curvefit.py:295-306), and the
    oracle disagree beyond 1e-4 on a fraction of a per cent of fresh noisy triexp voxels -- exp() differs in the last
    ulp between libm and NumPy, a discrete TRF decision flips, and a weakly determined voxel ends at an EQUALLY GOOD
    point a little further along its valley.  Every voxel has the same cost to 1e-6 relative (measured: 0.27 % beyond
    1e-4, worst cost difference 1.6e-7 -- the scale of ftol = 1e-8 times a few iterations)."""
    from pyneapple_amd import synth

    def tri(x, f1, D1, f2, D2, D3):
        return f1 * np.exp(-x * D1) + f2 * np.exp(-x * D2) + (1 - f1 - f2) * np.exp(-x * D3)

    n = 1500
    b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01, seed=20261004)
    _, p0, lo, hi = synth.shared_arrays("tri_reduced")
    r = oracle.curvefit("tri_reduced", b, y, p0, lo, hi, n_threads=4)
    fits = [_scipy_fit(tri, b, y[i], p0, lo, hi) for i in range(n)]
    ref = np.array([f[0] for f in fits])
    ok = np.array([f[1] for f in fits])
    assert ((r["status"] > 0) == ok).all()
    worst = rel_err(r["popt"].T, ref).max(axis=1)
    off = worst > RTOL
    assert off.mean() <= 0.01, f"{off.mean():.4f} of the voxels beyond rtol 1e-4"   # measured: ~0.2 %
    assert np.median(worst) < 1e-6
    cost_ref = 0.5 * ((tri(b[None, :], *ref.T[:, :, None]) - y) ** 2).sum(axis=1)
    assert (np.abs(r["cost"] - cost_ref) <= 1e-6 * cost_ref)[ok].all()               # equally good minima, every voxel
