"""Pins oracle/pnx_oracle_nnls.c against the reference's golden outputs and scipy.optimize.nnls."""
from __future__ import annotations

import numpy as np
import pytest
from conftest import NNLS_FIXTURES, load_golden


@pytest.mark.parametrize("name", NNLS_FIXTURES)
def test_oracle_nnls_matches_reference_golden(oracle, name):
    d = load_golden(name)
    r = oracle.nnls(d["basis"], d["reg"], d["y"], int(d["max_iter"]))
    assert ((r["status"] == 1) == d["success"]).all()
    c, cr = r["coefficients"], d["coefficients"]
    scale = np.abs(cr).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(c - cr) / scale).max() < 1e-9
    np.testing.assert_allclose(r["residual"], d["residual"], rtol=1e-10)
    assert (((c > 0) != (cr > 0)).sum(axis=1) == 0).all()  # identical support


def test_oracle_nnls_vs_scipy_and_iteration_limit(oracle):
    from scipy.optimize import nnls

    d = load_golden("g4_nnls_50_r2")
    A = np.vstack([d["basis"], d["reg"]])
    for v in range(4):
        y_ext = np.concatenate([d["y"][v], np.zeros(A.shape[1])])
        full = oracle.nnls(d["basis"], d["reg"], d["y"][v:v + 1], 100000)
        it = int(full["iters"][0])
        x, rn = nnls(A, y_ext, maxiter=100000)
        np.testing.assert_allclose(full["coefficients"][0], x, rtol=0, atol=1e-10 * np.abs(x).max())
        np.testing.assert_allclose(full["residual"][0], rn, rtol=1e-12)
        # SciPy 1.15 fails when its iteration counter reaches maxiter: smallest passing maxiter = iters + 1
        for mi, expect_ok in ((it, False), (it + 1, True)):
            try:
                nnls(A, y_ext, maxiter=mi)
                ok = True
            except RuntimeError:
                ok = False
            assert ok == expect_ok
            assert bool(oracle.nnls(d["basis"], d["reg"], d["y"][v:v + 1], mi)["status"][0] == 1) == expect_ok


def test_oracle_nnls_failure_path(oracle):
    d = load_golden("g4_nnls_50_r2")
    y = d["y"][:3].copy()
    y[1, 2] = np.nan
    r = oracle.nnls(d["basis"], d["reg"], y, 5)
    assert r["status"][0] == 0 and r["status"][1] == -2
    assert (r["coefficients"][:2] == 0).all()
    np.testing.assert_allclose(r["residual"][0], np.linalg.norm(y[0]))  # ||y_ext|| = ||y||
    assert np.isnan(r["residual"][1])
