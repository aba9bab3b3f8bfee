"""ctypes front-end of the CPU restatement (oracle).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (pyneapple_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libpnx_oracle.so")

MODELS = {
    "mono": 0, "bi_reduced": 1, "bi_s0": 2, "bi_full": 3,
    "tri_reduced": 4, "tri_s0": 5, "tri_full": 6,
}
MODEL_PARAM_NAMES = {
    "mono": ["S0", "D"],
    "bi_reduced": ["f1", "D1", "D2"],
    "bi_s0": ["f1", "D1", "D2", "S0"],
    "bi_full": ["f1", "D1", "f2", "D2"],
    "tri_reduced": ["f1", "D1", "f2", "D2", "D3"],
    "tri_s0": ["f1", "D1", "f2", "D2", "D3", "S0"],
    "tri_full": ["f1", "D1", "f2", "D2", "f3", "D3"],
}


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("pnx_oracle_trf.c", "pnx_oracle_nnls.c", "Makefile")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        _lib.pnxo_curvefit_batch.restype = C.c_int
        _lib.pnxo_curvefit_batch.argtypes = [
            C.c_int, C.c_int, C.c_double, C.c_double, C.c_long, C.c_int, dp, dp, C.c_int, ip, C.c_int, ip, dp, C.c_int,
            dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp,
            C.POINTER(C.c_int8), C.POINTER(C.c_int32), dp, C.c_int]
        _lib.pnxo_curvefit_batch_sigma.restype = C.c_int
        _lib.pnxo_curvefit_batch_sigma.argtypes = [
            C.c_int, C.c_int, C.c_double, C.c_double, C.c_long, C.c_int, dp, dp, C.c_int, ip, C.c_int, ip, dp, C.c_int,
            dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, dp, C.c_int, dp, dp,
            C.POINTER(C.c_int8), C.POINTER(C.c_int32), dp, C.c_int]
        _lib.pnxo_nnls_batch.restype = C.c_int
        _lib.pnxo_nnls_batch.argtypes = [
            C.c_long, C.c_int, C.c_int, dp, dp, C.c_int, dp, C.c_int, dp, dp,
            C.POINTER(C.c_int8), C.POINTER(C.c_int32), C.c_int]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def curvefit(model: str, b, y, p0, lo, hi, *, t1_mode=0, tr=0.0, tm=0.0, fixed_idx=(), fixed_vals=None,
             max_nfev=250, ftol=1e-8, xtol=1e-8, gtol=1e-8, jac="fd", want_pcov=True, n_threads=1, sigma=None,
             absolute_sigma=False):
    """Batched bounded NLLS (SciPy-TRF restatement).

    p0/lo/hi: (n_free,) shared or (n_free, n_vox) per voxel.  fixed_vals: (n_fixed,) or (n_fixed, n_vox).
    sigma: None, a scalar or (n_b,) -- curve_fit's 1-D sigma, shared by the voxels; absolute_sigma as in curve_fit.
    Returns dict(popt (n_free,n_vox), pcov (n_vox,n_free,n_free), status int8, nfev int32, cost).
    """
    L = lib()
    b = np.ascontiguousarray(b, np.float64)
    y = np.ascontiguousarray(np.atleast_2d(y), np.float64)
    n_vox, n_b = y.shape
    n_all = len(MODEL_PARAM_NAMES[model]) + (1 if t1_mode else 0)
    fixed_idx = np.ascontiguousarray(list(fixed_idx), np.int32)
    free_idx = np.ascontiguousarray([i for i in range(n_all) if i not in set(fixed_idx.tolist())], np.int32)
    n_free = len(free_idx)
    p0 = np.ascontiguousarray(p0, np.float64)
    lo = np.ascontiguousarray(lo, np.float64)
    hi = np.ascontiguousarray(hi, np.float64)
    per_voxel = int(p0.ndim == 2)
    if per_voxel:
        assert p0.shape == (n_free, n_vox) and lo.shape == p0.shape and hi.shape == p0.shape
    else:
        assert p0.shape == (n_free,) and lo.shape == p0.shape and hi.shape == p0.shape
    fv = None
    fpv = 0
    if len(fixed_idx):
        fv = np.ascontiguousarray(fixed_vals, np.float64)
        fpv = int(fv.ndim == 2)
    popt = np.empty((n_free, n_vox))
    pcov = np.empty((n_vox, n_free, n_free)) if want_pcov else None
    status = np.empty(n_vox, np.int8)
    nfev = np.empty(n_vox, np.int32)
    cost = np.empty(n_vox)
    if sigma is not None:
        sigma = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, np.float64).reshape(-1), (n_b,)))
    rc = L.pnxo_curvefit_batch_sigma(
        MODELS[model], t1_mode, tr, tm, n_vox, n_b, _dp(b), _dp(y), n_free,
        free_idx.ctypes.data_as(C.POINTER(C.c_int)), len(fixed_idx), fixed_idx.ctypes.data_as(C.POINTER(C.c_int)),
        _dp(fv), fpv, _dp(p0), _dp(lo), _dp(hi), per_voxel, max_nfev, ftol, xtol, gtol,
        0 if jac == "fd" else 1, _dp(sigma), int(bool(absolute_sigma)), _dp(popt), _dp(pcov),
        status.ctypes.data_as(C.POINTER(C.c_int8)), nfev.ctypes.data_as(C.POINTER(C.c_int32)), _dp(cost), n_threads)
    if rc != 0:
        raise ValueError(f"pnxo_curvefit_batch rc={rc}")
    return dict(popt=popt, pcov=pcov, status=status, nfev=nfev, cost=cost)


def nnls(basis, reg, y, max_iter, n_threads=1):
    """Batched Lawson-Hanson NNLS on A=[basis; reg], y_ext=[y | 0] (SciPy 1.15 `nnls` restatement)."""
    L = lib()
    basis = np.ascontiguousarray(basis, np.float64)
    n_meas, n_bins = basis.shape
    y = np.ascontiguousarray(np.atleast_2d(y), np.float64)
    n_vox = y.shape[0]
    assert y.shape[1] == n_meas
    n_reg = 0
    if reg is not None:
        reg = np.ascontiguousarray(reg, np.float64)
        n_reg = reg.shape[0]
        assert reg.shape[1] == n_bins
    coeff = np.empty((n_vox, n_bins))
    rnorm = np.empty(n_vox)
    status = np.empty(n_vox, np.int8)
    iters = np.empty(n_vox, np.int32)
    rc = L.pnxo_nnls_batch(n_vox, n_meas, n_bins, _dp(basis), _dp(reg), n_reg, _dp(y), max_iter, _dp(coeff),
                           _dp(rnorm), status.ctypes.data_as(C.POINTER(C.c_int8)),
                           iters.ctypes.data_as(C.POINTER(C.c_int32)), n_threads)
    if rc != 0:
        raise ValueError(f"pnxo_nnls_batch rc={rc}")
    return dict(coefficients=coeff, residual=rnorm, status=status, iters=iters)
