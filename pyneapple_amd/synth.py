"""Synthetic DWI volumes of the shapes BASELINE.json names.

Ground truth ranges, p0 and bounds are the ones listed there (they mirror the reference's example
configs).  `numpy` variants are used by tests / the CPU baseline, `torch` variants generate straight
into HBM for bench.py.
"""
from __future__ import annotations

import numpy as np

SEED = 20260503

WORKLOADS = {
    # name: (model, n_b, volume shape)
    "mono": ("mono", 16, (32, 32, 1)),
    "biexp": ("bi_reduced", 24, (128, 128, 32)),
    "triexp": ("tri_reduced", 32, (256, 256, 64)),
    "nnls": ("nnls", 32, (256, 256, 64)),
}

TRUTH = {
    "mono": {"S0": (500, 1500), "D": (5e-4, 3e-3)},
    "bi_reduced": {"f1": (0.1, 0.4), "D1": (5e-3, 5e-2), "D2": (5e-4, 2e-3)},
    "tri_reduced": {"f1": (0.1, 0.3), "D1": (0.03, 0.1), "f2": (0.2, 0.4), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)},
}
P0 = {
    "mono": {"S0": 1000.0, "D": 1e-3},
    "bi_reduced": {"f1": 0.2, "D1": 0.01, "D2": 0.001},
    "tri_reduced": {"f1": 0.2, "D1": 0.05, "f2": 0.3, "D2": 0.005, "D3": 0.001},
}
BOUNDS = {
    "mono": {"S0": (1.0, 5000.0), "D": (1e-5, 0.1)},
    "bi_reduced": {"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)},
    "tri_reduced": {"f1": (0.0, 1.0), "D1": (0.01, 0.5), "f2": (0.0, 1.0), "D2": (2e-3, 0.01), "D3": (1e-5, 2e-3)},
}
NNLS_CFG = dict(d_range=(0.0008, 0.5), n_bins=250, reg_order=2, mu=0.02, max_iter=250)


def bvalues(n_b: int) -> np.ndarray:
    return np.linspace(0.0, 1200.0, n_b)


def _signal(xp, model, b, P):
    e = lambda D: xp.exp(-b[None, :] * D[:, None])
    if model == "mono":
        return P["S0"][:, None] * e(P["D"])
    if model == "bi_reduced":
        return P["f1"][:, None] * e(P["D1"]) + (1 - P["f1"])[:, None] * e(P["D2"])
    if model == "tri_reduced":
        return (P["f1"][:, None] * e(P["D1"]) + P["f2"][:, None] * e(P["D2"])
                + (1 - P["f1"] - P["f2"])[:, None] * e(P["D3"]))
    raise ValueError(model)


def make_numpy(model: str, n_vox: int, n_b: int, sigma: float = 0.01, seed: int = SEED, scale: float = 1.0):
    """Returns (b, y (n_vox, n_b) float64, truth dict)."""
    rng = np.random.default_rng(seed)
    b = bvalues(n_b)
    P = {k: rng.uniform(lo, hi, n_vox) for k, (lo, hi) in TRUTH[model].items()}
    y = _signal(np, model, b, P) * scale
    if sigma:
        y = y * (1.0 + sigma * rng.standard_normal(y.shape))
    return b, np.ascontiguousarray(y), P


ROW_CHUNK = 1 << 18  # rows per independently seeded block of a synthetic volume


def make_torch_rows(model: str, start: int, stop: int, n_b: int, device, sigma: float = 0.01, seed: int = SEED,
                    scale: float = 1.0, dtype=None):
    """Rows [start, stop) of ONE seed-fixed synthetic volume, generated on `device` (float64).

    The volume is defined block-wise: block c (rows c*ROW_CHUNK ...) is drawn from a generator seeded with
    (seed, c), so any rank can produce any row range without the others, and the concatenation of the ranks'
    shards is bit-identical to the volume a single process generates (bench.py --gpus N: strong scaling)."""
    import torch

    dtype = dtype or torch.float64
    b = torch.linspace(0.0, 1200.0, n_b, dtype=torch.float64, device=device)
    y = torch.empty((max(0, stop - start), n_b), dtype=dtype, device=device)
    g = torch.Generator(device=device)
    for c in range(start // ROW_CHUNK, (max(stop, start + 1) - 1) // ROW_CHUNK + 1):
        if stop <= start:
            break
        g.manual_seed(seed * 1000003 + c)
        n = ROW_CHUNK
        P = {k: lo + (hi - lo) * torch.rand(n, generator=g, dtype=torch.float64, device=device)
             for k, (lo, hi) in TRUTH[model].items()}
        sig = _signal(torch, model, b, P) * scale
        if sigma:
            sig = sig * (1.0 + sigma * torch.randn(sig.shape, generator=g, dtype=torch.float64, device=device))
        a, e = max(start, c * ROW_CHUNK), min(stop, (c + 1) * ROW_CHUNK)
        y[a - start:e - start] = sig[a - c * ROW_CHUNK:e - c * ROW_CHUNK].to(dtype)
    return bvalues(n_b), y


def make_torch(model: str, n_vox: int, n_b: int, device, sigma: float = 0.01, seed: int = SEED, scale: float = 1.0):
    """The whole volume (rows [0, n_vox)) on `device`."""
    return make_torch_rows(model, 0, n_vox, n_b, device, sigma, seed, scale)


def shared_arrays(model: str):
    names = list(P0[model].keys())
    p0 = np.array([P0[model][n] for n in names], float)
    lo = np.array([BOUNDS[model][n][0] for n in names], float)
    hi = np.array([BOUNDS[model][n][1] for n in names], float)
    return names, p0, lo, hi


def nnls_matrices(n_b: int, cfg=NNLS_CFG):
    """bins / basis / regulariser exactly as model_functions/nnls.py:17-85 builds them (numpy, host)."""
    d0, d1 = cfg["d_range"]
    n = cfg["n_bins"]
    bins = np.logspace(np.log10(d0), np.log10(d1), n)
    b = bvalues(n_b)
    basis = np.exp(-b.reshape(-1, 1) * bins.reshape(1, -1))
    order, mu = cfg["reg_order"], cfg["mu"]
    if order == 0:
        reg = np.zeros((n, n))
    elif order == 1:
        reg = (np.diag(np.full(n, -1.0)) + np.diag(np.ones(n - 1), 1)) * mu
    elif order == 2:
        reg = (np.diag(np.ones(n - 1), -1) + np.diag(np.full(n, -2.0)) + np.diag(np.ones(n - 1), 1)) * mu
    elif order == 3:
        reg = (np.diag(np.ones(n - 2), -2) + np.diag(np.full(n - 1, 2.0), -1) + np.diag(np.full(n, -6.0))
               + np.diag(np.full(n - 1, 2.0), 1) + np.diag(np.ones(n - 2), 2)) * mu
    else:
        raise NotImplementedError(order)
    return bins, basis, reg
