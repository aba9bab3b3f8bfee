#!/usr/bin/env python3
"""Generate golden input/output vectors by running the *reference* itself.

TEST INFRASTRUCTURE ONLY.  This script imports darksim33/Pyneapple from
``/root/reference/src`` (read-only mount, survey container only) with two
in-process stand-ins for the absent ``loguru`` / ``cv2`` modules (SURVEY.md
Appendix B), runs the reference's own ``CurveFitSolver`` / ``NNLSSolver`` on
seeded synthetic inputs and stores inputs + outputs as small ``.npz`` fixtures
under ``tests/golden/``.  Only data is written; no reference source travels.

Run (from any cwd):  python3 oracle/gen_golden.py            (batches g1-g9)
                     python3 oracle/gen_golden.py g10 | g11 | g12   (later batches, each with its own seed)
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np

REF_SRC = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEED = 20260503


def _install_shims():
    class _Noop:
        def __getattr__(self, name):
            return lambda *a, **k: None

    loguru = types.ModuleType("loguru")
    loguru.logger = _Noop()
    sys.modules["loguru"] = loguru
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.INTER_CUBIC = 2
    sys.modules["cv2"] = cv2
    for name in ("nibabel", "h5py"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)


def _versions():
    import scipy

    return dict(scipy_version=scipy.__version__, numpy_version=np.__version__)


def _pack_curvefit(solver, n_params):
    prs = solver.pixel_results_
    popt = np.array([pr.params for pr in prs])  # (n_px, n_free)
    pcov = np.array(
        [pr.covariance if pr.covariance is not None else np.full((n_params, n_params), np.nan) for pr in prs]
    )
    success = np.array([pr.success for pr in prs])
    return popt, pcov, success


def _noise(rng, clean, sigma):
    return clean * (1.0 + sigma * rng.standard_normal(clean.shape))


def gen_curvefit(name, model, names, truth_ranges, p0, bounds, nb, n_vox, sigmas, rng,
                 max_iter=250, tol=1e-8, bvalues=None, extra=None, per_voxel=False,
                 fixed=None, scale=1.0, solver_kwargs=None):
    from pyneapple import CurveFitSolver

    b = np.linspace(0.0, 1200.0, nb) if bvalues is None else np.asarray(bvalues, float)
    all_names = model._all_param_names
    truth = np.empty((n_vox, len(all_names)))
    for j, nm in enumerate(all_names):
        lo, hi = truth_ranges[nm]
        truth[:, j] = rng.uniform(lo, hi, n_vox)
    if extra is not None:
        extra(truth)
    clean = np.array([model.forward(b, *truth[i]) for i in range(n_vox)]) * scale
    sig = np.empty(n_vox)
    parts = np.array_split(np.arange(n_vox), len(sigmas))
    for s, idx in zip(sigmas, parts):
        sig[idx] = s
    y = clean * (1.0 + sig[:, None] * rng.standard_normal(clean.shape))

    solver = CurveFitSolver(model=model, max_iter=max_iter, tol=tol, p0=p0, bounds=bounds, **(solver_kwargs or {}))
    kw = {}
    out = dict(bvalues=b, y=y, truth=truth, sigma=sig, max_iter=max_iter, tol=tol,
               param_names=np.array(model.param_names), all_param_names=np.array(all_names))
    if solver_kwargs:  # what the reference forwards into scipy.optimize.curve_fit (solvers/curvefit.py:295-306): sigma, absolute_sigma
        if solver_kwargs.get("sigma") is not None:
            out["fit_sigma"] = np.atleast_1d(np.asarray(solver_kwargs["sigma"], float))
        out["absolute_sigma"] = bool(solver_kwargs.get("absolute_sigma", False))
    free_names = list(model.param_names)
    if per_voxel:
        # IDEAL-style per-voxel p0 / bounds arrays (n_params, n_px); fitters/ideal.py:182-189
        p0v = np.array([p0[n] for n in names])[:, None] * (1.0 + 0.2 * rng.uniform(-1, 1, (len(names), n_vox)))
        lo = np.array([bounds[n][0] for n in names])[:, None]
        hi = np.array([bounds[n][1] for n in names])[:, None]
        p0v = np.clip(p0v, lo, hi)
        lov = np.clip(p0v * 0.5, lo, hi)
        hiv = np.clip(p0v * 1.5, lo, hi)
        bad = lov >= hiv
        lov[bad] = np.broadcast_to(lo, lov.shape)[bad]
        hiv[bad] = np.broadcast_to(hi, hiv.shape)[bad]
        kw.update(p0=p0v, bounds=(lov, hiv))
        out.update(p0_arr=p0v, lo_arr=lov, hi_arr=hiv)
    if fixed is not None:
        fx = {k: rng.uniform(*v, n_vox) for k, v in fixed.items()}
        # make data consistent with the fixed maps
        for k, v in fx.items():
            truth[:, all_names.index(k)] = v
        clean = np.array([model.forward(b, *truth[i]) for i in range(n_vox)]) * scale
        y = clean * (1.0 + sig[:, None] * rng.standard_normal(clean.shape))
        out.update(y=y, truth=truth)
        kw.update(pixel_fixed_params=fx)
        for k, v in fx.items():
            out["fixed_" + k] = v
        free_names = [n for n in free_names if n not in fx]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        solver.fit(b, y, **kw)
    popt, pcov, success = _pack_curvefit(solver, len(free_names))
    out.update(popt=popt, pcov=pcov, success=success, free_names=np.array(free_names),
               p0_names=np.array(names), p0_vals=np.array([p0[n] for n in names], float),
               lo_vals=np.array([bounds[n][0] for n in names], float),
               hi_vals=np.array([bounds[n][1] for n in names], float), **_versions())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: n_vox={n_vox} success={success.mean():.3f}")


def gen_nnls(name, d_range, n_bins, reg_order, mu, nb, n_vox, rng, max_iter=250, sigma=0.01):
    from pyneapple import NNLSModel, NNLSSolver, TriExpModel

    b = np.linspace(0.0, 1200.0, nb)
    tri = TriExpModel()
    truth = np.column_stack([
        rng.uniform(0.1, 0.3, n_vox), rng.uniform(0.03, 0.1, n_vox), rng.uniform(0.2, 0.4, n_vox),
        rng.uniform(3e-3, 8e-3, n_vox), rng.uniform(5e-4, 1.5e-3, n_vox)])
    clean = np.array([tri.forward(b, *truth[i]) for i in range(n_vox)]) * 1000.0
    y = _noise(rng, clean, sigma)
    model = NNLSModel(d_range=d_range, n_bins=n_bins)
    solver = NNLSSolver(model=model, reg_order=reg_order, mu=mu, max_iter=max_iter)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        solver.fit(b, y)
    A = solver._build_regularized_basis(b)
    success = np.array([pr.success for pr in solver.pixel_results_])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), bvalues=b, y=y, d_range=np.array(d_range), n_bins=n_bins,
        reg_order=reg_order, mu=mu, max_iter=max_iter, bins=model.bins, basis=model.get_basis(b),
        reg=solver.get_regularization_matrix(), A_checksum=np.array([A.sum(), (A * A).sum()]),
        coefficients=solver.params_["coefficients"], residual=solver.diagnostics_["residual"],
        success=success, **_versions())
    print(f"{name}: n_vox={n_vox} success={success.mean():.3f} nnz~{(solver.params_['coefficients']>0).sum(1).mean():.1f}")


def main():
    os.environ.setdefault("PYNEAPPLE_QUIET", "1")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_SRC)
    _install_shims()
    os.makedirs(OUT, exist_ok=True)
    from pyneapple import BiExpModel, MonoExpModel, TriExpModel

    rng = np.random.default_rng(SEED)
    B16 = [0, 5, 10, 20, 30, 40, 50, 75, 100, 150, 200, 250, 350, 450, 550, 650]  # tests/test_toolbox.py:18-21 is the analogous 16-pt set
    B8 = [0, 50, 100, 200, 400, 600, 800, 1000]  # tests/test_solver_curvefit.py:15

    mono_p0 = {"S0": 1000.0, "D": 1e-3}
    mono_bd = {"S0": (1.0, 5000.0), "D": (1e-5, 0.1)}
    mono_tr = {"S0": (500, 1500), "D": (5e-4, 3e-3)}
    gen_curvefit("g1_mono_b16", MonoExpModel(), ["S0", "D"], mono_tr, mono_p0, mono_bd, 16, 128, [0.0, 0.01], rng,
                 bvalues=np.linspace(0, 1200, 16))
    gen_curvefit("g1_mono_b8", MonoExpModel(), ["S0", "D"], mono_tr, mono_p0, mono_bd, 8, 128, [0.0, 0.01], rng,
                 bvalues=B8)

    bi_p0 = {"f1": 0.2, "D1": 0.01, "D2": 0.001}
    bi_bd = {"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}
    bi_tr = {"f1": (0.1, 0.4), "D1": (5e-3, 5e-2), "D2": (5e-4, 2e-3)}
    gen_curvefit("g2_bi_reduced", BiExpModel(), ["f1", "D1", "D2"], bi_tr, bi_p0, bi_bd, 24, 192, [0.0, 0.01, 0.05], rng)
    gen_curvefit("g2_bi_s0", BiExpModel(fit_s0=True), ["f1", "D1", "D2", "S0"], dict(bi_tr, S0=(500, 1500)),
                 dict(bi_p0, S0=1000.0), dict(bi_bd, S0=(1.0, 5000.0)), 24, 128, [0.0, 0.01], rng)
    gen_curvefit("g2_bi_full", BiExpModel(fit_reduced=False), ["f1", "D1", "f2", "D2"],
                 {"f1": (100, 400), "D1": (5e-3, 5e-2), "f2": (500, 900), "D2": (5e-4, 2e-3)},
                 {"f1": 200.0, "D1": 0.01, "f2": 800.0, "D2": 0.001},
                 {"f1": (0.0, 2000.0), "D1": (1e-3, 0.1), "f2": (0.0, 2000.0), "D2": (1e-5, 5e-3)},
                 24, 128, [0.0, 0.01], rng)

    tri_p0 = {"f1": 0.2, "D1": 0.05, "f2": 0.3, "D2": 0.005, "D3": 0.001}
    tri_bd = {"f1": (0.0, 1.0), "D1": (0.01, 0.5), "f2": (0.0, 1.0), "D2": (2e-3, 0.01), "D3": (1e-5, 2e-3)}
    tri_tr = {"f1": (0.1, 0.3), "D1": (0.03, 0.1), "f2": (0.2, 0.4), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)}
    tri_names = ["f1", "D1", "f2", "D2", "D3"]
    gen_curvefit("g3_tri_reduced", TriExpModel(), tri_names, tri_tr, tri_p0, tri_bd, 32, 384, [0.0, 0.01, 0.03], rng)
    gen_curvefit("g3_tri_reduced_maxiter4", TriExpModel(), tri_names, tri_tr, tri_p0, tri_bd, 32, 96, [0.01], rng,
                 max_iter=4)
    gen_curvefit("g3_tri_s0", TriExpModel(fit_s0=True), tri_names + ["S0"], dict(tri_tr, S0=(500, 1500)),
                 dict(tri_p0, S0=1000.0), dict(tri_bd, S0=(1.0, 5000.0)), 32, 128, [0.0, 0.01], rng)
    gen_curvefit("g3_tri_full", TriExpModel(fit_reduced=False), ["f1", "D1", "f2", "D2", "f3", "D3"],
                 {"f1": (100, 300), "D1": (0.03, 0.1), "f2": (200, 400), "D2": (3e-3, 8e-3), "f3": (300, 700), "D3": (5e-4, 1.5e-3)},
                 {"f1": 200.0, "D1": 0.05, "f2": 300.0, "D2": 0.005, "f3": 500.0, "D3": 0.001},
                 {"f1": (0.0, 2000.0), "D1": (0.01, 0.5), "f2": (0.0, 2000.0), "D2": (2e-3, 0.01), "f3": (0.0, 2000.0), "D3": (1e-5, 2e-3)},
                 32, 128, [0.0, 0.01], rng)

    # G5: per-voxel p0 / bounds arrays (IDEAL-style)
    gen_curvefit("g5_bi_pervoxel", BiExpModel(), ["f1", "D1", "D2"], bi_tr, bi_p0, bi_bd, 24, 128, [0.0, 0.01], rng,
                 per_voxel=True)
    gen_curvefit("g5_tri_pervoxel", TriExpModel(), tri_names, tri_tr, tri_p0, tri_bd, 32, 128, [0.0, 0.01], rng,
                 per_voxel=True)

    # G6: fixed parameters (analytic-Jacobian path, curvefit.py:274-288)
    gen_curvefit("g6_bi_fixed_D1", BiExpModel(), ["f1", "D1", "D2"], bi_tr, bi_p0, bi_bd, 24, 96, [0.0, 0.01], rng,
                 fixed={"D1": (5e-3, 5e-2)})
    gen_curvefit("g6_mono_t1_fixed", MonoExpModel(fit_t1=True, repetition_time=3000.0), ["S0", "D", "T1"],
                 dict(mono_tr, T1=(800, 1600)), dict(mono_p0, T1=1000.0), dict(mono_bd, T1=(100.0, 5000.0)),
                 16, 96, [0.0, 0.01], rng, fixed={"T1": (800, 1600)})

    # G4: NNLS
    gen_nnls("g4_nnls_250_r2", (0.0008, 0.5), 250, 2, 0.02, 32, 64, rng)
    gen_nnls("g4_nnls_250_r1", (0.0008, 0.5), 250, 1, 0.02, 32, 32, rng)
    gen_nnls("g4_nnls_250_r3", (0.0008, 0.5), 250, 3, 0.02, 32, 32, rng)
    gen_nnls("g4_nnls_50_r2", (1e-4, 0.1), 50, 2, 0.02, 16, 64, rng)
    gen_nnls("g4_nnls_50_r0", (1e-4, 0.1), 50, 0, 0.02, 16, 32, rng)
    gen_nnls("g4_nnls_250_r2_maxiter20", (0.0008, 0.5), 250, 2, 0.02, 32, 32, rng, max_iter=20)


def main_g7():
    """Second batch (own generator, so the first batch stays byte-identical): free T1 / STEAM factors through the
    reference's default FD Jacobian, and infinite bounds (Coleman-Li scaling with v = 1 on unbounded sides)."""
    from pyneapple import BiExpModel, MonoExpModel

    rng = np.random.default_rng(SEED + 7)
    mono_p0 = {"S0": 1000.0, "D": 1e-3}
    mono_bd = {"S0": (1.0, 5000.0), "D": (1e-5, 0.1)}
    mono_tr = {"S0": (500, 1500), "D": (5e-4, 3e-3)}
    gen_curvefit("g7_mono_t1_free", MonoExpModel(fit_t1=True, repetition_time=3000.0), ["S0", "D", "T1"],
                 dict(mono_tr, T1=(800, 1600)), dict(mono_p0, T1=1000.0), dict(mono_bd, T1=(100.0, 5000.0)),
                 16, 96, [0.0, 0.01], rng)
    bi_p0 = {"f1": 0.2, "D1": 0.01, "D2": 0.001}
    bi_bd = {"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}
    bi_tr = {"f1": (0.1, 0.4), "D1": (5e-3, 5e-2), "D2": (5e-4, 2e-3)}
    gen_curvefit("g7_bi_s0_steam_free", BiExpModel(fit_s0=True, fit_t1=True, fit_t1_steam=True, repetition_time=2500.0, mixing_time=30.0),
                 ["f1", "D1", "D2", "S0", "T1"], dict(bi_tr, S0=(500, 1500), T1=(800, 1600)),
                 dict(bi_p0, S0=1000.0, T1=1000.0), dict(bi_bd, S0=(1.0, 5000.0), T1=(100.0, 5000.0)), 24, 96, [0.0, 0.01], rng)
    inf = float("inf")
    gen_curvefit("g7_mono_inf_bounds", MonoExpModel(), ["S0", "D"], mono_tr, mono_p0, {"S0": (0.0, inf), "D": (-inf, inf)},
                 16, 96, [0.0, 0.01], rng, bvalues=np.linspace(0, 1200, 16))
    gen_curvefit("g7_bi_half_inf_bounds", BiExpModel(), ["f1", "D1", "D2"], bi_tr, bi_p0,
                 {"f1": (-inf, inf), "D1": (1e-3, inf), "D2": (-inf, 5e-3)}, 24, 128, [0.0, 0.01, 0.05], rng)


def main_g8():
    """Third batch: two per-pixel fixed parameters (SegmentedFitter with two maps from step 1; analytic Jacobian)."""
    from pyneapple import MonoExpModel, TriExpModel

    rng = np.random.default_rng(SEED + 8)
    tri_p0 = {"f1": 0.2, "D1": 0.05, "f2": 0.3, "D2": 0.005, "D3": 0.001}
    tri_bd = {"f1": (0.0, 1.0), "D1": (0.01, 0.5), "f2": (0.0, 1.0), "D2": (2e-3, 0.01), "D3": (1e-5, 2e-3)}
    tri_tr = {"f1": (0.1, 0.3), "D1": (0.03, 0.1), "f2": (0.2, 0.4), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)}
    gen_curvefit("g8_tri_fixed_D2_D3", TriExpModel(), ["f1", "D1", "f2", "D2", "D3"], tri_tr, tri_p0, tri_bd, 32, 96,
                 [0.0, 0.01], rng, fixed={"D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)})
    mono_p0 = {"S0": 1000.0, "D": 1e-3, "T1": 1000.0}
    mono_bd = {"S0": (1.0, 5000.0), "D": (1e-5, 0.1), "T1": (100.0, 5000.0)}
    mono_tr = {"S0": (500, 1500), "D": (5e-4, 3e-3), "T1": (800, 1600)}
    gen_curvefit("g8_mono_t1_fixed_S0_T1", MonoExpModel(fit_t1=True, repetition_time=3000.0), ["S0", "D", "T1"], mono_tr,
                 mono_p0, mono_bd, 16, 64, [0.0, 0.01], rng, fixed={"S0": (500, 1500), "T1": (800, 1600)})


def main_g9():
    """Fourth batch (round 2): the reference's own PixelWiseFitter result assembly (fitters/base.py:142-253), the
    reference-default 250-bin reg_order=0 NNLS, three fixed parameters, and NNLS spectrum post-processing
    (utility/spectrum.py:13-139)."""
    from pyneapple import BiExpModel, CurveFitSolver, NNLSModel, NNLSSolver, PixelWiseFitter, TriExpModel
    from pyneapple.utility.spectrum import apply_cutoffs, find_spectrum_peaks

    rng = np.random.default_rng(SEED + 9)
    # --- f-2: PixelWiseFitter(CurveFitSolver) with a mask and one NaN voxel
    b = np.linspace(0.0, 1200.0, 24)
    shape = (6, 5, 2)
    f1 = rng.uniform(0.1, 0.4, shape)
    D1 = rng.uniform(5e-3, 5e-2, shape)
    D2 = rng.uniform(5e-4, 2e-3, shape)
    img = f1[..., None] * np.exp(-b * D1[..., None]) + (1 - f1[..., None]) * np.exp(-b * D2[..., None])
    img = img * (1.0 + 0.01 * rng.standard_normal(img.shape))
    seg = np.zeros(shape, dtype=int)
    seg[1:5, :, :] = 1
    seg[2, 2, 0] = 0
    img[3, 1, 1, 4] = np.nan          # inside the mask: curve_fit raises -> failure sentinel, R^2 NaN
    img[1, 0, 0, :] = 0.75            # constant signal inside the mask: SS_tot = 0 -> R^2 NaN (base.py:182)
    bi_p0 = {"f1": 0.2, "D1": 0.01, "D2": 0.001}
    bi_bd = {"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}

    def run(fixed_maps):
        solver = CurveFitSolver(model=BiExpModel(), max_iter=250, tol=1e-8, p0=bi_p0, bounds=bi_bd)
        fit = PixelWiseFitter(solver)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fit.fit(b, img, segmentation=seg, fixed_param_maps=fixed_maps)
        r = fit.results_
        return dict(bvalues=b, image=img, segmentation=seg, r_squared=r.r_squared, success=r.success,
                    pixel_indices=np.array(r.pixel_indices), covariance=r.covariance, n_pixels=r.n_pixels,
                    param_names=np.array(list(r.params)), params=np.stack([np.asarray(v) for v in r.params.values()]),
                    messages_none=np.array([m is None for m in (r.messages or [None] * r.n_pixels)]),
                    p0_vals=np.array(list(bi_p0.values())), lo_vals=np.array([v[0] for v in bi_bd.values()]),
                    hi_vals=np.array([v[1] for v in bi_bd.values()]), **_versions())

    out = run(None)
    np.savez_compressed(os.path.join(OUT, "g9_pixelwise_bi.npz"), **out)
    print(f"g9_pixelwise_bi: n_px={out['n_pixels']} success={out['success'].mean():.3f} r2 nan={np.isnan(out['r_squared']).sum()}")
    out = run({"D1": D1})
    np.savez_compressed(os.path.join(OUT, "g9_pixelwise_bi_fixedmap.npz"), fixed_D1=D1, **out)
    print(f"g9_pixelwise_bi_fixedmap: n_px={out['n_pixels']} success={out['success'].mean():.3f} r2 nan={np.isnan(out['r_squared']).sum()}")

    # --- f-2: PixelWiseFitter(NNLSSolver): residuals and the basis-path R^2 (base.py:162-169)
    bn = np.linspace(0.0, 1200.0, 16)
    tri = TriExpModel()
    shape_n = (4, 3, 2)
    n = int(np.prod(shape_n))
    truth = np.column_stack([rng.uniform(0.1, 0.3, n), rng.uniform(0.03, 0.1, n), rng.uniform(0.2, 0.4, n),
                             rng.uniform(3e-3, 8e-3, n), rng.uniform(5e-4, 1.5e-3, n)])
    imgn = (np.array([tri.forward(bn, *t) for t in truth]) * 1000.0).reshape(*shape_n, 16)
    imgn = imgn * (1.0 + 0.01 * rng.standard_normal(imgn.shape))
    segn = np.ones(shape_n, dtype=int)
    segn[0, 0, :] = 0
    model = NNLSModel(d_range=(1e-4, 0.1), n_bins=50)
    fit = PixelWiseFitter(NNLSSolver(model=model, reg_order=2, mu=0.02, max_iter=250))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fit.fit(bn, imgn, segmentation=segn)
    r = fit.results_
    np.savez_compressed(os.path.join(OUT, "g9_pixelwise_nnls.npz"), bvalues=bn, image=imgn, segmentation=segn,
                        d_range=np.array([1e-4, 0.1]), n_bins=50, reg_order=2, mu=0.02, max_iter=250,
                        r_squared=r.r_squared, success=r.success, residuals=r.residuals,
                        pixel_indices=np.array(r.pixel_indices), coefficients=r.params["coefficients"],
                        covariance_is_none=np.array(r.covariance is None), **_versions())
    print(f"g9_pixelwise_nnls: n_px={r.n_pixels} mean r2={np.nanmean(r.r_squared):.6f}")

    # --- the reference's DEFAULT regulariser: reg_order=0 (nnls_solver.py:37), 250 bins -- rank deficient A
    gen_nnls("g9_nnls_250_r0", (0.0008, 0.5), 250, 0, 0.02, 32, 48, rng)

    # --- three fixed parameters (SegmentedFitter step 2 with all diffusivities from step 1; models/base.py:145-230)
    tri_tr = {"f1": (0.1, 0.3), "D1": (0.03, 0.1), "f2": (0.2, 0.4), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3), "S0": (500, 1500)}
    tri_p0 = {"f1": 0.2, "D1": 0.05, "f2": 0.3, "D2": 0.005, "D3": 0.001, "S0": 1000.0}
    tri_bd = {"f1": (0.0, 1.0), "D1": (0.01, 0.5), "f2": (0.0, 1.0), "D2": (2e-3, 0.01), "D3": (1e-5, 2e-3), "S0": (1.0, 5000.0)}
    gen_curvefit("g9_tri_s0_fixed_D1_D2_D3", TriExpModel(fit_s0=True), ["f1", "D1", "f2", "D2", "D3", "S0"], tri_tr, tri_p0,
                 tri_bd, 32, 96, [0.0, 0.01], rng, fixed={"D1": (0.03, 0.1), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)})
    full_tr = {"f1": (100, 300), "D1": (0.03, 0.1), "f2": (200, 400), "D2": (3e-3, 8e-3), "f3": (300, 700), "D3": (5e-4, 1.5e-3)}
    full_p0 = {"f1": 200.0, "D1": 0.05, "f2": 300.0, "D2": 0.005, "f3": 500.0, "D3": 0.001}
    full_bd = {"f1": (0.0, 2000.0), "D1": (0.01, 0.5), "f2": (0.0, 2000.0), "D2": (2e-3, 0.01), "f3": (0.0, 2000.0), "D3": (1e-5, 2e-3)}
    gen_curvefit("g9_tri_full_fixed_D1_D2_D3", TriExpModel(fit_reduced=False), ["f1", "D1", "f2", "D2", "f3", "D3"], full_tr,
                 full_p0, full_bd, 32, 96, [0.0, 0.01], rng, fixed={"D1": (0.03, 0.1), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)})
    gen_curvefit("g9_tri_fixed_4_of_5", TriExpModel(), ["f1", "D1", "f2", "D2", "D3"], tri_tr, {k: tri_p0[k] for k in ("f1", "D1", "f2", "D2", "D3")},
                 {k: tri_bd[k] for k in ("f1", "D1", "f2", "D2", "D3")}, 32, 64, [0.0, 0.01], rng,
                 fixed={"D1": (0.03, 0.1), "f2": (0.2, 0.4), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)})

    # --- NNLS spectrum post-processing on reference spectra (utility/spectrum.py:13-139)
    MAXP = 8
    cutoffs = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
    for src, regularized in (("g4_nnls_250_r2", True), ("g4_nnls_250_r1", True), ("g9_nnls_250_r0", False)):
        d = np.load(os.path.join(OUT, src + ".npz"))
        spec, bins = d["coefficients"], d["bins"]
        nv = spec.shape[0]
        for height in (0.1, 5.0):
            n_peaks = np.zeros(nv, dtype=np.int32)
            dv = np.full((nv, MAXP), np.nan)
            fv = np.full((nv, MAXP), np.nan)
            dc = np.full((nv, len(cutoffs)), np.nan)
            fc = np.full((nv, len(cutoffs)), np.nan)
            for i in range(nv):
                dd, ff = find_spectrum_peaks(spec[i], bins, height=height, regularized=regularized)
                assert len(dd) <= MAXP
                n_peaks[i] = len(dd)
                dv[i, :len(dd)] = dd
                fv[i, :len(dd)] = ff
                dc[i], fc[i] = apply_cutoffs(dd, ff, cutoffs)
            name = f"g9_spectrum_{src[3:]}_h{height:g}".replace(".", "p")
            np.savez_compressed(os.path.join(OUT, name + ".npz"), spectrum=spec, bins=bins, height=height,
                                regularized=np.array(regularized), n_peaks=n_peaks, d_values=dv, f_values=fv,
                                cutoffs=np.array(cutoffs), d_cut=dc, f_cut=fc, **_versions())
            print(f"{name}: peaks/voxel mean {n_peaks.mean():.2f} max {n_peaks.max()}")


def main_g10():
    """Fifth batch (round 2): the two remaining callers of the solver path -- SegmentedFitter (two chained pixel-wise fits,
    fitters/segmented.py:159-242) and SegmentationWiseFitter (one fit per label on the mean signal,
    fitters/segmentationwise.py:43-139)."""
    from pyneapple import BiExpModel, CurveFitSolver, MonoExpModel, SegmentationWiseFitter, SegmentedFitter

    rng = np.random.default_rng(SEED + 10)
    b = np.array([0, 10, 20, 40, 60, 80, 100, 150, 200, 300, 400, 500, 600, 700, 800, 1000], float)
    shape = (6, 5, 2)
    f1 = rng.uniform(0.1, 0.4, shape)
    D1 = rng.uniform(1e-2, 5e-2, shape)
    D2 = rng.uniform(5e-4, 2e-3, shape)
    S0 = rng.uniform(0.8, 1.2, shape)
    img = S0[..., None] * (f1[..., None] * np.exp(-b * D1[..., None]) + (1 - f1[..., None]) * np.exp(-b * D2[..., None]))
    img = img * (1.0 + 0.005 * rng.standard_normal(img.shape))
    seg = np.zeros(shape, dtype=int)
    seg[1:5, :, :] = 1
    seg[2, 2, 0] = 0
    mono_p0, mono_bd = {"S0": 1.0, "D": 0.001}, {"S0": (0.0, 5.0), "D": (0.0, 0.01)}
    bi_p0 = {"f1": 0.2, "D1": 0.01, "D2": 0.001, "S0": 1.0}
    bi_bd = {"f1": (0.0, 1.0), "D1": (0.003, 0.1), "D2": (0.0, 0.005), "S0": (0.1, 5.0)}

    def two_step(fixed, mapping, rng_b):
        s1 = CurveFitSolver(model=MonoExpModel(), max_iter=200, tol=1e-8, p0=mono_p0, bounds=mono_bd)
        s2 = CurveFitSolver(model=BiExpModel(fit_reduced=True, fit_s0=True), max_iter=500, tol=1e-8, p0=bi_p0, bounds=bi_bd)
        fit = SegmentedFitter(step1_solver=s1, step2_solver=s2, step1_bvalue_range=rng_b, fixed_from_step1=fixed,
                              param_mapping=mapping)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fit.fit(b, img, segmentation=seg)
        r, r1 = fit.results_, fit.step1_result_
        names = list(fit.fitted_params_)
        return dict(bvalues=b, image=img, segmentation=seg, pixel_indices=np.array(fit.pixel_indices),
                    param_names=np.array(names), params=np.stack([np.asarray(fit.fitted_params_[n]) for n in names]),
                    step1_names=np.array(list(fit.step1_params_)),
                    step1_params=np.stack([np.asarray(v) for v in fit.step1_params_.values()]),
                    step1_success=r1.success, success=r.success, r_squared=r.r_squared,
                    result_param_names=np.array(list(r.params)), covariance_shape=np.array(r.covariance.shape),
                    fixed_from_step1=np.array(fixed), mapping_src=np.array(list(mapping)),
                    mapping_dst=np.array(list(mapping.values())),
                    bvalue_lo=np.nan if rng_b is None or rng_b[0] is None else rng_b[0],
                    bvalue_hi=np.nan if rng_b is None or rng_b[1] is None else rng_b[1], **_versions())

    out = two_step(["D"], {"D": "D2"}, (200, None))
    np.savez_compressed(os.path.join(OUT, "g10_segmented_fix_D.npz"), **out)
    print(f"g10_segmented_fix_D: names={list(out['param_names'])} success={out['success'].mean():.3f}")
    out = two_step([], {}, None)
    np.savez_compressed(os.path.join(OUT, "g10_segmented_nofix.npz"), **out)
    print(f"g10_segmented_nofix: names={list(out['param_names'])} success={out['success'].mean():.3f}")

    # --- SegmentationWiseFitter: labels 0..3 (0 is fitted too: np.unique keeps it), with and without a fixed map
    lab = rng.integers(0, 4, shape)
    lab[0, 0, 0] = 7                      # a label with one voxel, and a gap in the label values
    for tag, fixed_maps in (("", None), ("_fixedmap", {"D1": D1})):
        s = CurveFitSolver(model=BiExpModel(fit_reduced=True, fit_s0=True), max_iter=500, tol=1e-8, p0=bi_p0, bounds=bi_bd)
        fit = SegmentationWiseFitter(s)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fit.fit(b, img, segmentation=lab, fixed_param_maps=fixed_maps)
        names = list(fit.fitted_params_)
        try:
            pred, predict_raises = fit.predict(b), False
        except KeyError:      # with a fixed map the reference's predict() indexes fitted_params_ with the fixed name
            pred, predict_raises = np.zeros(0), True
        extra = {"predict_raises": np.array(predict_raises)}
        if fixed_maps is not None:
            extra["fixed_D1"] = D1
        np.savez_compressed(os.path.join(OUT, f"g10_segmentationwise{tag}.npz"), bvalues=b, image=img, segmentation=lab,
                            segment_labels=fit.segment_labels, param_names=np.array(names),
                            params=np.stack([np.asarray(fit.fitted_params_[n]) for n in names]),
                            pixel_indices=np.array(fit.pixel_indices), predict=pred,
                            success=fit.results_.success, n_pixels=fit.results_.n_pixels,
                            p0_names=np.array(list(bi_p0)), p0_vals=np.array(list(bi_p0.values())),
                            lo_vals=np.array([v[0] for v in bi_bd.values()]), hi_vals=np.array([v[1] for v in bi_bd.values()]),
                            **extra, **_versions())
        print(f"g10_segmentationwise{tag}: labels={fit.segment_labels} names={names}")


def main_g11():
    """Sixth batch (round 4): spectra with more than 256 bins -- the reference takes any n_bins (models/nnls.py:37-77); the HIP
    library runs them on its eight-bins-per-lane instantiations (include/pnx.h) -- and the spectrum post-processing on them."""
    from pyneapple.utility.spectrum import apply_cutoffs, find_spectrum_peaks

    rng = np.random.default_rng(SEED + 11)
    gen_nnls("g11_nnls_300_r2", (0.0008, 0.5), 300, 2, 0.02, 32, 32, rng)
    gen_nnls("g11_nnls_512_r2", (0.0008, 0.5), 512, 2, 0.02, 32, 16, rng)
    gen_nnls("g11_nnls_350_r1", (0.0008, 0.5), 350, 1, 0.02, 24, 16, rng)
    gen_nnls("g11_nnls_300_r3", (0.0008, 0.5), 300, 3, 0.02, 32, 16, rng)
    gen_nnls("g11_nnls_400_r0", (0.0008, 0.5), 400, 0, 0.02, 32, 32, rng)
    gen_nnls("g11_nnls_300_r2_maxiter20", (0.0008, 0.5), 300, 2, 0.02, 32, 16, rng, max_iter=20)
    MAXP = 8
    cutoffs = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
    for src, regularized in (("g11_nnls_300_r2", True), ("g11_nnls_512_r2", True), ("g11_nnls_400_r0", False)):
        d = np.load(os.path.join(OUT, src + ".npz"))
        spec, bins = d["coefficients"], d["bins"]
        nv = spec.shape[0]
        for height in (0.1, 5.0):
            n_peaks = np.zeros(nv, dtype=np.int32)
            dv = np.full((nv, MAXP), np.nan)
            fv = np.full((nv, MAXP), np.nan)
            dc = np.full((nv, len(cutoffs)), np.nan)
            fc = np.full((nv, len(cutoffs)), np.nan)
            for i in range(nv):
                dd, ff = find_spectrum_peaks(spec[i], bins, height=height, regularized=regularized)
                assert len(dd) <= MAXP
                n_peaks[i] = len(dd)
                dv[i, :len(dd)] = dd
                fv[i, :len(dd)] = ff
                dc[i], fc[i] = apply_cutoffs(dd, ff, cutoffs)
            name = f"g11_spectrum_{src[4:]}_h{height:g}".replace(".", "p")
            np.savez_compressed(os.path.join(OUT, name + ".npz"), spectrum=spec, bins=bins, height=height,
                                regularized=np.array(regularized), n_peaks=n_peaks, d_values=dv, f_values=fv,
                                cutoffs=np.array(cutoffs), d_cut=dc, f_cut=fc, **_versions())
            print(f"{name}: peaks/voxel mean {n_peaks.mean():.2f} max {n_peaks.max()}")


def main_g12():
    """Seventh batch (round 5): curve_fit(sigma=..., absolute_sigma=...) -- the two keyword arguments the reference's docstring
    names and forwards (solvers/curvefit.py:33, 295-306): a 1-D sigma shared by the voxels (vector or scalar), with the
    finite-difference Jacobian (no fixed parameter) and with the analytic one (a fixed parameter: _wrap_jac scales its rows)."""
    from pyneapple import BiExpModel, TriExpModel

    rng = np.random.default_rng(SEED + 12)
    bi_p0 = {"f1": 0.2, "D1": 0.01, "D2": 0.001}
    bi_bd = {"f1": (0.0, 1.0), "D1": (1e-3, 0.1), "D2": (1e-5, 5e-3)}
    bi_tr = {"f1": (0.1, 0.4), "D1": (5e-3, 5e-2), "D2": (5e-4, 2e-3)}
    tri_p0 = {"f1": 0.2, "D1": 0.05, "f2": 0.3, "D2": 0.005, "D3": 0.001}
    tri_bd = {"f1": (0.0, 1.0), "D1": (0.01, 0.5), "f2": (0.0, 1.0), "D2": (2e-3, 0.01), "D3": (1e-5, 2e-3)}
    tri_tr = {"f1": (0.1, 0.3), "D1": (0.03, 0.1), "f2": (0.2, 0.4), "D2": (3e-3, 8e-3), "D3": (5e-4, 1.5e-3)}
    tri_names = ["f1", "D1", "f2", "D2", "D3"]
    sig24 = 0.004 * (1.0 + np.linspace(0.0, 1200.0, 24) / 300.0)  # noise floor that grows with the b-value
    sig32 = 0.004 * (1.0 + np.linspace(0.0, 1200.0, 32) / 300.0)
    gen_curvefit("g12_bi_sigma_rel", BiExpModel(), ["f1", "D1", "D2"], bi_tr, bi_p0, bi_bd, 24, 96, [0.0, 0.01, 0.03], rng,
                 solver_kwargs=dict(sigma=sig24, absolute_sigma=False))
    gen_curvefit("g12_bi_sigma_abs", BiExpModel(), ["f1", "D1", "D2"], bi_tr, bi_p0, bi_bd, 24, 96, [0.0, 0.01, 0.03], rng,
                 solver_kwargs=dict(sigma=sig24, absolute_sigma=True))
    gen_curvefit("g12_tri_sigma_abs", TriExpModel(), tri_names, tri_tr, tri_p0, tri_bd, 32, 128, [0.0, 0.01, 0.03], rng,
                 solver_kwargs=dict(sigma=sig32, absolute_sigma=True))
    gen_curvefit("g12_tri_sigma_rel", TriExpModel(), tri_names, tri_tr, tri_p0, tri_bd, 32, 128, [0.01, 0.03], rng,
                 solver_kwargs=dict(sigma=sig32, absolute_sigma=False))
    gen_curvefit("g12_tri_sigma_scalar", TriExpModel(), tri_names, tri_tr, tri_p0, tri_bd, 32, 64, [0.01], rng,
                 solver_kwargs=dict(sigma=0.05, absolute_sigma=True))
    gen_curvefit("g12_tri_abs_nosigma", TriExpModel(), tri_names, tri_tr, tri_p0, tri_bd, 32, 64, [0.01], rng,
                 solver_kwargs=dict(absolute_sigma=True))
    gen_curvefit("g12_bi_s0_fixed_D1_sigma_abs", BiExpModel(fit_s0=True), ["f1", "D1", "D2", "S0"], dict(bi_tr, S0=(500, 1500)),
                 dict(bi_p0, S0=1000.0), dict(bi_bd, S0=(1.0, 5000.0)), 24, 64, [0.0, 0.01], rng, fixed={"D1": (5e-3, 5e-2)},
                 solver_kwargs=dict(sigma=4.0 * (1.0 + np.linspace(0.0, 1200.0, 24) / 300.0), absolute_sigma=True))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g12":
        os.environ.setdefault("PYNEAPPLE_QUIET", "1")
        sys.dont_write_bytecode = True
        sys.path.insert(0, REF_SRC)
        _install_shims()
        os.makedirs(OUT, exist_ok=True)
        main_g12()
    elif len(sys.argv) > 1 and sys.argv[1] == "g7":  # only the second batch
        os.environ.setdefault("PYNEAPPLE_QUIET", "1")
        sys.dont_write_bytecode = True
        sys.path.insert(0, REF_SRC)
        _install_shims()
        main_g7()
    elif len(sys.argv) > 1 and sys.argv[1] == "g9":
        os.environ.setdefault("PYNEAPPLE_QUIET", "1")
        sys.dont_write_bytecode = True
        sys.path.insert(0, REF_SRC)
        _install_shims()
        os.makedirs(OUT, exist_ok=True)
        main_g9()
    elif len(sys.argv) > 1 and sys.argv[1] == "g10":
        os.environ.setdefault("PYNEAPPLE_QUIET", "1")
        sys.dont_write_bytecode = True
        sys.path.insert(0, REF_SRC)
        _install_shims()
        main_g10()
    elif len(sys.argv) > 1 and sys.argv[1] == "g11":
        os.environ.setdefault("PYNEAPPLE_QUIET", "1")
        sys.dont_write_bytecode = True
        sys.path.insert(0, REF_SRC)
        _install_shims()
        os.makedirs(OUT, exist_ok=True)
        main_g11()
    elif len(sys.argv) > 1 and sys.argv[1] == "g8":
        os.environ.setdefault("PYNEAPPLE_QUIET", "1")
        sys.dont_write_bytecode = True
        sys.path.insert(0, REF_SRC)
        _install_shims()
        main_g8()
    else:
        main()
        main_g7()
        main_g8()
        main_g9()
