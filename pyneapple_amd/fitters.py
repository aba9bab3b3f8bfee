"""Pixel-wise fitter with array-level result assembly.

Mirrors `pyneapple.fitters.PixelWiseFitter.fit` (reference src/pyneapple/fitters/pixelwise.py:32-104) and
`BaseFitter._assemble_fit_result / _compute_r_squared / _reconstruct_volume` (fitters/base.py:142-274), but without
their per-voxel Python loops: with 4.2 M voxels the reference's R^2 loop (`model.forward` per voxel) and its
list-of-tuples pixel index alone take longer than the GPU fit.  Here
  * R^2 of a curve fit comes from the kernel's own cost output (SS_res = 2 * cost), SS_tot is one numpy reduction;
  * R^2 of an NNLS fit is a chunked matrix product with the basis;
  * `pixel_indices` is an (n_pixels, 3) integer array (same C order as `np.where`).
Registered as `hip_pixelwise` under the `pyneapple.fitters` entry-point group.
"""
from __future__ import annotations

import time
from dataclasses import dataclass

import numpy as np

try:  # pragma: no cover - only where the reference is installed
    from pyneapple.result import FitResult  # type: ignore
except Exception:

    @dataclass
    class FitResult:  # field-for-field the reference's result.py:11-92
        params: dict
        success: np.ndarray
        n_iterations: np.ndarray | None = None
        messages: list | None = None
        covariance: np.ndarray | None = None
        residuals: np.ndarray | None = None
        r_squared: np.ndarray | None = None
        fit_time: float = 0.0
        image_shape: tuple | None = None
        pixel_indices: object = None
        n_pixels: int = 0
        solver_name: str = ""
        model_name: str = ""

        @property
        def n_converged(self) -> int:
            return int(np.sum(self.success))

        @property
        def convergence_rate(self) -> float:
            return 0.0 if self.n_pixels == 0 else float(self.n_converged / self.n_pixels)

        @property
        def mean_r_squared(self):
            if self.r_squared is None:
                return None
            if np.all(np.isnan(self.r_squared)):
                return float("nan")
            return float(np.nanmean(self.r_squared))


def _ss_tot(signals: np.ndarray, workers: int = 8) -> np.ndarray:
    """sum_i (y_i - mean(y))^2 per row; row blocks on a few threads (numpy releases the GIL) -- 134 M elements for the
    C3 volume take 0.35 s on one thread."""
    n = signals.shape[0]
    out = np.empty(n)

    def block(a, e):
        blk = signals[a:e]
        d = blk - blk.mean(axis=1, keepdims=True)
        np.einsum("ij,ij->i", d, d, out=out[a:e])

    if n < (1 << 16):
        block(0, n)
        return out
    from concurrent.futures import ThreadPoolExecutor

    edges = np.linspace(0, n, 4 * workers + 1).astype(np.int64)
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(lambda k: block(int(edges[k]), int(edges[k + 1])), range(len(edges) - 1)))
    return out


def r_squared_from_ss(ss_res: np.ndarray, signals: np.ndarray) -> np.ndarray:
    """1 - SS_res / SS_tot, NaN where the signal is constant (fitters/base.py:179-183)."""
    ss_tot = _ss_tot(signals)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(ss_tot > 0, 1.0 - ss_res / ss_tot, np.nan).astype(np.float64)


try:  # pragma: no cover - only where the reference is installed (tests/test_plugin_dropin.py runs this branch)
    from pyneapple.fitters.base import BaseFitter as _RefBaseFitter  # type: ignore
except Exception:
    _RefBaseFitter = None



def validate_segmentation(segmentation, image_shape):
    """utility/validation.py:133-149: one dimension less than the image; a trailing singleton axis (a NIfTI mask loaded as
    4-D) is squeezed away, any other mismatch is a ValueError with the reference's wording."""
    segmentation = np.asarray(segmentation)
    spatial = tuple(image_shape[:-1])
    if segmentation.ndim != len(image_shape) - 1:
        if segmentation.ndim and segmentation.shape[-1] == 1:
            segmentation = np.squeeze(segmentation, axis=-1)
        else:
            raise ValueError(f"Segmentation must have one less dimension than image shape {tuple(image_shape)}, "
                             f"but got shape {segmentation.shape}.")
    if segmentation.shape != spatial:
        raise ValueError(f"Segmentation shape {segmentation.shape} does not match expected image shape {spatial}.")
    return segmentation


class _StandInBaseFitter:
    """The state `BaseFitter.__init__` sets (fitters/base.py:39-62), for hosts without the reference installed."""

    def __init__(self, solver, verbose: bool = False, **fitter_kwargs):
        self.solver = solver
        self.verbose = verbose
        self.fitter_kwargs = fitter_kwargs
        self.results_ = None
        self.fitted_params_: dict = {}
        self.image_shape = None
        self.pixel_indices = None
        self.n_measurements = None


_FitterBase = _RefBaseFitter if _RefBaseFitter is not None else _StandInBaseFitter


def _solver_is_nnls(solver) -> bool:
    """The reference decides with isinstance(solver, NNLSSolver) (fitters/base.py:162-169); the stand-in tree has no such
    class, so fall back on what only distribution solvers have."""
    from ._compat import HAVE_PYNEAPPLE, RefNNLSSolver

    if HAVE_PYNEAPPLE and isinstance(solver, RefNNLSSolver):
        return True
    return hasattr(solver, "reg_order") and hasattr(solver.model, "get_basis")


def _accepts_pixel_fixed_params(solver) -> bool:
    import inspect

    try:
        sig = inspect.signature(solver.fit)
    except (TypeError, ValueError):
        return False
    return "pixel_fixed_params" in sig.parameters or any(p.kind is p.VAR_KEYWORD for p in sig.parameters.values())


class HipFitterBase(_FitterBase):
    """What the two plugin fitters share: array-level FitResult assembly, volume reconstruction, predict.

    Derives from the reference's `BaseFitter` when it is importable (so `isinstance(f, BaseFitter)` holds for a fitter
    built by `FittingConfig.build_fitter`, io/toml.py:236), from an attribute-identical stand-in otherwise."""

    def get_fitted_params(self):
        return self.fitted_params_

    def _check_fitted(self) -> None:
        if not self.fitted_params_:
            raise RuntimeError(f"{self.__class__.__name__} has not been fitted yet. "
                               f"Call fit() before predict() or get_fitted_params().")
        if self.pixel_indices is None:
            raise RuntimeError(f"{self.__class__.__name__} has not extracted pixel data yet. "
                               f"Call _extract_pixel_data() before predict() or get_fitted_params().")

    # fitters/base.py:188-274, array level ---------------------------------------------------------
    def _assemble(self, xdata, pixels, fit_time, ss_tot=None):
        """FitResult of the solver's current state.  `pixels` (n_px, N) are the fitted signals; `ss_tot` may be passed when
        it was already reduced elsewhere (the device-resident IDEAL pyramid reduces it in HBM)."""
        s = self.solver
        d = getattr(s, "diagnostics_", None) or {}
        n_px = pixels.shape[0]
        is_nnls = _solver_is_nnls(s)
        if d.get("status") is None:
            # a solver without the HIP solvers' array diagnostics (e.g. the reference's own CurveFitSolver): success from
            # its per-pixel records, SS_res from one vectorised forward (fitters/base.py:142-186 without the loop)
            prs = getattr(s, "pixel_results_", None)
            success = np.array([pr.success for pr in prs], dtype=bool) if prs is not None and len(prs) == n_px \
                else np.ones(n_px, dtype=bool)
            ss_res = np.sum((pixels - self.predict_pixels(xdata)) ** 2, axis=1)
            pc = d.get("pcov")
            covariance = None if pc is None else np.asarray(pc).reshape(n_px, *np.asarray(pc).shape[-2:])
            res = d.get("residual")
            residuals = None if res is None else np.atleast_1d(res).astype(np.float64)
        elif is_nnls:
            status = np.asarray(d["status"])
            success = status == 1
            coeffs = np.atleast_2d(s.params_["coefficients"])
            basis = np.asarray(s.model.get_basis(xdata))
            ss_res = np.empty(n_px)
            for a in range(0, n_px, 1 << 18):
                e = min(n_px, a + (1 << 18))
                ss_res[a:e] = np.sum((pixels[a:e] - coeffs[a:e] @ basis.T) ** 2, axis=1)
            covariance, residuals = None, np.atleast_1d(d["residual"]).astype(np.float64)
        else:
            status = np.asarray(d["status"])
            success = status > 0
            ss_res = 2.0 * np.atleast_1d(d["cost"]).astype(np.float64)  # cost = 0.5 * sum(res^2) at the returned x
            bad = ~success
            if bad.any():  # failed voxels return p0: evaluate the model there (a handful of voxels)
                names = list(self.fitted_params_.keys())
                arr = np.stack([np.atleast_1d(self.fitted_params_[n]) for n in names])[:, bad]
                fixed = getattr(s.model, "fixed_params", None) or {}
                all_names = list(s.model._all_param_names)
                for k, i in enumerate(np.nonzero(bad)[0]):
                    vals = dict(zip(names, arr[:, k]))
                    vals.update(fixed)
                    vals.update({n: np.asarray(v)[i] for n, v in (self._pixel_fixed or {}).items()})
                    try:
                        pred = s.model.forward(xdata, *[vals[n] for n in all_names])
                        ss_res[i] = np.sum((pixels[i] - pred) ** 2)
                    except Exception:
                        ss_res[i] = np.nan
            covariance = np.asarray(d["pcov"]).reshape(n_px, *np.asarray(d["pcov"]).shape[-2:])
            residuals = None
        msgs = None
        if not success.all() and getattr(s, "pixel_results_", None) is not None and len(s.pixel_results_) == n_px:
            view = s.pixel_results_
            msgs = [None] * n_px
            for i in np.nonzero(~success)[0]:
                msgs[i] = view[int(i)].message
        if ss_tot is None:
            ss_tot = _ss_tot(pixels)
        with np.errstate(divide="ignore", invalid="ignore"):
            r2 = np.where(ss_tot > 0, 1.0 - ss_res / ss_tot, np.nan).astype(np.float64)
        return FitResult(params=dict(s.params_), success=success, n_iterations=None, messages=msgs,
                         covariance=covariance, residuals=residuals, r_squared=r2,
                         fit_time=fit_time, image_shape=self.image_shape, pixel_indices=self.pixel_indices,
                         n_pixels=n_px, solver_name=type(s).__name__, model_name=type(s.model).__name__)

    _pixel_fixed = None  # per-pixel fixed parameter columns of the last fit (HipPixelWiseFitter sets it)

    def _reconstruct_volume(self, flat_values, pixel_indices, spatial_shape):
        """fitters/base.py:310-331 with an (n_px, 3) index array instead of a list of tuples."""
        vol = np.zeros(spatial_shape, dtype=np.float64)
        idx = np.asarray(pixel_indices)
        vol[tuple(idx.T)] = flat_values
        return vol

    def parameter_maps(self, dtype=np.float32, on_device: bool = False) -> dict:
        """{name: (X, Y, Z[, n_bins]) volume}, zeros outside the mask (io/nifti.py:279-312 reconstruct_maps).
        on_device: narrow to float32 and lay the volumes out on the GPU (pnx_scatter_maps_f32) -- half the bytes of the
        float64 values go back over PCIe, and no host-side zero fill + fancy-index scatter."""
        if on_device:
            from . import api

            if np.dtype(dtype) != np.float32:
                raise ValueError("on_device parameter maps are float32 (the NIfTI writer's type, io/nifti.py:306)")
            dev = getattr(self.solver, "device", 0)
            return {name: api.scatter_maps(np.asarray(v), self.pixel_indices, self.image_shape[:-1], dev)
                    for name, v in self.fitted_params_.items()}
        idx = tuple(np.asarray(self.pixel_indices).T)
        spatial = self.image_shape[:-1]
        out = {}
        for name, v in self.fitted_params_.items():
            v = np.asarray(v)
            vol = np.zeros(spatial + v.shape[1:], dtype=dtype)
            vol[idx] = v
            out[name] = vol
        return out

    def predict_pixels(self, xdata):
        """Model prediction per fitted pixel, (n_pixels, N) -- one vectorised forward, no per-voxel loop."""
        xdata = np.asarray(xdata, float)
        s = self.solver
        if "coefficients" in self.fitted_params_:
            return np.atleast_2d(self.fitted_params_["coefficients"]) @ np.asarray(s.model.get_basis(xdata)).T
        names = list(s.model.param_names)
        fixed = dict(getattr(s.model, "fixed_params", None) or {})
        cols = {n: np.atleast_1d(self.fitted_params_[n]) for n in names if n in self.fitted_params_}
        for n, v in (self._pixel_fixed or {}).items():
            cols[n] = np.atleast_1d(v)
        n_px = len(next(iter(cols.values())))
        args = [cols[n][:, None] if n in cols else np.full((n_px, 1), float(fixed[n])) for n in s.model._all_param_names]
        return s.model.forward(xdata[None, :], *args)

    def predict(self, xdata, **predict_kwargs):
        """(X, Y, Z, len(xdata)) volume of model predictions, zeros outside the fitted voxels (fitters/base.py:90-131)."""
        self._check_fitted()
        xdata = np.asarray(xdata, float)
        if xdata.ndim != 1:
            raise ValueError(f"xdata must be a 1D array of independent variable values, got shape {xdata.shape}.")
        pred = self.predict_pixels(xdata)
        return self._reconstruct_volume(pred, self.pixel_indices, tuple(self.image_shape[:-1]) + (xdata.size,))


class HipPixelWiseFitter(HipFitterBase):
    """fit(xdata, image (X,Y,Z,N), segmentation=None, fixed_param_maps=None) -> self; results in `results_`."""

    def __init__(self, solver, **fitter_kwargs):
        super().__init__(solver=solver, **fitter_kwargs)

    def fit(self, xdata, image, segmentation=None, fixed_param_maps=None, **fit_kwargs):
        t0 = time.perf_counter()
        xdata = np.asarray(xdata, float)
        image = np.asarray(image)
        if xdata.ndim != 1:
            raise ValueError(f"xdata must be a 1D array, but got shape {xdata.shape}.")
        if image.shape[-1] != xdata.shape[0]:
            raise ValueError(f"ydata second dimension {image.shape[-1]} does not match xdata length {xdata.shape[0]}.")
        self.n_measurements = len(xdata)
        self.image_shape = image.shape
        spatial = image.shape[:-1]
        if segmentation is None:
            mask = np.ones(spatial, dtype=bool)
        else:
            segmentation = validate_segmentation(segmentation, image.shape)
            mask = segmentation != 0
        if segmentation is None:  # every voxel: a view, not a 1 GB fancy-index copy
            pixels = np.ascontiguousarray(image.reshape(-1, image.shape[-1]), dtype=np.float64)
        else:
            pixels = np.ascontiguousarray(image[mask], dtype=np.float64)  # (n_px, N), C order of np.where
        self.pixel_indices = np.argwhere(mask)
        pixel_fixed = None
        if fixed_param_maps is not None:
            names = list(self.solver.model._all_param_names)
            pixel_fixed = {}
            for name, vol in fixed_param_maps.items():
                if name not in names:
                    raise ValueError(f"Unknown fixed parameter {name!r}. Valid: {names}")
                vol = np.asarray(vol)
                if vol.shape != spatial:
                    raise ValueError(f"fixed_param_maps[{name!r}] must have shape {spatial}, got {vol.shape}.")
                pixel_fixed[name] = np.ascontiguousarray(vol[mask], dtype=np.float64)
        self._pixel_fixed = pixel_fixed
        # pixelwise.py:91-96 always forwards pixel_fixed_params; a solver that cannot take it must not silently lose the maps
        if _accepts_pixel_fixed_params(self.solver):
            self.solver.fit(xdata, pixels, pixel_fixed_params=pixel_fixed, **fit_kwargs)
        elif pixel_fixed is not None:
            raise ValueError(f"{type(self.solver).__name__}.fit does not accept pixel_fixed_params; fixed_param_maps "
                             "cannot be honoured by this solver")
        else:
            self.solver.fit(xdata, pixels, **fit_kwargs)
        self.fitted_params_ = dict(self.solver.params_)
        self.results_ = self._assemble(xdata, pixels, time.perf_counter() - t0)
        return self
