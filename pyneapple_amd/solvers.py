"""Pyneapple solver plugins backed by the MI355X HIP library (no CPU fallback).

`HipCurveFitSolver` mirrors `pyneapple.solvers.CurveFitSolver` (reference src/pyneapple/solvers/curvefit.py)
and `HipNNLSSolver` mirrors `pyneapple.solvers.NNLSSolver` (src/pyneapple/solvers/nnls_solver.py): same
constructor arguments, same `fit()` signature, same fitted state (`params_`, `diagnostics_`,
`pixel_results_`), same error behaviour -- only `_fit_data` runs all voxels at once through the C ABI.

Registered under the `pyneapple.solvers` entry-point group (pyproject.toml) as `hip_curvefit` / `hip_nnls`;
extra scalar keys of `[Fitting.solver]` arrive as keyword arguments (io/toml.py:328-338): `device`,
`n_gpus`, `jacobian`.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Any

import numpy as np

from . import api
from .sharding import pinned_to_gpu
from ._compat import HAVE_PYNEAPPLE, CurveFitBase, NNLSBase, PixelResultsView, RefBaseSolver

_CURVEFIT_MESSAGES = {
    0: "Optimal parameters not found: The maximum number of function evaluations is exceeded.",
    -1: "Each lower bound must be strictly less than each upper bound.",
    -2: "array must not contain infs or NaNs",
    -3: "Initial guess is outside of provided bounds",
    -4: "Residuals are not finite in the initial point.",
}
# scipy.optimize.curve_fit / least_squares arguments whose SciPy default is what the kernel computes
_CURVE_FIT_DEFAULTS = {"check_finite": True, "nan_policy": None, "loss": "linear",
                       "x_scale": 1.0, "f_scale": 1.0, "tr_solver": None, "full_output": False}
_NNLS_MESSAGES = {0: "Maximum number of iterations reached.", -2: "array must not contain infs or NaNs"}


def _validate_parameter_names(parameters: dict, param_names: list[str]):
    # utility/validation.py:153-173
    missing = set(param_names) - set(parameters.keys())
    if missing:
        raise ValueError(f"Missing bounds for required parameters: {missing}. Required: {param_names}")


def _validate_data_shapes(xdata: np.ndarray, ydata: np.ndarray):
    # utility/validation.py:84-112
    if xdata.ndim != 1:
        raise ValueError(f"xdata must be a 1D array, but got shape {xdata.shape}.")
    if ydata.ndim == 1:
        if ydata.shape[0] != xdata.shape[0]:
            raise ValueError(f"ydata length {ydata.shape[0]} does not match xdata length {xdata.shape[0]}.")
    elif ydata.ndim >= 2:
        if ydata.shape[-1] != xdata.shape[0]:
            raise ValueError(f"ydata second dimension {ydata.shape[-1]} does not match xdata length {xdata.shape[0]}.")
    else:
        raise ValueError(f"ydata must be 1D or 2D array, but got shape {ydata.shape}.")


def kernel_model_key(model) -> str:
    """Map a (reference or stand-in) parametric model object to the library's model id (T1 suffix ignored)."""
    names = list(model._all_param_names)
    if names and names[-1] == "T1":
        names = names[:-1]
    for key, ref in api.MODEL_PARAM_NAMES.items():
        if names == ref:
            return key
    raise NotImplementedError(f"model with parameters {list(model._all_param_names)} is not supported by the HIP backend")


def kernel_t1(model) -> dict:
    """T1 / STEAM settings of a model object (models/*.py: fit_t1, fit_t1_steam, repetition_time, mixing_time)."""
    steam = bool(getattr(model, "fit_t1_steam", False))
    t1 = bool(getattr(model, "fit_t1", False)) or steam
    if not t1:
        return dict(t1_mode=0, tr=0.0, tm=0.0)
    tr = getattr(model, "repetition_time", None)
    tm = getattr(model, "mixing_time", None)
    if tr is None:
        raise ValueError("repetition_time is required when fit_t1=True.")
    if steam and tm is None:
        raise ValueError("mixing_time is required when fit_t1_steam=True.")
    return dict(t1_mode=2 if steam else 1, tr=float(tr), tm=float(tm) if steam else 0.0)


def _io_dtype(name):
    name = str(np.dtype(name)) if not isinstance(name, str) else name
    if name not in ("float64", "float32"):
        raise ValueError("io_dtype must be 'float64' or 'float32'")
    return np.float32 if name == "float32" else np.float64


def _shard_device(first: int, k: int) -> int:
    """Device of shard k.  PNX_SHARE_DEVICE=1 (rehearsal on a one-GPU box, used by the GPU tests) maps every shard to the
    first device: the per-device host threads then run against one card, which is what the C ABI's thread-safety promise
    (include/pnx.h) has to carry anyway."""
    import os

    return first if os.environ.get("PNX_SHARE_DEVICE") == "1" else first + k


def _split(n: int, parts: int):
    edges = np.linspace(0, n, parts + 1).astype(np.int64)
    return [(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a]


class HipCurveFitSolver(CurveFitBase):
    """Batched bounded NLLS on MI355X with SciPy `curve_fit(method="trf")` semantics.

    Args mirror CurveFitSolver (curvefit.py:36-89).  `use_jacobian` is stored and, as in the reference (curvefit.py:289-293
    passes `jac_fn`, which is None without fixed parameters), does not select the Jacobian: finite differences without
    fixed parameters, the analytic model Jacobian with them.  `multi_threading` / `n_pools` describe the reference's
    joblib pool and have no meaning here.  Extra keyword arguments:
        xtol, gtol: least_squares tolerances (SciPy default 1e-8); any other curve_fit argument is refused.
        jacobian: "fd" (default; SciPy 2-point finite differences, what the reference uses when no parameter
            is fixed) or "analytic" (model Jacobian; always used when parameters are fixed, like the reference).
        device: first HIP device index (default 0).  n_gpus: number of devices to shard voxels over (default 1).
        io_dtype: "float64" (default, the reference's array types) or "float32": signals, start values, bounds and
            fixed maps cross the ABI as float32 and results come back as float32 (pnx_curvefit_batch_f32); the
            arithmetic stays fp64 -- for float32 images this is what the reference computes, without the float64
            host copy.
    """

    def __init__(self, model: Any, max_iter: int, tol: float, p0: dict[str, float],
                 bounds: dict[str, tuple[float, float]], verbose: bool = False, method: str = "trf",
                 multi_threading: bool = False, use_jacobian: bool = True, **solver_kwargs):
        self.jacobian_mode = str(solver_kwargs.pop("jacobian", "fd"))
        self.device = int(solver_kwargs.pop("device", 0))
        self.n_gpus = int(solver_kwargs.pop("n_gpus", 1))
        self.io_dtype = _io_dtype(solver_kwargs.pop("io_dtype", "float64"))
        if self.jacobian_mode not in ("fd", "analytic"):
            raise ValueError("jacobian must be 'fd' or 'analytic'")
        # The reference forwards every remaining key into scipy.optimize.curve_fit (curvefit.py:295-306).  The kernel
        # implements least_squares' xtol / gtol; anything else would change SciPy's result and there is no CPU path to
        # honour it, so it is refused instead of being dropped silently (keys at their SciPy default are accepted).
        self.xtol = float(solver_kwargs.pop("xtol", 1e-8))
        self.gtol = float(solver_kwargs.pop("gtol", 1e-8))
        # sigma / absolute_sigma: the two curve_fit arguments the reference's docstring names (curvefit.py:33).  A scalar or 1-D
        # sigma (one standard deviation per b-value, shared by the voxels) runs in the kernel; a 2-D sigma (covariance matrix of
        # the measurements: SciPy whitens with its Cholesky factor) is not implemented and refused here, not per voxel
        self.sigma = solver_kwargs.pop("sigma", None)
        self.absolute_sigma = bool(solver_kwargs.pop("absolute_sigma", False))
        if self.sigma is not None:
            self.sigma = np.asarray(self.sigma, float)
            if self.sigma.ndim > 1 and self.sigma.size != 1:
                raise ValueError("HipCurveFitSolver implements a scalar or 1-D sigma (one standard deviation per b-value); "
                                 "a 2-D sigma is not implemented")
            self.sigma = self.sigma.reshape(-1)
        for key, default in _CURVE_FIT_DEFAULTS.items():
            if key in solver_kwargs:
                v = solver_kwargs.pop(key)
                if not (v is default or v == default):
                    raise ValueError(f"HipCurveFitSolver does not implement curve_fit({key}={v!r}); only the SciPy "
                                     f"default ({default!r}) is supported")
        unknown = set(solver_kwargs) - {"n_pools"}
        if unknown:
            raise ValueError(f"HipCurveFitSolver got solver arguments it cannot honour: {sorted(unknown)} "
                             "(supported: sigma, absolute_sigma, xtol, gtol, jacobian, device, n_gpus, io_dtype, n_pools)")
        if method != "trf":
            raise ValueError(f"HipCurveFitSolver implements method='trf' only (got {method!r}); "
                             "use the reference CurveFitSolver for 'dogbox' / 'lm'.")
        if HAVE_PYNEAPPLE:
            super().__init__(model=model, max_iter=max_iter, tol=tol, p0=p0, bounds=bounds, verbose=verbose,
                             method=method, multi_threading=multi_threading, use_jacobian=use_jacobian,
                             **solver_kwargs)
        else:
            RefBaseSolver.__init__(self, model=model, max_iter=max_iter, tol=tol, verbose=verbose)
            self.method = method
            self.multi_threading = multi_threading
            self.use_jacobian = use_jacobian and hasattr(model, "jacobian")
            self.n_pools = solver_kwargs.pop("n_pools", None)
            self.solver_kwargs = solver_kwargs
            # curvefit.py:75-89
            if isinstance(p0[self.model.param_names[0]], (int, float, np.ndarray)):
                _validate_parameter_names(p0, self.model.param_names)
                self.p0 = p0
            else:
                raise ValueError("p0 must be a dict with parameter names as keys and initial values as values.")
            if isinstance(bounds[self.model.param_names[0]], tuple):
                _validate_parameter_names(bounds, self.model.param_names)
                self.bounds = bounds
            else:
                raise ValueError(
                    "bounds must be a dict with parameter names as keys and (lower, upper) tuples as values.")
        self._kernel_model = kernel_model_key(model)  # fail early and loudly for unsupported models
        self._kernel_t1 = kernel_t1(model)

    # ------------------------------------------------------------------ p0 / bounds (curvefit.py:319-392)
    def _prepare_p0_bounds(self, p0, bounds, n_pixels):
        names = self.model.param_names
        if p0 is not None:
            if isinstance(p0, dict):
                first = p0[names[0]]
                if isinstance(first, np.ndarray):
                    raise ValueError(
                        "p0 should either be a basic dict with scalar values or a single np.ndarray of initial values "
                        "for all parameters. Spatial non-uniform p0 should be handled separately before calling fit().")
                elif isinstance(first, (int, float)):
                    _validate_parameter_names(p0, names)
                    p0 = np.array([p0[n] for n in names], dtype=float)
                else:
                    raise ValueError("p0 dict values must be either all scalars or a single np.ndarray of initial "
                                     "values for all parameters.")
            elif isinstance(p0, np.ndarray):
                pass
            else:
                raise ValueError("p0 must be either a dict or a single np.ndarray.")
        else:
            _validate_parameter_names(self.p0, names)
            p0 = np.array([self.p0[n] for n in names], dtype=float)

        if bounds is not None:
            if isinstance(bounds, dict):
                if isinstance(bounds[names[0]], tuple):
                    _validate_parameter_names(bounds, names)
                    bounds = (np.array([bounds[n][0] for n in names], float),
                              np.array([bounds[n][1] for n in names], float))
                else:
                    raise ValueError("bounds dict values must be tuples of (lower, upper) for each parameter.")
            elif isinstance(bounds, tuple) and len(bounds) == 2:
                if not all(isinstance(b, np.ndarray) for b in bounds):
                    raise ValueError("bounds tuple must contain two np.ndarrays (lower, upper) of shape "
                                     "(n_params, n_pixels).")
            else:
                raise ValueError("bounds must be either a dict with parameter names as keys and (lower, upper) tuples "
                                 "as values, or a tuple of (lower, upper) np.ndarrays.")
        else:
            _validate_parameter_names(self.bounds, names)
            bounds = (np.array([self.bounds[n][0] for n in names], float),
                      np.array([self.bounds[n][1] for n in names], float))

        # the reference tiles everything to (n_params, n_pixels); shared vectors stay (n_params,) here
        # and the tiling happens implicitly on the device
        if p0.ndim == 2 and p0.shape[1] != n_pixels:
            raise ValueError(f"p0 shape {p0.shape} does not match number of voxels in ydata {n_pixels}.")
        lo, hi = bounds
        for arr in (lo, hi):
            if arr.ndim == 2 and arr.shape[1] != n_pixels:
                raise ValueError(
                    f"bounds shape {lo.shape} and {hi.shape} do not match number of voxels in ydata {n_pixels}.")
        per_voxel = p0.ndim == 2 or lo.ndim == 2 or hi.ndim == 2
        if per_voxel:
            def tile(a):
                a = np.asarray(a, float)
                return np.ascontiguousarray(a if a.ndim == 2 else np.repeat(a[:, None], n_pixels, axis=1))
            p0, lo, hi = tile(p0), tile(lo), tile(hi)
        return np.asarray(p0, float), np.asarray(lo, float), np.asarray(hi, float), per_voxel

    # ------------------------------------------------------------------ device memory kept between fits
    def close(self) -> None:
        """Give back what host-array fits keep on the device between calls: the staging slab of a streamed fit (the size of the
        volume: 2.1 GB for 256 x 256 x 64 x 32), its pinned block and its streams (`pnx_release_staging`).  The next fit builds
        them again (2-3 ms).  Called when the solver is garbage collected; safe to call twice."""
        if getattr(self, "_closed", True):
            return
        self._closed = True
        for dev in sorted({_shard_device(int(self.device), k) for k in range(max(1, int(getattr(self, "n_gpus", 1))))}):
            try:  # per device: one that is busy (another solver's call is using the set) or gone must not keep the others' slabs alive
                api.release_staging(dev)
            except Exception:
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    # ------------------------------------------------------------------ fit (curvefit.py:91-159)
    def fit(self, xdata: np.ndarray, ydata: np.ndarray, p0=None, bounds=None,
            pixel_fixed_params: dict[str, np.ndarray] | None = None, **fit_kwargs) -> "HipCurveFitSolver":
        self._closed = False  # from here on there may be something to give back
        # **fit_kwargs: accepted and unused, exactly like the reference (curvefit.py:91-159 never reads them)
        self._reset_state()
        xdata = np.asarray(xdata)
        ydata = np.asarray(ydata)
        _validate_data_shapes(xdata, ydata)
        n_pixels = ydata.shape[0] if ydata.ndim > 1 else 1
        if ydata.ndim == 1:
            ydata = ydata[np.newaxis, :]
        if ydata.ndim > 2:
            raise ValueError(f"ydata must be 1D or 2D array, but got shape {ydata.shape}.")
        p0_a, lo_a, hi_a, per_voxel = self._prepare_p0_bounds(p0, bounds, n_pixels)
        if self.sigma is not None and self.sigma.size not in (1, xdata.shape[0]):
            # curve_fit raises "`sigma` has incorrect shape." (scipy:_minpack_py.py:969), which the reference turns into a failed
            # fit of every voxel with one warning each (curvefit.py:308-317); a usage error is reported once here instead
            raise ValueError(f"`sigma` has incorrect shape: {self.sigma.size} values for {xdata.shape[0]} b-values.")

        all_names = list(self.model._all_param_names)
        # curvefit.py:274: per-pixel fixed values win, else the model's scalar fixed parameters
        fixed = pixel_fixed_params if pixel_fixed_params else (getattr(self.model, "fixed_params", None) or None)
        free_names = list(self.model.param_names)
        fixed_idx, fixed_vals, jac = [], None, self.jacobian_mode
        if fixed:
            free_idx = [i for i, n in enumerate(all_names) if n not in fixed]
            fixed_idx = [i for i, n in enumerate(all_names) if n in fixed]
            if len(free_idx) < p0_a.shape[0]:  # curvefit.py:285-288
                p0_a, lo_a, hi_a = p0_a[free_idx], lo_a[free_idx], hi_a[free_idx]
            if pixel_fixed_params:
                free_names = [n for n in free_names if n not in pixel_fixed_params]
                fv = []
                for i in fixed_idx:
                    v = np.asarray(fixed[all_names[i]], float)
                    if v.shape != (n_pixels,):
                        raise ValueError(f"pixel_fixed_params[{all_names[i]!r}] must have shape ({n_pixels},)")
                    fv.append(v)
                fixed_vals = np.ascontiguousarray(np.stack(fv, axis=0))
            else:
                fixed_vals = np.array([float(fixed[all_names[i]]) for i in fixed_idx])
            jac = "analytic"  # the reference passes model.jacobian_with_fixed here (curvefit.py:279-281)

        res = self._run(xdata, np.ascontiguousarray(ydata, self.io_dtype), p0_a, lo_a, hi_a, per_voxel, fixed_idx,
                        fixed_vals, jac)
        self._pack(res, n_pixels, free_names)
        return self

    def _pack(self, res, n_pixels, free_names):
        """Result dict of the batch call -> the reference's solver state (curvefit.py:150-159, 214-244)."""
        popt, pcov, status = res["popt"], res["pcov"], res["status"]
        success = status > 0
        params_rows = popt.T  # (n_px, n_free) strided view: a record's params is one column of popt, no 168 MB transpose
        self.pixel_results_ = PixelResultsView(
            params_rows, pcov, success,
            lambda i, st=status: None if st[i] > 0 else _CURVEFIT_MESSAGES.get(int(st[i]), "fit failed"))
        self.params_ = {name: [float(popt[i, 0])] if n_pixels == 1 else popt[i] for i, name in enumerate(free_names)}
        self.diagnostics_ = {"pcov": (pcov[0] if n_pixels == 1 else pcov) if pcov is not None else None, "n_pixels": n_pixels,
                             "status": status, "nfev": res["nfev"], "cost": res["cost"]}

    def _run(self, xdata, ydata, p0, lo, hi, per_voxel, fixed_idx, fixed_vals, jac):
        n_vox = ydata.shape[0]
        kw = dict(max_nfev=int(self.max_iter), ftol=float(self.tol), xtol=self.xtol, gtol=self.gtol, jac=jac,
                  fixed_idx=fixed_idx, sigma=self.sigma, absolute_sigma=self.absolute_sigma, **self._kernel_t1)
        n_dev = max(1, min(self.n_gpus, n_vox))
        if n_dev == 1:
            return api.curvefit(self._kernel_model, xdata, ydata, p0, lo, hi, fixed_vals=fixed_vals,
                                device=self.device, **kw)
        parts = _split(n_vox, n_dev)
        # the shards write their voxel-major results straight into row ranges of the full arrays (pcov is 200 B per triexp voxel:
        # concatenating it afterwards would cost more than the fit); popt is parameter-major and is joined afterwards
        n = p0.shape[0]
        dt = ydata.dtype
        full = {"pcov": np.empty((n_vox, n, n), dt), "status": np.empty(n_vox, np.int8), "nfev": np.empty(n_vox, np.int32),
                "cost": np.empty(n_vox, dt)}

        def work(k):
            a, b = parts[k]
            sl = slice(a, b)
            fv = fixed_vals
            if fv is not None and fv.ndim == 2:
                fv = np.ascontiguousarray(fv[:, sl])
            pv = (np.ascontiguousarray(p0[:, sl]), np.ascontiguousarray(lo[:, sl]),
                  np.ascontiguousarray(hi[:, sl])) if per_voxel else (p0, lo, hi)
            dev_k = _shard_device(self.device, k)
            with pinned_to_gpu(dev_k):  # this thread and the call's helper threads stay on the NUMA node of their GPU
                return api.curvefit(self._kernel_model, xdata, ydata[sl], *pv, fixed_vals=fv,
                                    device=dev_k, out={key: v[sl] for key, v in full.items()}, **kw)["popt"]

        with ThreadPoolExecutor(len(parts)) as ex:  # ctypes releases the GIL during the call
            popts = list(ex.map(work, range(len(parts))))
        return dict(full, popt=np.concatenate(popts, axis=1))


class HipNNLSSolver(NNLSBase):
    """Batched regularised NNLS on MI355X with `scipy.optimize.nnls` semantics (nnls_solver.py:18-210)."""

    def __init__(self, model: Any, reg_order: int = 0, mu: float = 0.02, max_iter: int = 250, tol: float = 1e-8,
                 verbose=False, multi_threading: bool = False, **solver_kwargs: Any) -> None:
        self.device = int(solver_kwargs.pop("device", 0))
        self.n_gpus = int(solver_kwargs.pop("n_gpus", 1))
        self.io_dtype = _io_dtype(solver_kwargs.pop("io_dtype", "float64"))  # "float32": fp32 signal in / spectra out
        if HAVE_PYNEAPPLE:
            super().__init__(model, reg_order=reg_order, mu=mu, max_iter=max_iter, tol=tol, verbose=verbose,
                             multi_threading=multi_threading, **solver_kwargs)
        else:
            RefBaseSolver.__init__(self, model, max_iter, tol, verbose)
            self.multi_threading = multi_threading
            self.n_pools = solver_kwargs.pop("n_pools", None)
            self.reg_order = reg_order
            self.mu = mu

    def get_regularization_matrix(self) -> np.ndarray:
        """(n_bins, n_bins) Tikhonov matrix, model_functions/nnls.py:46-85."""
        return api.nnls_regularization_matrix(self.model.n_bins, self.reg_order, self.mu)

    def _build_regularized_basis(self, xdata: np.ndarray) -> np.ndarray:
        return np.concatenate([self.model.get_basis(xdata), self.get_regularization_matrix()], axis=0)

    def _extend_signal(self, signal: np.ndarray) -> np.ndarray:
        """Kept for interface parity (nnls_solver.py:75-86); the HIP path never materialises it."""
        return np.concatenate((signal, np.zeros((signal.shape[0], self.model.n_bins))), axis=1)

    def fit_peaks(self, xdata: np.ndarray, signal: np.ndarray, height: float = 0.1, cutoffs=None, max_peaks: int = 8,
                  regularized: bool | None = None) -> "HipNNLSSolver":
        """Fit and reduce every spectrum to its peak table on the device (pnx_nnls_solve_peaks_f64): what a caller of the
        reference does with `fit` followed by a Python loop over `find_spectrum_peaks` / `apply_cutoffs`
        (utility/spectrum.py:51-215), without the (n_pixels, n_bins) coefficient array -- 8.4 GB for a 256 x 256 x 64
        volume -- crossing PCIe.  `regularized` defaults to reg_order != 0 (spectrum.py:68-71).
        params_: "d_values", "f_values" (n_pixels, max_peaks) NaN padded, "n_peaks", and with `cutoffs` "d_cut", "f_cut"."""
        self._reset_state()
        xdata = np.asarray(xdata, float)
        signal = np.asarray(signal, float)
        basis = np.asarray(self.model.get_basis(xdata), float)
        if signal.ndim == 1:
            signal = signal[np.newaxis, :]
        if signal.ndim != 2 or signal.shape[1] != basis.shape[0]:
            raise ValueError(f"signal shape {signal.shape} does not match the basis ({basis.shape[0]} measurements)")
        self.n_pixels = signal.shape[0]
        plan = api.NnlsPlan(basis, self.get_regularization_matrix(), self.device)
        try:
            res = plan.solve_peaks(np.ascontiguousarray(signal), np.asarray(self.model.bins, float),
                                   int(self.max_iter) if self.max_iter else 0, height=height,
                                   regularized=bool(self.reg_order) if regularized is None else regularized,
                                   max_peaks=max_peaks, cutoffs=cutoffs)
        finally:
            plan.close()
        for key in ("d_values", "f_values", "n_peaks", "d_cut", "f_cut"):
            if res[key] is not None:
                self.params_[key] = res[key]
        self.diagnostics_.update(residual=res["residual"], status=res["status"], iters=res["iters"])
        return self

    def fit(self, xdata: np.ndarray, signal: np.ndarray, pixel_fixed_params=None, **kwargs) -> "HipNNLSSolver":
        self._reset_state()
        xdata = np.asarray(xdata, float)
        signal = np.asarray(signal, self.io_dtype)
        basis = np.asarray(self.model.get_basis(xdata), float)
        reg = self.get_regularization_matrix()
        if signal.ndim == 1:
            signal = signal[np.newaxis, :]
        if signal.ndim != 2 or signal.shape[1] != basis.shape[0]:
            raise ValueError(f"signal shape {signal.shape} does not match the basis ({basis.shape[0]} measurements)")
        self.n_pixels = signal.shape[0]
        signal = np.ascontiguousarray(signal)
        n_dev = max(1, min(self.n_gpus, self.n_pixels))
        # SciPy: `if not maxiter: maxiter = 3*n` (scipy/optimize/_nnls.py:93-94)
        max_iter = int(self.max_iter) if self.max_iter else 0
        if n_dev == 1:
            res = api.nnls(basis, reg, signal, max_iter, self.device)
        else:
            parts = _split(self.n_pixels, n_dev)
            # every result is voxel-major: the shards write into row ranges of the full arrays (the spectra are 2 KB per voxel)
            res = {"coefficients": np.empty((self.n_pixels, basis.shape[1]), signal.dtype), "residual": np.empty(self.n_pixels, signal.dtype),
                   "status": np.empty(self.n_pixels, np.int8), "iters": np.empty(self.n_pixels, np.int32)}
            def work(k):
                dev_k = _shard_device(self.device, k)
                with pinned_to_gpu(dev_k):  # this thread and the call's helper threads stay on the NUMA node of their GPU
                    api.nnls(basis, reg, signal[parts[k][0]:parts[k][1]], max_iter, dev_k,
                             out={key: v[parts[k][0]:parts[k][1]] for key, v in res.items()})

            with ThreadPoolExecutor(len(parts)) as ex:
                list(ex.map(work, range(len(parts))))
        status = res["status"]
        self.pixel_results_ = PixelResultsView(
            res["coefficients"], None, status == 1,
            lambda i, st=status: None if st[i] == 1 else _NNLS_MESSAGES.get(int(st[i]), "fit failed"),
            residual=res["residual"])
        self.params_["coefficients"] = res["coefficients"]
        self.diagnostics_["residual"] = res["residual"]
        self.diagnostics_["status"] = status
        self.diagnostics_["iters"] = res["iters"]
        return self
