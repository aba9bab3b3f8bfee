"""IDEAL multi-resolution driver on top of the batched HIP solver (BASELINE config 5).

Mirrors `pyneapple.fitters.IDEALFitter.fit` (reference src/pyneapple/fitters/ideal.py:95-259): for every row of
`dim_steps` the image, the segmentation and the previous level's parameter maps are resized to the level's
grid, p0 is clipped to the solver's global bounds, per-voxel bounds `p0 * (1 -/+ tol)` are derived, the mask
is thresholded (all-voxel fallback when it vanishes) and ONE batched `solver.fit` call with per-voxel
`(n_params, n_pixels)` p0 / bounds arrays runs the whole level on the GPU.

The reference resizes with `cv2.resize` per slice and channel (ideal.py:299-320); OpenCV is not a dependency here:
`resize2d` restates its INTER_LINEAR / INTER_CUBIC arithmetic (half-pixel centres, cubic a = -0.75, replicated
border, no anti-aliasing) as two small dense weight matrices applied to all slices and channels at once.
cv2 is absent from the build container, so this restatement is pinned by analytic cases only
(tests/test_ideal.py) -- "parity unpinned" against OpenCV itself.
"""
from __future__ import annotations

import time

import numpy as np

from .fitters import HipFitterBase, validate_segmentation

_INTERPOLATION_METHODS = ("linear", "cubic")


def _cubic_coeffs(x: np.ndarray) -> np.ndarray:
    """OpenCV interpolateCubic, A = -0.75: weights of the taps at offsets -1, 0, +1, +2."""
    A = -0.75
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    return np.stack([c0, c1, c2, 1.0 - c0 - c1 - c2], axis=-1)


def resize_weights(n_src: int, n_dst: int, method: str) -> np.ndarray:
    """(n_dst, n_src) matrix W with out = W @ in along one axis, OpenCV `resize` semantics."""
    if method not in _INTERPOLATION_METHODS:
        raise ValueError(f"Invalid interpolation method: {method}. Must be one of {_INTERPOLATION_METHODS}.")
    W = np.zeros((n_dst, n_src))
    scale = n_src / n_dst
    f = (np.arange(n_dst) + 0.5) * scale - 0.5
    s = np.floor(f).astype(int)
    fr = f - s
    rows = np.arange(n_dst)
    if method == "linear":
        lo = s < 0
        fr = np.where(lo, 0.0, fr)
        s = np.where(lo, 0, s)
        hi = s >= n_src - 1
        fr = np.where(hi, 0.0, fr)
        s = np.where(hi, n_src - 1, s)
        np.add.at(W, (rows, s), 1.0 - fr)
        np.add.at(W, (rows, np.minimum(s + 1, n_src - 1)), fr)
    else:
        c = _cubic_coeffs(fr)
        for k in range(4):
            np.add.at(W, (rows, np.clip(s - 1 + k, 0, n_src - 1)), c[:, k])  # replicated border
    return W


def resize2d(array: np.ndarray, target_shape, method: str = "cubic") -> np.ndarray:
    """Resize the first two axes of `array` (X, Y, ...) to `target_shape[:2]` (reference `_interpolate_array`)."""
    tx, ty = int(target_shape[0]), int(target_shape[1])
    if array.dtype.kind != "f":
        array = array.astype(np.float32)  # ideal.py:309-310
    if (tx, ty) == tuple(array.shape[:2]):
        return array  # cv2.resize to the same size returns the input values (last pyramid level: no 1 GB copy)
    Wx = resize_weights(array.shape[0], tx, method).astype(array.dtype)
    Wy = resize_weights(array.shape[1], ty, method).astype(array.dtype)
    out = np.tensordot(Wx, array, axes=(1, 0))            # (tx, Y, ...)
    out = np.tensordot(Wy, out, axes=(1, 1))              # (ty, tx, ...)
    return np.ascontiguousarray(np.swapaxes(out, 0, 1))   # (tx, ty, ...)


class HipIDEALFitter(HipFitterBase):
    """IDEAL pyramid around a `HipCurveFitSolver` (any solver with the CurveFitSolver interface works).

    Args follow IDEALFitter.__init__ (ideal.py:46-92): solver, dim_steps (n_steps, ideal_dims), step_tol
    {param: fraction}, ideal_dims, segmentation_threshold, interpolation_method ("linear" | "cubic").
    `dim_steps` / `step_tol` may be left out at construction -- `FittingConfig.build_fitter` builds a plugin fitter
    as `cls(solver=solver)` and forwards `[Fitting.ideal]` only to the built-in "ideal" type (io/toml.py:217-236) --
    and assigned as attributes before `fit`, which raises a ValueError naming them otherwise.
    After `fit`: `step_params` (one (X, Y, Z, n_params) map per level), `fitted_params_`, `pixel_indices`,
    `image_shape`, `results_` (FitResult of the last level), `fit_time`, `level_stats_` (device path: per level the mean
    cost at the level's start values -- one pass of the residual sweep kernel, pnx_sweep_f64 -- and at the estimates).
    """

    def __init__(self, solver, dim_steps=None, step_tol: dict | None = None, ideal_dims: int = 2,
                 segmentation_threshold: float = 0.2, interpolation_method: str = "cubic",
                 device_resident: bool | None = None, keep_level_inputs: bool = False, **fitter_kwargs):
        if interpolation_method not in _INTERPOLATION_METHODS:
            raise ValueError(
                f"Invalid interpolation method: {interpolation_method}. Must be one of {_INTERPOLATION_METHODS}.")
        super().__init__(solver=solver, **fitter_kwargs)
        self.dim_steps = None if dim_steps is None else np.asarray(dim_steps)
        self.step_tol = step_tol
        self.keep_level_inputs = keep_level_inputs  # device path: keep the last level's per-voxel p0 / bounds in HBM
        self.last_level_inputs_ = None
        self.level_stats_: list[dict] = []
        self.ideal_dims = ideal_dims
        self.segmentation_threshold = segmentation_threshold
        self.interpolation_method = interpolation_method
        # True: the whole pyramid stays in HBM (image uploaded once, resize / bounds / fits on the device, maps downloaded
        # per level); False: numpy resize + host-array solver calls; None: device resident when the solver is a
        # HipCurveFitSolver without fixed parameters and torch (device memory) is importable
        self.device_resident = device_resident
        self.step_params: list[np.ndarray] = []
        self.fit_time = None

    # ideal.py:261-297 --------------------------------------------------------------------------
    def _validate(self, xdata, image):
        if self.dim_steps is None or self.step_tol is None:
            raise ValueError(
                "HipIDEALFitter needs dim_steps (n_steps, ideal_dims) and step_tol {param: fraction}: pass them to the "
                "constructor or assign fitter.dim_steps / fitter.step_tol before fit() (a fitter built from a TOML config "
                "receives only the solver: io/toml.py forwards [Fitting.ideal] to the built-in 'ideal' type alone).")
        self.dim_steps = np.asarray(self.dim_steps)
        if np.ndim(xdata) != 1:
            raise ValueError(f"xdata must be a 1D array, but got shape {np.shape(xdata)}.")
        if image.shape[-1] != len(xdata):
            raise ValueError(f"ydata second dimension {image.shape[-1]} does not match xdata length {len(xdata)}.")
        if not isinstance(self.step_tol, dict):
            raise ValueError("step_tol must be a dict mapping parameter names to tolerance fractions, "
                             f"e.g. {{'S0': 0.5, 'D': 0.2}}. Got {type(self.step_tol).__name__}.")
        names = self.solver.model.param_names
        if set(names) - set(self.step_tol):
            raise ValueError(f"step_tol keys {set(self.step_tol)} do not match model parameter names {names}")
        ds = self.dim_steps
        if ds.ndim != 2:
            raise ValueError("dim_steps must be a 2D array of shape (n_steps, ideal_dims).")
        if ds.shape[1] != self.ideal_dims:
            raise ValueError(f"dim_steps must have {self.ideal_dims} columns corresponding to ideal_dims.")
        for i in range(ds.shape[0] - 1):
            if not np.all(ds[i + 1] > ds[i]):
                raise ValueError(f"dim_steps row {i + 1} must be greater than row {i} (monotonic increase).")
        if image.ndim == 3:
            if self.ideal_dims == 3:
                raise ValueError(f"Image dimension ({image.ndim}) not sufficient for 3D interpolation "
                                 f"(ideal_dims={self.ideal_dims})")
            image = np.expand_dims(image, axis=-2)
        elif image.ndim != 4:
            raise ValueError(f"Image Array needs to be 3 or 4 not {image.ndim}")
        if not np.allclose(ds[-1], image.shape[: self.ideal_dims]):
            raise ValueError("The last step in dim_steps must match the spatial dimensions of the image.")
        return image

    def fit(self, xdata, image, segmentation=None, **fit_kwargs):
        xdata = np.asarray(xdata, float)
        image = self._validate(xdata, np.asarray(image))
        self.n_measurements = len(xdata)
        self.image_shape = image.shape
        self.level_stats_ = []
        self.last_level_inputs_ = None
        t0 = time.perf_counter()
        if self.ideal_dims == 2 and image.ndim == 4:
            dim_steps = np.hstack([self.dim_steps, np.full((self.dim_steps.shape[0], 1), image.shape[2])])
        else:
            dim_steps = self.dim_steps
        if segmentation is None:
            segmentation = np.ones(image.shape[:3], dtype=int)
        segmentation = validate_segmentation(segmentation, image.shape)
        seg4 = segmentation[..., np.newaxis]
        names = list(self.solver.model.param_names)
        n_params = len(names)
        p0_vals = np.array([self.solver.p0[n] for n in names], float)
        lo_vals = np.array([self.solver.bounds[n][0] for n in names], float)
        hi_vals = np.array([self.solver.bounds[n][1] for n in names], float)
        tol_vals = np.array([self.step_tol[n] for n in names], float)
        if self._use_device_path():
            return self._fit_device(xdata, image, segmentation, dim_steps, names, p0_vals, lo_vals, hi_vals, tol_vals, t0,
                                    fit_kwargs)
        self.step_params = []
        self.stage_times_ = []  # per level: seconds spent resizing / masking, in solver.fit, in map assembly
        method = self.interpolation_method
        for step_index, step in enumerate(dim_steps):
            t_a = time.perf_counter()
            shape = tuple(int(s) for s in step)
            if step_index == 0:
                p0 = np.broadcast_to(p0_vals, (*shape, n_params)).copy()
                lower = np.broadcast_to(lo_vals, (*shape, n_params)).copy()
                upper = np.broadcast_to(hi_vals, (*shape, n_params)).copy()
            else:
                p0 = np.clip(resize2d(self.step_params[-1], shape, method), lo_vals, hi_vals)
                lower = np.clip(p0 * (1 - tol_vals), lo_vals, hi_vals)
                upper = np.clip(p0 * (1 + tol_vals), lo_vals, hi_vals)
            img = resize2d(image, shape, method)
            mask = resize2d(seg4, shape, method)[..., 0] > self.segmentation_threshold
            if not mask.any():  # ideal.py:199-209: ROI too sparse at this level -> fit everything
                mask = np.ones(shape, dtype=bool)
            pixels = np.ascontiguousarray(img[mask], dtype=np.float64)       # (n_px, N), C order of np.where
            idx = np.nonzero(mask)
            t_b = time.perf_counter()
            self.solver.fit(xdata, pixels, p0=np.ascontiguousarray(p0[mask].T),
                            bounds=(np.ascontiguousarray(lower[mask].T), np.ascontiguousarray(upper[mask].T)),
                            **fit_kwargs)
            t_c = time.perf_counter()
            param_map = np.zeros((*shape, n_params))
            for k, n in enumerate(names):
                param_map[idx[0], idx[1], idx[2], k] = np.atleast_1d(self.solver.params_[n])
            self.step_params.append(param_map)
            self.stage_times_.append((round(t_b - t_a, 3), round(t_c - t_b, 3), round(time.perf_counter() - t_c, 3)))
        self.pixel_indices = np.stack(idx, axis=1)  # (n_px, 3), C order of np.where
        self.fitted_params_ = dict(self.solver.params_)
        self.fit_time = time.perf_counter() - t0
        self.results_ = self._assemble(xdata, pixels, self.fit_time)  # ideal.py:256-259: FitResult of the final level
        return self

    # ------------------------------------------------------------------ device-resident pyramid
    def _use_device_path(self) -> bool:
        if self.device_resident is False:
            return False
        s = self.solver
        ok = hasattr(s, "_kernel_model") and hasattr(s, "_pack") and not (getattr(s.model, "fixed_params", None) or None) \
            and getattr(s, "n_gpus", 1) == 1 and getattr(s, "io_dtype", np.float64) is np.float64
        if ok:
            try:
                import torch  # noqa: F401  (device memory only)
            except Exception:
                ok = False
        if self.device_resident and not ok:
            raise ValueError("device_resident=True needs a single-GPU HipCurveFitSolver without fixed parameters and torch")
        return ok

    def _fit_device(self, xdata, image, segmentation, dim_steps, names, p0_vals, lo_vals, hi_vals, tol_vals, t0, fit_kwargs):
        """Same level loop as above with everything between the first upload and the per-level map download in HBM:
        pnx_resize2d_f64 (image, mask, previous maps), pnx_ideal_bounds_f64 (p0 / bounds of the level) and the
        device-pointer fit (pnx_curvefit_batch_f64, per-voxel p0 / bounds); mask thresholding + compaction, row gathers, the
        map scatter and SS_tot are HIP kernels too (pnx_mask_select_f64, pnx_gather_rows_f64, pnx_scatter_rows_t_f64,
        pnx_row_ss_tot_f64).  torch allocates, uploads and downloads; the level statistics use its reductions."""
        import torch

        from . import api

        s = self.solver
        dev_i = int(s.device)
        dev = torch.device("cuda", dev_i)
        torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        method = self.interpolation_method
        X, Y, Z, N = image.shape
        n = len(names)
        # the volume crosses PCIe once each way: threaded, page-pre-touching copies (pnx_upload / pnx_download) instead of
        # tensor.to() / tensor.cpu() -- 1.07 GB up and 1.05 GB down at C5
        img_d = api.upload(np.ascontiguousarray(image, np.float64), torch.empty(image.shape, dtype=torch.float64, device=dev),
                           dev_i, stream)
        # ideal.py:309-310: a mask that is not floating point is resized as float32 (and compared with the threshold in
        # float32, like the host path above); the resize kernel works in fp64, its result is rounded to float32 before the test
        seg_f32 = segmentation.dtype.kind != "f"
        seg_d = torch.from_numpy(np.ascontiguousarray(segmentation, np.float64)).to(dev)
        seg_thr = float(np.float32(self.segmentation_threshold)) if seg_f32 else float(self.segmentation_threshold)
        kw = dict(s._kernel_t1)
        self.step_params, self.stage_times_ = [], []
        prev = None  # (px, py, Z, n) device map of the previous level
        for step_index, step in enumerate(dim_steps):
            t_a = time.perf_counter()
            tx, ty = int(step[0]), int(step[1])
            shape = (tx, ty, Z)

            def resize(src, c):
                sx, sy = int(src.shape[0]), int(src.shape[1])
                if (sx, sy) == (tx, ty):
                    return src
                dst = torch.empty((tx, ty) + tuple(src.shape[2:]), dtype=torch.float64, device=dev)
                api.resize2d_device(src, sx, sy, c, dst, tx, ty, method, dev_i, stream)
                return dst

            img_l = resize(img_d, Z * N)
            n_all = tx * ty * Z
            # fitted voxels of the level: resized segmentation above the threshold (ideal.py:199), all voxels when the ROI
            # does not survive the down-sampling (ideal.py:199-209) -- thresholding + stable compaction on the device
            idx_buf = torch.empty(n_all, dtype=torch.int64, device=dev)
            seg_l = resize(seg_d, Z)
            if seg_f32 and seg_l is not seg_d:
                seg_l = seg_l.to(torch.float32).to(torch.float64)
            n_sel = api.mask_select_device(seg_l, seg_thr, idx_buf, dev_i, stream)
            all_px = n_sel == 0 or n_sel == n_all
            idx = None if all_px else idx_buf[:n_sel]
            n_px = n_all if all_px else n_sel
            if all_px:
                pixels = img_l.reshape(-1, N)
            else:
                pixels = torch.empty((n_px, N), dtype=torch.float64, device=dev)
                api.gather_rows_device(img_l, N, idx, n_px, pixels, dev_i, stream)
            per_voxel = prev is not None
            if per_voxel:
                m = resize(prev, Z * n).reshape(-1, n)
                if not all_px:
                    mg = torch.empty((n_px, n), dtype=torch.float64, device=dev)
                    api.gather_rows_device(m, n, idx, n_px, mg, dev_i, stream)
                    m = mg
                p0_d = torch.empty((n, n_px), dtype=torch.float64, device=dev)
                lo_d, hi_d = torch.empty_like(p0_d), torch.empty_like(p0_d)
                api.ideal_bounds_device(m, n_px, lo_vals, hi_vals, tol_vals, p0_d, lo_d, hi_d, dev_i, stream)
                p0_a, lo_a, hi_a = p0_d, lo_d, hi_d
            else:
                p0_a, lo_a, hi_a = p0_vals, lo_vals, hi_vals
            last = step_index == len(dim_steps) - 1
            popt = torch.empty((n, n_px), dtype=torch.float64, device=dev)
            status = torch.empty(n_px, dtype=torch.int8, device=dev)
            nfev = torch.empty(n_px, dtype=torch.int32, device=dev)
            cost = torch.empty(n_px, dtype=torch.float64, device=dev)
            pcov = torch.empty((n_px, n, n), dtype=torch.float64, device=dev) if last else None
            opts = api.make_opts(s._kernel_model, N, [], per_voxel, False, int(s.max_iter), float(s.tol),
                                 float(getattr(s, "xtol", 1e-8)), float(getattr(s, "gtol", 1e-8)),
                                 s.jacobian_mode, kw["t1_mode"], kw["tr"], kw["tm"])
            t_b = time.perf_counter()
            cost0 = None
            if per_voxel and kw["t1_mode"] == 0:
                # cost at the level's start values: one pass of the HBM-streaming residual sweep (pnx_sweep_f64) over
                # the same signal rows and the parameter-major p0 array the fit is about to start from
                cost0 = torch.empty(n_px, dtype=torch.float64, device=dev)
                g0 = torch.empty((n, n_px), dtype=torch.float64, device=dev)
                h0 = torch.empty((n * (n + 1) // 2, n_px), dtype=torch.float64, device=dev)
                api.sweep_device(s._kernel_model, n_px, xdata, pixels, p0_d, cost0, g0, h0, dev_i, stream)
                del g0, h0
            api.curvefit_device(opts, n_px, xdata, pixels, p0_a, lo_a, hi_a, None, popt, pcov, status, nfev, cost,
                                dev_i, stream)
            stats = {"shape": shape, "n_pixels": n_px, "converged_frac": float((status > 0).double().mean()),
                     "cost_mean": float(cost.mean())}
            if cost0 is not None:
                ok = status > 0
                stats["cost_p0_mean"] = float(cost0.mean())
                # TRF never accepts a step that raises the cost: the estimate is at least as good as the start value
                stats["not_worse_than_p0_frac"] = float((cost[ok] <= cost0[ok] * (1 + 1e-12)).double().mean()) if bool(ok.any()) else 1.0
            self.level_stats_.append(stats)
            if last and self.keep_level_inputs and per_voxel:
                self.last_level_inputs_ = {"p0": p0_d, "lo": lo_d, "hi": hi_d, "idx": idx, "cost_p0": cost0}
            # the level's parameter map (ideal.py:243-252): estimates scattered to their voxels, zero elsewhere
            pmap = torch.empty((tx, ty, Z, n), dtype=torch.float64, device=dev)
            api.scatter_rows_t_device(popt, idx, n_px, n, n_all, pmap, dev_i, stream)
            torch.cuda.synchronize(dev)
            t_c = time.perf_counter()
            self.step_params.append(api.download(pmap, dev_i, stream))
            prev = pmap
            self.stage_times_.append((round(t_b - t_a, 3), round(t_c - t_b, 3), round(time.perf_counter() - t_c, 3)))
        res = {"popt": api.download(popt, dev_i, stream), "pcov": api.download(pcov, dev_i, stream),
               "status": status.cpu().numpy(), "nfev": nfev.cpu().numpy(), "cost": api.download(cost, dev_i, stream)}
        ss_d = torch.empty(n_px, dtype=torch.float64, device=dev)
        api.row_ss_tot_device(pixels, n_px, N, ss_d, dev_i, stream)  # SS_tot of the fitted rows, reduced in HBM
        ss_tot = ss_d.cpu().numpy()
        s._reset_state()
        s._pack(res, n_px, list(names))  # the solver ends in the state solver.fit() of the last level leaves it in
        lin = np.arange(n_all) if all_px else idx.cpu().numpy()
        self.pixel_indices = np.stack(np.unravel_index(lin, shape), axis=1)  # (n_px, 3), C order of np.where
        self.fitted_params_ = dict(s.params_)
        self.fit_time = time.perf_counter() - t0
        host_pixels = image.reshape(-1, N) if all_px else image.reshape(-1, N)[lin]
        self.results_ = self._assemble(xdata, host_pixels, self.fit_time, ss_tot=ss_tot)
        return self
