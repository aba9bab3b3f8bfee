"""The two remaining callers of the solver path, array level, for the HIP solvers.

`HipSegmentedFitter`       two chained pixel-wise fits: a simple model on a b-value subset, then the full model with some
                           of the first step's parameters fixed per voxel (reference: fitters/segmented.py:56-256).
`HipSegmentationWiseFitter` one fit per segmentation label on the label's mean signal (reference:
                           fitters/segmentationwise.py:19-179).

The first is two batched solver calls whose second takes the first's result columns as its per-voxel fixed columns (same
mask, same voxel order); the second reduces the image to a handful of rows before the solver sees it -- on the device
(pnx_label_sums_f64: per-label column sums in a fixed order), with the label bookkeeping as array operations instead of the
reference's per-voxel dictionary.  Both register under `pyneapple.fitters` like the pixel-wise fitter.
"""
from __future__ import annotations

import time
from dataclasses import replace

import numpy as np

from .fitters import HipFitterBase, HipPixelWiseFitter, _accepts_pixel_fixed_params, validate_segmentation


def _check_image(xdata, image):
    xdata = np.asarray(xdata, float)
    image = np.asarray(image)
    if xdata.ndim != 1:
        raise ValueError(f"xdata must be a 1D array, but got shape {xdata.shape}.")
    if image.shape[-1] != xdata.shape[0]:
        raise ValueError(f"ydata second dimension {image.shape[-1]} does not match xdata length {xdata.shape[0]}.")
    return xdata, image


class HipSegmentedFitter(HipFitterBase):
    """SegmentedFitter(step1_solver, step2_solver, step1_bvalue_range=None, fixed_from_step1=None, param_mapping=None).

    `step1_bvalue_range=(lo, hi)` keeps lo <= b <= hi for step 1 (None = open end); `fixed_from_step1` names step-1
    parameters that become per-voxel constants of step 2 under the names `param_mapping` gives them (identity when absent).
    After `fit`: `fitted_params_` = step-2 parameters followed by the fixed ones (fitters/segmented.py:219-226),
    `step1_params_`, `step1_result_`, `results_` (step-2 FitResult, `fit_time` = both steps)."""

    def __init__(self, step1_solver, step2_solver, step1_bvalue_range=None, fixed_from_step1=None, param_mapping=None,
                 **fitter_kwargs):
        super().__init__(solver=step2_solver, **fitter_kwargs)
        self.step1_solver = step1_solver
        self.step2_solver = step2_solver
        self.step1_bvalue_range = step1_bvalue_range
        self.fixed_from_step1 = list(fixed_from_step1 or [])
        self.param_mapping = dict(param_mapping or {})
        self.step1_params_: dict = {}
        self.step1_result_ = None
        have1 = list(step1_solver.model._all_param_names)
        have2 = list(step2_solver.model._all_param_names)
        for src in self.fixed_from_step1:
            if src not in have1:
                raise ValueError(f"fixed_from_step1 name {src!r} is not a parameter of the Step 1 model. Available: {have1}")
            dst = self.param_mapping.get(src, src)
            if dst not in have2:
                raise ValueError(f"Mapped parameter {dst!r} (from {src!r}) is not a parameter of the Step 2 model. "
                                 f"Available: {have2}")

    def _bvalue_mask(self, xdata):
        if self.step1_bvalue_range is None:
            return np.ones(xdata.size, dtype=bool)
        lo, hi = self.step1_bvalue_range
        keep = np.ones(xdata.size, dtype=bool)
        if lo is not None:
            keep &= xdata >= lo
        if hi is not None:
            keep &= xdata <= hi
        if not keep.any():
            raise ValueError(f"No b-values fall within the range {self.step1_bvalue_range}. Available b-values: {xdata}")
        if keep.sum() < 3:
            raise ValueError(f"Step 1 requires at least 3 b-values for fitting, but only {int(keep.sum())} fall within "
                             f"the range {self.step1_bvalue_range}.")
        return keep

    def fit(self, xdata, image, segmentation=None, **fit_kwargs):
        t0 = time.perf_counter()
        xdata, image = _check_image(xdata, image)
        keep = self._bvalue_mask(xdata)
        self.n_measurements = len(xdata)
        self.image_shape = image.shape
        # step 1: the simple model on the b-value subset
        one = HipPixelWiseFitter(self.step1_solver)
        one.fit(xdata[keep], image if keep.all() else image[..., keep], segmentation, **fit_kwargs)
        self.step1_params_ = dict(one.get_fitted_params())
        self.step1_result_ = one.results_
        # step 2: same mask, same voxel order -> the step-1 columns ARE the per-voxel fixed columns.  The pixel-wise fitter
        # takes volumes (its public contract), so hand it volumes only when something is fixed.
        fixed_maps = None
        if self.fixed_from_step1:
            spatial = image.shape[:-1]
            fixed_maps = {self.param_mapping.get(src, src):
                          self._reconstruct_volume(np.asarray(self.step1_params_[src], float), one.pixel_indices, spatial)
                          for src in self.fixed_from_step1}
        two = HipPixelWiseFitter(self.step2_solver)
        two.fit(xdata, image, segmentation, fixed_param_maps=fixed_maps, **fit_kwargs)
        self.pixel_indices = two.pixel_indices
        self._pixel_fixed = two._pixel_fixed
        self.fitted_params_ = dict(two.get_fitted_params())
        for src in self.fixed_from_step1:
            self.fitted_params_[self.param_mapping.get(src, src)] = self.step1_params_[src]
        self.results_ = None if two.results_ is None else replace(two.results_, fit_time=time.perf_counter() - t0)
        return self

    def _get_param_names(self):
        return list(self.solver.model._all_param_names)


def _label_positions(segmentation):
    """(labels in np.unique order, position of every voxel's label in that list) -- through a histogram when the labels are
    small non-negative integers (the usual ROI mask), through np.unique otherwise."""
    flat = np.asarray(segmentation).reshape(-1)
    if flat.size and np.issubdtype(flat.dtype, np.integer) and flat.min() >= 0 and flat.max() < (1 << 20):
        hist = np.bincount(flat)
        labels = np.nonzero(hist)[0].astype(flat.dtype)
        lut = np.full(hist.size, -1, dtype=np.int32)
        lut[labels] = np.arange(labels.size, dtype=np.int32)
        return labels, lut[flat]
    labels, inv = np.unique(flat, return_inverse=True)
    return labels, inv.reshape(-1).astype(np.int32)


def _label_means(rows, inv, n_seg, device):
    """Mean row per label position and the voxel counts.  On the GPU (pnx_label_sums_f64: fixed summation order) whenever the
    per-wave table fits its LDS; a label set too large for that is reduced with one numpy histogram per column."""
    from . import _lib, api

    c = rows.shape[1]
    if n_seg * (c + 1) <= api.LABEL_TABLE_MAX and _lib.device_count() > 0:
        sums, counts = api.label_sums(rows, inv, n_seg, device)
        counts = counts.astype(np.float64)
        return sums / counts[:, None], counts
    counts = np.bincount(inv, minlength=n_seg).astype(np.float64)
    means = np.empty((n_seg, c))
    for k in range(c):
        means[:, k] = np.bincount(inv, weights=rows[:, k], minlength=n_seg) / counts
    return means, counts


class HipSegmentationWiseFitter(HipFitterBase):
    """fit(xdata, image, segmentation, fixed_param_maps=None): one solver row per label (np.unique order, label 0 included),
    each the mean signal of the label's voxels; a fixed map contributes its mean over the label.  `fitted_params_[name]` has
    one entry per label (`segment_labels`), `pixel_indices` lists every voxel label by label, `segment_of_pixel` maps those
    rows to label positions; `predict` broadcasts the per-label prediction back to the voxels."""

    def __init__(self, solver, **fitter_kwargs):
        super().__init__(solver=solver, **fitter_kwargs)
        self.segment_labels = None
        self.segment_of_pixel = None
        self.pixel_indices = None

    @property
    def pixel_to_segment(self):
        """{(x, y, z): label position} as the reference keeps it (segmentationwise.py:108-121); built on demand."""
        if self.pixel_indices is None:
            return None
        return {tuple(int(v) for v in c): int(s) for c, s in zip(self.pixel_indices, self.segment_of_pixel)}

    def fit(self, xdata, image, segmentation=None, fixed_param_maps=None, **fit_kwargs):
        if segmentation is None:
            raise ValueError("segmentation is required for segmentation-wise fitting")
        t0 = time.perf_counter()
        xdata, image = _check_image(xdata, image)
        segmentation = validate_segmentation(segmentation, image.shape)
        spatial = image.shape[:-1]
        self.n_measurements = len(xdata)
        self.image_shape = image.shape
        labels, inv = _label_positions(segmentation)
        n_seg = labels.size
        flat = np.ascontiguousarray(image.reshape(-1, image.shape[-1]), dtype=np.float64)
        means, counts = _label_means(flat, inv, n_seg, getattr(self.solver, "device", 0))
        # voxels label by label, C order inside a label: a stable sort of small unsigned keys is a radix sort in numpy
        key = inv.astype(np.uint8 if n_seg <= 256 else np.uint16 if n_seg <= 65536 else np.int64)
        order = np.argsort(key, kind="stable")
        self.segment_labels = labels
        self.segment_of_pixel = inv[order]
        self.pixel_indices = np.stack(np.unravel_index(order, spatial), axis=1)
        pixel_fixed = None
        if fixed_param_maps is not None:
            names = list(self.solver.model._all_param_names)
            pixel_fixed = {}
            for name, vol in fixed_param_maps.items():
                if name not in names:
                    raise ValueError(f"Unknown fixed parameter {name!r}. Valid: {names}")
                vol = np.asarray(vol, float)
                if vol.shape != spatial:
                    raise ValueError(f"fixed_param_maps[{name!r}] must have shape {spatial}, got {vol.shape}.")
                pixel_fixed[name] = np.bincount(inv, weights=vol.reshape(-1), minlength=n_seg) / counts
        self._pixel_fixed = pixel_fixed
        if _accepts_pixel_fixed_params(self.solver):
            self.solver.fit(xdata, means, pixel_fixed_params=pixel_fixed, **fit_kwargs)
        elif pixel_fixed is not None:
            raise ValueError(f"{type(self.solver).__name__}.fit does not accept pixel_fixed_params; fixed_param_maps "
                             "cannot be honoured by this solver")
        else:
            self.solver.fit(xdata, means, **fit_kwargs)
        self.fitted_params_ = dict(self.solver.params_)
        self.segment_signals_ = means
        self.results_ = self._assemble(xdata, means, time.perf_counter() - t0)
        return self

    def predict(self, xdata, **predict_kwargs):
        """(X, Y, Z, len(xdata)): every voxel gets its label's prediction (segmentationwise.py:141-179)."""
        self._check_fitted()
        xdata = np.asarray(xdata, float)
        if xdata.ndim != 1:
            raise ValueError(f"Expected xdata to be 1D array, but got shape {xdata.shape}")
        per_label = self.predict_pixels(xdata)                       # (n_segments, N)
        out = np.zeros(tuple(self.image_shape[:-1]) + (xdata.size,))
        out[tuple(self.pixel_indices.T)] = per_label[self.segment_of_pixel]
        return out

    def parameter_maps(self, dtype=np.float32, on_device: bool = False) -> dict:
        """{name: (X, Y, Z) volume}: the label's value at each of its voxels."""
        if on_device:
            raise ValueError("segmentation-wise maps are a host-side broadcast of a handful of values")
        idx = tuple(self.pixel_indices.T)
        out = {}
        for name, v in self.fitted_params_.items():
            vol = np.zeros(self.image_shape[:-1], dtype=dtype)
            vol[idx] = np.asarray(v)[self.segment_of_pixel]
            out[name] = vol
        return out
