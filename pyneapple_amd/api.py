"""Array-level front-end of the C ABI: numpy (host staging) or torch-cuda tensors (HBM resident)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import CurvefitOpts, JAC_ANALYTIC, JAC_FD, MEM_DEVICE, MEM_HOST, check, load, ptr

MODEL_IDS = {"mono": 0, "bi_reduced": 1, "bi_s0": 2, "bi_full": 3, "tri_reduced": 4, "tri_s0": 5, "tri_full": 6}
MODEL_PARAM_NAMES = {
    "mono": ["S0", "D"],
    "bi_reduced": ["f1", "D1", "D2"],
    "bi_s0": ["f1", "D1", "D2", "S0"],
    "bi_full": ["f1", "D1", "f2", "D2"],
    "tri_reduced": ["f1", "D1", "f2", "D2", "D3"],
    "tri_s0": ["f1", "D1", "f2", "D2", "D3", "S0"],
    "tri_full": ["f1", "D1", "f2", "D2", "f3", "D3"],
}


def _is_torch(a):
    return a is not None and not isinstance(a, np.ndarray) and hasattr(a, "data_ptr")


def make_opts(model, n_b, fixed_idx=(), per_voxel=False, fixed_per_voxel=False, max_nfev=250, ftol=1e-8, xtol=1e-8,
              gtol=1e-8, jac="fd", t1_mode=0, tr=0.0, tm=0.0, sigma=None, absolute_sigma=False):
    """pnx_curvefit_opts.  sigma: None, a scalar or (n_b,) -- curve_fit's 1-D sigma shared by all voxels (a 2-D sigma is
    refused as SciPy refuses a wrong shape); absolute_sigma as in curve_fit.  The struct keeps the sigma array alive."""
    n_all = len(MODEL_PARAM_NAMES[model]) + (1 if t1_mode else 0)
    fixed_idx = [int(i) for i in fixed_idx]
    free_idx = [i for i in range(n_all) if i not in fixed_idx]
    o = CurvefitOpts()
    o.model = MODEL_IDS[model]
    o.n_b = int(n_b)
    o.n_free = len(free_idx)
    o.n_fixed = len(fixed_idx)
    for k, i in enumerate(free_idx):
        o.free_idx[k] = i
    for k, i in enumerate(fixed_idx):
        o.fixed_idx[k] = i
    o.per_voxel_p0_bounds = int(per_voxel)
    o.fixed_per_voxel = int(fixed_per_voxel)
    o.max_nfev = int(max_nfev)
    o.jac_mode = JAC_FD if jac == "fd" else JAC_ANALYTIC
    o.t1_mode, o.tr, o.tm = int(t1_mode), float(tr), float(tm)
    o.ftol, o.xtol, o.gtol = float(ftol), float(xtol), float(gtol)
    o.absolute_sigma = int(bool(absolute_sigma))
    if sigma is not None:
        sg = np.asarray(sigma, np.float64)
        if sg.size == 1:
            sg = np.full(int(n_b), float(sg.reshape(-1)[0]))
        elif sg.shape != (int(n_b),):
            raise ValueError("`sigma` has incorrect shape." if sg.ndim != 2 else
                             "a 2-D sigma (covariance matrix of the measurements) is not implemented by the HIP solver")
        o._sigma_keepalive = np.ascontiguousarray(sg)  # the struct holds a raw pointer
        o.sigma = o._sigma_keepalive.ctypes.data
    return o


def _out(out, key, shape, dtype):
    """A caller-provided result array (`out[key]`: C-contiguous, right shape and dtype -- e.g. a row range of a larger array
    that several device shards fill) or a fresh one."""
    a = None if out is None else out.get(key)
    if a is None:
        return np.empty(shape, dtype)
    if a.shape != tuple(np.atleast_1d(shape)) or a.dtype != np.dtype(dtype) or not a.flags.c_contiguous:
        raise ValueError(f"out[{key!r}] must be a C-contiguous {np.dtype(dtype).name} array of shape {shape}")
    return a


def curvefit(model, b, y, p0, lo, hi, *, fixed_idx=(), fixed_vals=None, max_nfev=250, ftol=1e-8, xtol=1e-8,
             gtol=1e-8, jac="fd", want_pcov=True, device=0, t1_mode=0, tr=0.0, tm=0.0, out=None, sigma=None,
             absolute_sigma=False):
    """Batched bounded NLLS on host (numpy) arrays.  Shapes as in include/pnx.h.

    A float32 signal array selects the fp32-storage entry point (pnx_curvefit_batch_f32: every data array float32 in
    and out, fp64 arithmetic on the device); anything else goes through pnx_curvefit_batch_f64.
    Returns dict(popt (n_free, n_vox), pcov (n_vox, n_free, n_free) | None, status int8, nfev int32, cost).
    `out`: optional dict of preallocated result arrays ("pcov", "status", "nfev", "cost") to write into.
    """
    _lib.require_device()
    dt = np.float32 if getattr(y, "dtype", None) == np.float32 else np.float64
    b = np.ascontiguousarray(b, dt)
    y = np.ascontiguousarray(np.atleast_2d(y), dt)
    n_vox, n_b = y.shape
    if b.shape != (n_b,):
        raise ValueError(f"b has shape {b.shape}, expected ({n_b},)")
    p0 = np.ascontiguousarray(p0, dt)
    lo = np.ascontiguousarray(lo, dt)
    hi = np.ascontiguousarray(hi, dt)
    per_voxel = p0.ndim == 2
    fv = None
    fpv = False
    if len(fixed_idx):
        fv = np.ascontiguousarray(fixed_vals, dt)
        fpv = fv.ndim == 2
    o = make_opts(model, n_b, fixed_idx, per_voxel, fpv, max_nfev, ftol, xtol, gtol, jac, t1_mode, tr, tm, sigma, absolute_sigma)
    n = o.n_free
    want = (n, n_vox) if per_voxel else (n,)
    if p0.shape != want or lo.shape != want or hi.shape != want:
        raise ValueError(f"p0/lo/hi must have shape {want}")
    if fv is not None and fv.shape != ((o.n_fixed, n_vox) if fpv else (o.n_fixed,)):
        raise ValueError("fixed_vals has the wrong shape")
    popt = np.empty((n, n_vox), dt)
    pcov = _out(out, "pcov", (n_vox, n, n), dt) if want_pcov else None
    status = _out(out, "status", (n_vox,), np.int8)
    nfev = _out(out, "nfev", (n_vox,), np.int32)
    cost = _out(out, "cost", (n_vox,), dt)
    fn = load().pnx_curvefit_batch_f32 if dt is np.float32 else load().pnx_curvefit_batch_f64
    check(fn(C.byref(o), n_vox, ptr(b), ptr(y), ptr(p0), ptr(lo), ptr(hi), ptr(fv),
                                        ptr(popt), ptr(pcov), ptr(status), ptr(nfev), ptr(cost), MEM_HOST, device, None))
    return dict(popt=popt, pcov=pcov, status=status, nfev=nfev, cost=cost)


def release_staging(device=0):
    """Free the device staging slab / pinned block / streams a streamed host-array curve fit keeps for the next call."""
    check(load().pnx_release_staging(int(device)))


def curvefit_device(opts, n_vox, b, y, p0, lo, hi, fixed, popt, pcov, status, nfev, cost, device, stream=None, order=None):
    """Enqueue a batched fit on HBM-resident torch tensors (asynchronous; caller synchronises).  float32 tensors
    select the fp32-storage entry point.  `order`: optional int32 device tensor, a permutation of the voxel indices in which
    the kernel's queue hands the voxels out (longest fits first hides the straggler tail; results do not depend on it)."""
    f32 = "float32" in str(getattr(y, "dtype", ""))
    b = np.ascontiguousarray(b, np.float32 if f32 else np.float64)
    if not getattr(opts, "per_voxel_p0_bounds", 0):
        p0, lo, hi = (np.ascontiguousarray(a, b.dtype) for a in (p0, lo, hi))
    if isinstance(fixed, np.ndarray):
        fixed = np.ascontiguousarray(fixed, b.dtype)
    fn = load().pnx_curvefit_batch_f32 if f32 else load().pnx_curvefit_batch_f64
    if order is not None and (tuple(order.shape) != (int(n_vox),) or "int32" not in str(order.dtype)):
        raise ValueError("order must be an int32 device tensor of n_vox entries")
    opts.queue_order = ptr(order)  # explicit per call (None clears what an earlier call with these opts set)
    try:
        check(fn(C.byref(opts), int(n_vox), ptr(b), ptr(y), ptr(p0), ptr(lo), ptr(hi),
                 ptr(fixed), ptr(popt), ptr(pcov), ptr(status), ptr(nfev), ptr(cost), MEM_DEVICE, int(device), stream))
    finally:
        opts.queue_order = None


class NnlsPlan:
    """Shared part of one NNLS fit: A = [basis; reg] uploaded and reduced to its Gram form once."""

    def __init__(self, basis, reg=None, device=0):
        _lib.require_device()
        basis = np.ascontiguousarray(basis, np.float64)
        if basis.ndim != 2:
            raise ValueError("basis must be 2-D (n_meas, n_bins)")
        self.n_meas, self.n_bins = basis.shape
        n_reg = 0
        if reg is not None:
            reg = np.ascontiguousarray(reg, np.float64)
            if reg.ndim != 2 or reg.shape[1] != self.n_bins:
                raise ValueError("reg must be (n_reg, n_bins)")
            n_reg = reg.shape[0]
        self.device = device
        self._h = C.c_void_p()
        check(load().pnx_nnls_plan_create(C.byref(self._h), self.n_meas, self.n_bins, ptr(basis), ptr(reg), n_reg,
                                          device))

    def solve(self, y, max_iter=250, out=None):
        """`out`: optional dict of preallocated result arrays ("coefficients", "residual", "status", "iters")."""
        dt = np.float32 if getattr(y, "dtype", None) == np.float32 else np.float64  # float32 in -> float32 out
        y = np.ascontiguousarray(np.atleast_2d(y), dt)
        n_vox = y.shape[0]
        if y.shape[1] != self.n_meas:
            raise ValueError(f"signal has {y.shape[1]} measurements, basis has {self.n_meas}")
        coeff = _out(out, "coefficients", (n_vox, self.n_bins), dt)
        rnorm = _out(out, "residual", (n_vox,), dt)
        status = _out(out, "status", (n_vox,), np.int8)
        iters = _out(out, "iters", (n_vox,), np.int32)
        fn = load().pnx_nnls_solve_f32 if dt is np.float32 else load().pnx_nnls_solve_f64
        check(fn(self._h, n_vox, ptr(y), int(max_iter), ptr(coeff), ptr(rnorm), ptr(status),
                                        ptr(iters), MEM_HOST, None))
        return dict(coefficients=coeff, residual=rnorm, status=status, iters=iters)

    def solve_device(self, n_vox, y, max_iter, coeff, rnorm, status, iters, stream=None):
        fn = load().pnx_nnls_solve_f32 if "float32" in str(getattr(y, "dtype", "")) else load().pnx_nnls_solve_f64
        check(fn(self._h, int(n_vox), ptr(y), int(max_iter), ptr(coeff), ptr(rnorm),
                                        ptr(status), ptr(iters), MEM_DEVICE, stream))

    def aty_device(self, n_vox, y, aty=None, stream=None):
        """Enqueue the MFMA Gram step alone: aty (n_vox, 256) = y @ basis (None: into the plan's own scratch)."""
        check(load().pnx_nnls_aty_f64(self._h, int(n_vox), ptr(y), ptr(aty), stream))

    def solve_peaks(self, y, bins, max_iter=250, height=0.1, regularized=False, rel_height=0.5, max_peaks=8, cutoffs=None):
        """Solve and reduce every spectrum to its peak table on the device (pnx_nnls_solve_peaks_f64): the (n_vox, n_bins)
        spectra never cross PCIe.  Returns dict(n_peaks, d_values, f_values (n_vox, max_peaks) NaN padded, d_cut, f_cut
        (n_vox, n_cut) or None, residual, status, iters)."""
        y = np.ascontiguousarray(np.atleast_2d(y), np.float64)
        n_vox = y.shape[0]
        if y.shape[1] != self.n_meas:
            raise ValueError(f"signal has {y.shape[1]} measurements, basis has {self.n_meas}")
        bins = np.ascontiguousarray(bins, np.float64)
        if bins.shape != (self.n_bins,):
            raise ValueError(f"bins has shape {bins.shape}, expected ({self.n_bins},)")
        cut = None if cutoffs is None else np.ascontiguousarray(cutoffs, np.float64).reshape(-1, 2)
        n_cut = 0 if cut is None else cut.shape[0]
        out = dict(n_peaks=np.empty(n_vox, np.int32), d_values=np.empty((n_vox, max_peaks)), f_values=np.empty((n_vox, max_peaks)),
                   d_cut=np.empty((n_vox, n_cut)) if n_cut else None, f_cut=np.empty((n_vox, n_cut)) if n_cut else None,
                   residual=np.empty(n_vox), status=np.empty(n_vox, np.int8), iters=np.empty(n_vox, np.int32))
        check(load().pnx_nnls_solve_peaks_f64(self._h, n_vox, ptr(y), int(max_iter), ptr(bins), float(height), int(bool(regularized)),
                                              float(rel_height), int(max_peaks), ptr(out["n_peaks"]), ptr(out["d_values"]),
                                              ptr(out["f_values"]), n_cut, ptr(cut), ptr(out["d_cut"]), ptr(out["f_cut"]),
                                              ptr(out["residual"]), ptr(out["status"]), ptr(out["iters"]), MEM_HOST, None))
        return out

    def close(self):
        if self._h:
            load().pnx_nnls_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nnls(basis, reg, y, max_iter=250, device=0, out=None):
    plan = NnlsPlan(basis, reg, device)
    try:
        return plan.solve(y, max_iter, out=out)
    finally:
        plan.close()


def nnls_bins(d_min, d_max, n_bins):
    out = np.empty(int(n_bins))
    check(load().pnx_nnls_bins(float(d_min), float(d_max), int(n_bins), ptr(out)))
    return out


def nnls_regularization_matrix(n_bins, order, mu=1.0):
    out = np.empty((int(n_bins), int(n_bins)))
    check(load().pnx_nnls_regularization_matrix(int(n_bins), int(order), float(mu), ptr(out)))
    return out


def nnls_basis(b, bins, device=0):
    _lib.require_device()
    b = np.ascontiguousarray(b, np.float64)
    bins = np.ascontiguousarray(bins, np.float64)
    out = np.empty((b.size, bins.size))
    check(load().pnx_nnls_basis(b.size, ptr(b), bins.size, ptr(bins), ptr(out), device))
    return out


SPECTRUM_MAX_PEAKS = 64  # kMaxPeaks of csrc/pnx_spectrum.hip (one peak per lane of the wave that owns the spectrum)


def spectrum_peaks(spectrum, bins, height=0.1, regularized=False, rel_height=0.5, max_peaks=8, cutoffs=None, device=0):
    """find_spectrum_peaks (+ apply_cutoffs) of utility/spectrum.py for every row of `spectrum` (n_vox, n_bins) at once, on
    the device.  numpy arrays in and out, or torch-cuda tensors for `spectrum` (then the outputs are torch tensors)."""
    _lib.require_device()
    tensor = _is_torch(spectrum)
    if not tensor:
        spectrum = np.ascontiguousarray(np.atleast_2d(spectrum), np.float64)
    n_vox, n_bins = spectrum.shape
    bins = np.ascontiguousarray(bins, np.float64)
    if bins.shape != (n_bins,):
        raise ValueError(f"bins has shape {bins.shape}, expected ({n_bins},)")
    cut = None if cutoffs is None else np.ascontiguousarray(cutoffs, np.float64).reshape(-1, 2)
    n_cut = 0 if cut is None else cut.shape[0]
    if tensor:
        import torch

        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=spectrum.device)
        out = dict(n_peaks=mk(n_vox, torch.int32), d_values=mk((n_vox, max_peaks), torch.float64),
                   f_values=mk((n_vox, max_peaks), torch.float64),
                   d_cut=mk((n_vox, n_cut), torch.float64) if n_cut else None, f_cut=mk((n_vox, n_cut), torch.float64) if n_cut else None)
        device = spectrum.device.index
        stream = torch.cuda.current_stream(spectrum.device).cuda_stream
    else:
        out = dict(n_peaks=np.empty(n_vox, np.int32), d_values=np.empty((n_vox, max_peaks)), f_values=np.empty((n_vox, max_peaks)),
                   d_cut=np.empty((n_vox, n_cut)) if n_cut else None, f_cut=np.empty((n_vox, n_cut)) if n_cut else None)
        stream = None
    check(load().pnx_nnls_spectrum_peaks_f64(n_vox, n_bins, ptr(spectrum), ptr(bins), float(height), int(bool(regularized)),
                                             float(rel_height), int(max_peaks), ptr(out["n_peaks"]), ptr(out["d_values"]),
                                             ptr(out["f_values"]), n_cut, ptr(cut), ptr(out["d_cut"]), ptr(out["f_cut"]),
                                             MEM_DEVICE if tensor else MEM_HOST, int(device), stream))
    if not tensor and n_vox:
        # rows the device table could not hold: more than 64 peaks, or more than 16 in a spectrum with a flat-topped rise
        over = out["n_peaks"] > SPECTRUM_MAX_PEAKS
        if max_peaks:
            over |= (out["n_peaks"] > 16) & np.isnan(out["d_values"][:, 0])
        if over.any():
            import warnings

            warnings.warn(f"{int(over.sum())} spectra have more than {SPECTRUM_MAX_PEAKS} peaks (the device table's size; 16 for a spectrum "
                          f"with a flat-topped rise): their peak and cutoff rows are NaN; n_peaks holds the count", RuntimeWarning, stacklevel=2)
    return out


def scatter_maps(values, pixel_indices, spatial_shape, device=0):
    """float32 volume `spatial_shape (+ values.shape[1:])`, zero outside the fitted voxels, values[i] at pixel_indices[i]
    (io/nifti.py:279-312 reconstruct_maps) -- laid out on the device (pnx_scatter_maps_f32)."""
    _lib.require_device()
    values = np.ascontiguousarray(values, np.float64)
    idx = np.asarray(pixel_indices)
    spatial_shape = tuple(int(s) for s in spatial_shape)
    lin = np.ascontiguousarray(np.ravel_multi_index(tuple(idx.T), spatial_shape), np.int64)
    n_px = values.shape[0]
    k = int(np.prod(values.shape[1:], dtype=np.int64)) if values.ndim > 1 else 1
    out = np.empty(spatial_shape + values.shape[1:], np.float32)
    check(load().pnx_scatter_maps_f32(ptr(values), ptr(lin), n_px, k, int(np.prod(spatial_shape)), ptr(out), MEM_HOST, int(device), None))
    return out


_RESIZE_METHODS = {"linear": 0, "cubic": 1}


def resize2d(array, target_shape, method="cubic", device=0):
    """Resize the first two axes of a float64 numpy array (X, Y, ...) on the device (pnx_resize2d_f64, host arrays)."""
    _lib.require_device()
    a = np.ascontiguousarray(array, np.float64)
    tx, ty = int(target_shape[0]), int(target_shape[1])
    c = int(np.prod(a.shape[2:], dtype=np.int64)) if a.ndim > 2 else 1
    out = np.empty((tx, ty) + a.shape[2:])
    check(load().pnx_resize2d_f64(ptr(a), a.shape[0], a.shape[1], c, ptr(out), tx, ty, _RESIZE_METHODS[method], MEM_HOST,
                                  int(device), None))
    return out


def resize2d_device(src, x, y, c, dst, tx, ty, method, device, stream=None):
    """Enqueue the resize on HBM-resident float64 tensors: src (x, y, c) -> dst (tx, ty, c)."""
    check(load().pnx_resize2d_f64(ptr(src), int(x), int(y), int(c), ptr(dst), int(tx), int(ty), _RESIZE_METHODS[method],
                                  MEM_DEVICE, int(device), stream))


def ideal_bounds_device(param_map, n_px, lo, hi, tol, p0, lower, upper, device, stream=None):
    """Enqueue p0 / lower / upper (n_params, n_px) of the next IDEAL level from a resized map (n_px, n_params)."""
    lo, hi, tol = (np.ascontiguousarray(a, np.float64) for a in (lo, hi, tol))
    check(load().pnx_ideal_bounds_f64(ptr(param_map), int(n_px), int(lo.size), ptr(lo), ptr(hi), ptr(tol), ptr(p0), ptr(lower),
                                      ptr(upper), int(device), stream))


def mask_select_device(mask, threshold, idx, device, stream=None) -> int:
    """idx[:k] = C-order indices of mask > threshold (float64 tensor, any shape); returns k (synchronises the stream)."""
    k = C.c_int64(0)
    check(load().pnx_mask_select_f64(ptr(mask), int(mask.numel()), float(threshold), ptr(idx), C.byref(k), int(device), stream))
    return int(k.value)


def gather_rows_device(src, c, idx, n_sel, dst, device, stream=None):
    check(load().pnx_gather_rows_f64(ptr(src), int(c), ptr(idx), int(n_sel), ptr(dst), int(device), stream))


def scatter_rows_t_device(popt, idx, n_sel, k, n_total, pmap, device, stream=None):
    check(load().pnx_scatter_rows_t_f64(ptr(popt), ptr(idx), int(n_sel), int(k), int(n_total), ptr(pmap), int(device), stream))


def queue_order_device(key, order, device, stream=None):
    """order (int32 tensor) = voxel indices by descending `key` (float64 tensor of predicted evaluation counts), ties in index
    order: what `curvefit_device(..., order=)` takes."""
    check(load().pnx_queue_order_f64(ptr(key), int(key.numel()), ptr(order), int(device), stream))
    return order


def row_ss_tot_device(y, n, c, out, device, stream=None):
    check(load().pnx_row_ss_tot_f64(ptr(y), int(n), int(c), ptr(out), int(device), stream))


LABEL_TABLE_MAX = 8192  # n_labels * (c + 1) entries of the kernel's 64 KB LDS table


def label_sums(rows, label_pos, n_labels, device=0):
    """(sums (n_labels, c), counts (n_labels,)) of the rows per label position (host arrays in and out, pnx_label_sums_f64)."""
    rows = np.ascontiguousarray(rows, np.float64)
    lab = np.ascontiguousarray(label_pos, np.int32)
    n, c = rows.shape
    if lab.shape != (n,):
        raise ValueError(f"label_pos must have shape ({n},), got {lab.shape}")
    sums = np.empty((int(n_labels), c))
    counts = np.empty(int(n_labels), dtype=np.int64)
    check(load().pnx_label_sums_f64(ptr(rows), ptr(lab), n, c, int(n_labels), ptr(sums), ptr(counts), MEM_HOST, int(device), None))
    return sums, counts


def upload(array, tensor, device, stream=None, threads=0):
    """Host numpy array -> device tensor of the same byte size (pnx_upload: a few threads, 32 MiB pieces)."""
    a = np.ascontiguousarray(array)
    nbytes = int(tensor.numel()) * int(tensor.element_size())
    if a.nbytes != nbytes:
        raise ValueError(f"upload: {a.nbytes} host bytes into a {nbytes}-byte tensor")
    check(load().pnx_upload(ptr(tensor), ptr(a), nbytes, int(device), stream, int(threads)))
    return tensor


def download(tensor, device, stream=None, threads=0, dtype=None):
    """Contiguous device tensor -> fresh numpy array of the same shape (pnx_download: destination pages touched ahead of the
    copy, a few threads).  dtype: numpy dtype of the result (default: from the tensor's element size and kind)."""
    import torch

    if not tensor.is_contiguous():
        tensor = tensor.contiguous()
    if dtype is None:
        dtype = {torch.float64: np.float64, torch.float32: np.float32, torch.int8: np.int8, torch.int32: np.int32,
                 torch.int64: np.int64, torch.uint8: np.uint8}[tensor.dtype]
    out = np.empty(tuple(tensor.shape), dtype=dtype)
    if out.nbytes:
        check(load().pnx_download(ptr(out), ptr(tensor), out.nbytes, int(device), stream, int(threads)))
    return out


def sweep_device(model, n_vox, b, y, params, cost, g, jtj, device, stream=None):
    """Enqueue one residual/Jacobian/normal-equation sweep on HBM-resident torch tensors (f32 or f64)."""
    import torch

    f64 = y.dtype == torch.float64
    b = np.ascontiguousarray(b, np.float64 if f64 else np.float32)
    fn = load().pnx_sweep_f64 if f64 else load().pnx_sweep_f32
    check(fn(MODEL_IDS[model], int(n_vox), int(b.size), ptr(b), ptr(y), ptr(params), ptr(cost), ptr(g), ptr(jtj),
             int(device), stream))
