"""ctypes binding of libpnx_hip.so (the C ABI in include/pnx.h).

The product path has no CPU fallback: if the HIP library is missing or no MI355X is visible the
functions below raise -- they never route through oracle/ or SciPy.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNX_LIB") or os.path.join(_HERE, "libpnx_hip.so")  # PNX_LIB: load another build (kernel-variant experiments)

PNX_MAX_PARAMS = 8
MEM_HOST, MEM_DEVICE = 0, 1
JAC_FD, JAC_ANALYTIC = 0, 1

# every symbol include/pnx.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "pnx_version", "pnx_device_count", "pnx_last_error", "pnx_model_n_params", "pnx_curvefit_batch_f64",
    "pnx_curvefit_batch_f32", "pnx_nnls_solve_f32",
    "pnx_nnls_plan_create", "pnx_nnls_plan_destroy", "pnx_nnls_solve_f64", "pnx_nnls_aty_f64", "pnx_nnls_batch_f64",
    "pnx_nnls_bins", "pnx_nnls_basis", "pnx_nnls_regularization_matrix", "pnx_sweep_f32", "pnx_sweep_f64",
    "pnx_resize2d_f64", "pnx_ideal_bounds_f64", "pnx_nnls_spectrum_peaks_f64", "pnx_nnls_solve_peaks_f64", "pnx_scatter_maps_f32",
    "pnx_mask_select_f64", "pnx_gather_rows_f64", "pnx_scatter_rows_t_f64", "pnx_row_ss_tot_f64", "pnx_upload", "pnx_download",
    "pnx_label_sums_f64", "pnx_release_staging", "pnx_queue_order_f64",
]


class PnxError(RuntimeError):
    """A call through the C ABI failed (code < 0)."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"pnx error {code}: {msg}")
        self.code = code


class CurvefitOpts(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("n_b", C.c_int32), ("n_free", C.c_int32), ("n_fixed", C.c_int32),
        ("free_idx", C.c_int32 * PNX_MAX_PARAMS), ("fixed_idx", C.c_int32 * PNX_MAX_PARAMS),
        ("per_voxel_p0_bounds", C.c_int32), ("fixed_per_voxel", C.c_int32), ("max_nfev", C.c_int32),
        ("jac_mode", C.c_int32), ("t1_mode", C.c_int32), ("absolute_sigma", C.c_int32), ("tr", C.c_double),
        ("tm", C.c_double), ("ftol", C.c_double), ("xtol", C.c_double), ("gtol", C.c_double),
        ("sigma", C.c_void_p),        # (n_b,) float64 on the host, or None
        ("queue_order", C.c_void_p),  # (n_vox,) int32 on the device, or None (device-mode calls only)
    ]


_lib = None


def _preload_hip_runtime():
    """Make libpnx_hip.so and PyTorch share ONE HIP runtime.

    PyTorch-ROCm wheels bundle their own `libamdhip64.so` (SONAME libamdhip64.so.7).  If libpnx_hip.so were
    loaded first it would pull in /opt/rocm's copy, torch would then load its bundled one, and the process would
    hold two HIP runtimes (streams / device pointers of one are meaningless to the other; the second one does
    not even see the GPU).  Loading torch's copy first registers the SONAME, so libpnx_hip.so binds to it.
    Without PyTorch installed the system ROCm runtime is used.
    """
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libpnx_hip.so (building it is __graft_entry__.build()'s / `python -m pyneapple_amd._build`'s job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -m pyneapple_amd._build). "
            "pyneapple_amd has no CPU fallback.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, dp, fp = C.c_void_p, C.c_void_p, C.c_void_p
    lib.pnx_version.restype = C.c_int
    lib.pnx_release_staging.argtypes = [C.c_int]
    lib.pnx_queue_order_f64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    lib.pnx_queue_order_f64.restype = C.c_int
    lib.pnx_release_staging.restype = C.c_int
    lib.pnx_device_count.restype = C.c_int
    lib.pnx_last_error.restype = C.c_int
    lib.pnx_last_error.argtypes = [C.c_char_p, C.c_int]
    lib.pnx_model_n_params.restype = C.c_int
    lib.pnx_model_n_params.argtypes = [C.c_int]
    lib.pnx_curvefit_batch_f64.restype = C.c_int
    lib.pnx_curvefit_batch_f64.argtypes = [C.POINTER(CurvefitOpts), C.c_int64, dp, dp, dp, dp, dp, dp, dp, dp, vp, vp,
                                           dp, C.c_int, C.c_int, vp]
    lib.pnx_curvefit_batch_f32.restype = C.c_int
    lib.pnx_curvefit_batch_f32.argtypes = lib.pnx_curvefit_batch_f64.argtypes  # all data pointers are void*
    lib.pnx_nnls_plan_create.restype = C.c_int
    lib.pnx_nnls_plan_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, dp, C.c_int, C.c_int]
    lib.pnx_nnls_plan_destroy.restype = C.c_int
    lib.pnx_nnls_plan_destroy.argtypes = [vp]
    lib.pnx_nnls_solve_f64.restype = C.c_int
    lib.pnx_nnls_solve_f64.argtypes = [vp, C.c_int64, dp, C.c_int, dp, dp, vp, vp, C.c_int, vp]
    lib.pnx_resize2d_f64.restype = C.c_int
    lib.pnx_resize2d_f64.argtypes = [vp, C.c_int, C.c_int, C.c_int64, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.pnx_ideal_bounds_f64.restype = C.c_int
    lib.pnx_ideal_bounds_f64.argtypes = [vp, C.c_int64, C.c_int, dp, dp, dp, vp, vp, vp, C.c_int, vp]
    lib.pnx_nnls_solve_f32.restype = C.c_int
    lib.pnx_nnls_solve_f32.argtypes = lib.pnx_nnls_solve_f64.argtypes
    lib.pnx_nnls_aty_f64.restype = C.c_int
    lib.pnx_nnls_aty_f64.argtypes = [vp, C.c_int64, vp, vp, vp]
    lib.pnx_nnls_batch_f64.restype = C.c_int
    lib.pnx_nnls_batch_f64.argtypes = [C.c_int64, C.c_int, C.c_int, dp, dp, C.c_int, dp, C.c_int, dp, dp, vp, vp,
                                       C.c_int]
    lib.pnx_nnls_bins.restype = C.c_int
    lib.pnx_nnls_bins.argtypes = [C.c_double, C.c_double, C.c_int, dp]
    lib.pnx_nnls_basis.restype = C.c_int
    lib.pnx_nnls_basis.argtypes = [C.c_int, dp, C.c_int, dp, dp, C.c_int]
    lib.pnx_nnls_regularization_matrix.restype = C.c_int
    lib.pnx_nnls_regularization_matrix.argtypes = [C.c_int, C.c_int, C.c_double, dp]
    lib.pnx_nnls_spectrum_peaks_f64.restype = C.c_int
    lib.pnx_nnls_spectrum_peaks_f64.argtypes = [C.c_int64, C.c_int, vp, dp, C.c_double, C.c_int, C.c_double, C.c_int, vp, vp, vp,
                                                C.c_int, dp, vp, vp, C.c_int, C.c_int, vp]
    lib.pnx_nnls_solve_peaks_f64.restype = C.c_int
    lib.pnx_nnls_solve_peaks_f64.argtypes = [vp, C.c_int64, vp, C.c_int, dp, C.c_double, C.c_int, C.c_double, C.c_int, vp, vp, vp,
                                             C.c_int, dp, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.pnx_scatter_maps_f32.restype = C.c_int
    lib.pnx_scatter_maps_f32.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int64, vp, C.c_int, C.c_int, vp]
    lib.pnx_mask_select_f64.restype = C.c_int
    lib.pnx_mask_select_f64.argtypes = [vp, C.c_int64, C.c_double, vp, C.POINTER(C.c_int64), C.c_int, vp]
    lib.pnx_gather_rows_f64.restype = C.c_int
    lib.pnx_gather_rows_f64.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, C.c_int, vp]
    lib.pnx_scatter_rows_t_f64.restype = C.c_int
    lib.pnx_scatter_rows_t_f64.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int64, vp, C.c_int, vp]
    lib.pnx_row_ss_tot_f64.restype = C.c_int
    lib.pnx_row_ss_tot_f64.argtypes = [vp, C.c_int64, C.c_int, vp, C.c_int, vp]
    lib.pnx_label_sums_f64.restype = C.c_int
    lib.pnx_label_sums_f64.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp]
    for f in (lib.pnx_upload, lib.pnx_download):
        f.restype = C.c_int
        f.argtypes = [vp, vp, C.c_int64, C.c_int, vp, C.c_int]
    lib.pnx_sweep_f32.restype = C.c_int
    lib.pnx_sweep_f32.argtypes = [C.c_int, C.c_int64, C.c_int, fp, fp, fp, fp, fp, fp, C.c_int, vp]
    lib.pnx_sweep_f64.restype = C.c_int
    lib.pnx_sweep_f64.argtypes = [C.c_int, C.c_int64, C.c_int, dp, dp, dp, dp, dp, dp, C.c_int, vp]
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().pnx_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int):
    if rc != 0:
        raise PnxError(rc, last_error())


def ptr(a):
    """Address of a C-contiguous numpy array / a torch tensor / None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    return a.data_ptr()  # torch tensor


def device_count() -> int:
    return int(load().pnx_device_count())


def require_device():
    if device_count() < 1:
        raise PnxError(-3, "no HIP device visible: pyneapple_amd needs an MI355X (there is no CPU fallback)")
