/*
 * pnx.h -- C ABI of the MI355X (gfx950) batched per-voxel fitting library (libpnx_hip.so).
 *
 * This is the drop-in boundary for the hot path of darksim33/Pyneapple's pixelwise fitter:
 * everything a solver plugin needs in order to replace
 *     CurveFitSolver._fit_data / _fit_single_pixel   (src/pyneapple/solvers/curvefit.py:171-317)
 *     NNLSSolver._fit_data / _fit_single_pixel       (src/pyneapple/solvers/nnls_solver.py:129-210)
 * i.e. the per-voxel calls into scipy.optimize.curve_fit (method="trf") and scipy.optimize.nnls.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch/numpy types.
 *   - Every function returns 0 on success or a negative pnx_error; the message of the last
 *     error of the calling thread is available through pnx_last_error().
 *   - `mem` says where the *per-voxel* arrays live: PNX_MEM_HOST (pageable host memory; the library
 *     moves the volume over PCIe on its own streams and helper threads while the kernels run -- a curve fit
 *     as ONE kernel that waits at an upload watermark and whose finished voxels are
 *     downloaded while it is still fitting, everything else as chunks through a ring of device slots --
 *     and returns when every result is in place; `stream` is synchronised on entry) or PNX_MEM_DEVICE (pointers are HBM addresses on `device`; the call only
 *     enqueues work on `stream` and returns; the caller synchronises).  Small shared inputs
 *     (b-values, shared p0/bounds, basis, regulariser) are ALWAYS host pointers.
 *   - Thread safety: every entry point may be called from several host threads (e.g. one per
 *     device).  An NNLS plan owns device scratch that serves one solve at a time: host-mode solves
 *     on one plan serialise internally, device-mode solves on one plan must be enqueued on ONE
 *     stream (or be ordered by the caller).
 *   - Throughput: a curve-fit batch ends in a tail -- a few voxels that need ten times the average number of evaluations
 *     keep their lanes busy after the work queue is empty (about 6 of 37 ms for 4 M triexp voxels).  Device-mode callers
 *     with independent batches should enqueue them on two or more streams: the next batch fills the idle SIMDs (the host
 *     mode pays one tail per call and downloads during it).
 *   - The caller allocates all outputs.  The library owns only device scratch.
 *   - Per-voxel numerical failure never produces an error code: it is reported in `status[]`
 *     with the reference's sentinel outputs (curvefit.py:308-317, nnls_solver.py:201-210).
 *   - Environment.  A production process reads five variables, none of which changes a result or selects a kernel:
 *     PNX_STREAM_CACHE_MB (staging slab a host-array curve fit keeps per device, default 8192), PNX_COPY_PIECE_MB (32) and
 *     PNX_COPY_THREADS (4) of pnx_upload / pnx_download, PNX_HOST_TOUCHERS and PNX_STREAM_OUT_THREADS (helper threads of a
 *     host-array call).  Everything else the sources know -- A/B switches between kernels (PNX_NNLS_NO_BLK, PNX_BLK_NO_WIDE,
 *     PNX_NNLS_NO_QR, PNX_NNLS_NO_MFMA, PNX_BLK_ROUTE_PERMILLE ...), chunk sizes of the host pipelines, trace output and the
 *     test hook PNX_NNLS_TEST_REJECT (forces rejected candidate columns in the NNLS block kernel) -- is read ONLY when the
 *     process was started with PNX_ENABLE_TEST_HOOKS=1 (looked at once, at the first query); the test-suite and the
 *     profiling scripts set it, bench.py and the plugin never do.
 */
#ifndef PNX_H
#define PNX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the functions declared here are its only exports. */
#if defined(__GNUC__) || defined(__clang__)
#define PNX_API __attribute__((visibility("default")))
#else
#define PNX_API
#endif

#define PNX_VERSION_MAJOR 0
#define PNX_VERSION_MINOR 2
#define PNX_MAX_PARAMS 8  /* max model parameters (free + fixed) */
#define PNX_MAX_BVALUES 128

/* Forward models: parameter order is the reference's `_all_param_names`
 * (models/monoexp.py:91-101, biexp.py:103-118, triexp.py:103-121; formulas
 * model_functions/multiexp.py:35-202). */
typedef enum pnx_model {
    PNX_MODEL_MONO = 0,        /* [S0, D]                      S0*exp(-b D)                          */
    PNX_MODEL_BI_REDUCED = 1,  /* [f1, D1, D2]                 f1 e1 + (1-f1) e2                      */
    PNX_MODEL_BI_S0 = 2,       /* [f1, D1, D2, S0]             S0 (f1 e1 + (1-f1) e2)                 */
    PNX_MODEL_BI_FULL = 3,     /* [f1, D1, f2, D2]             f1 e1 + f2 e2                          */
    PNX_MODEL_TRI_REDUCED = 4, /* [f1, D1, f2, D2, D3]         f1 e1 + f2 e2 + (1-f1-f2) e3           */
    PNX_MODEL_TRI_S0 = 5,      /* [f1, D1, f2, D2, D3, S0]     S0 (...)                               */
    PNX_MODEL_TRI_FULL = 6     /* [f1, D1, f2, D2, f3, D3]     f1 e1 + f2 e2 + f3 e3                  */
} pnx_model;

typedef enum pnx_jac_mode {
    PNX_JAC_FD = 0,       /* SciPy '2-point' finite differences -- what the reference uses when no
                             parameter is fixed (curvefit.py:289-293 -> _minpack_py.py:1000-1001) */
    PNX_JAC_ANALYTIC = 1  /* model.jacobian() -- what the reference uses with fixed parameters
                             (curvefit.py:274-288) */
} pnx_jac_mode;

typedef enum pnx_mem { PNX_MEM_HOST = 0, PNX_MEM_DEVICE = 1 } pnx_mem;

typedef enum pnx_error {
    PNX_OK = 0,
    PNX_ERR_INVALID = -1,     /* bad argument (shape, enum, NULL) -- the reference raises ValueError */
    PNX_ERR_UNSUPPORTED = -2, /* combination not built into this library */
    PNX_ERR_NO_DEVICE = -3,   /* no usable MI355X / HIP runtime error at start-up */
    PNX_ERR_HIP = -4,         /* HIP runtime error during the call */
    PNX_ERR_NOMEM = -5
} pnx_error;

/* Per-voxel status written by pnx_curvefit_batch_f64 (int8):
 *    1..4  SciPy termination status (gtol / ftol / xtol / ftol+xtol): success
 *    0     max_nfev reached            -> reference: RuntimeError -> params=p0, cov=NaN, success=False
 *   -1     some lb >= ub               -> reference: ValueError   -> same sentinel
 *   -2     non-finite signal           -> reference: ValueError (asarray_chkfinite) -> same sentinel
 *   -3     p0 outside its bounds       -> reference: ValueError   -> same sentinel
 *   -4     non-finite residual at p0   -> reference: ValueError   -> same sentinel          */
/* Per-voxel status written by pnx_nnls_batch_f64 (int8):
 *    1 converged; 0 iteration limit (`iteration == max_iter`) -> zeros, ||y||; -2 non-finite input */

typedef struct pnx_curvefit_opts {
    int32_t model;                       /* pnx_model */
    int32_t n_b;                         /* number of b-values (<= PNX_MAX_BVALUES) */
    int32_t n_free;                      /* free parameters being fitted */
    int32_t n_fixed;                     /* fixed parameters; n_free + n_fixed == model's parameter count */
    int32_t free_idx[PNX_MAX_PARAMS];    /* ascending positions of the free parameters in the model's order
                                            (models/base.py:167-173 _free_indices) */
    int32_t fixed_idx[PNX_MAX_PARAMS];   /* positions of the fixed parameters */
    int32_t per_voxel_p0_bounds;         /* 0: p0/lo/hi are (n_free,) shared; 1: (n_free, n_vox) parameter-major
                                            (utility/validation.py:177-203,248-297; fitters/ideal.py:240-242) */
    int32_t fixed_per_voxel;             /* 0: fixed is (n_fixed,); 1: (n_fixed, n_vox)  (curvefit.py:161-169) */
    int32_t max_nfev;                    /* reference's max_iter -> SciPy max_nfev (curvefit.py:303) */
    int32_t jac_mode;                    /* pnx_jac_mode */
    int32_t t1_mode;                     /* 0 none; 1 T1: S*(1-exp(-TR/T1)); 2 STEAM: additionally *exp(-TM/T1).
                                            T1 is then one more model parameter, appended last
                                            (model_functions/multiexp.py:210-241, each model class appends "T1" to its names) */
    int32_t absolute_sigma;              /* curve_fit(absolute_sigma=True): pcov is returned unscaled instead of
                                            multiplied by 2 cost / (n_b - n_free) (scipy:_minpack_py.py:1057-1063) */
    double tr;                           /* repetition time, same unit as T1 */
    double tm;                           /* mixing time (STEAM) */
    double ftol;                         /* reference's tol (curvefit.py:304) */
    double xtol;                         /* SciPy default 1e-8 */
    double gtol;                         /* SciPy default 1e-8 */
    const double *sigma;                 /* NULL, or (n_b,) doubles on the HOST (also for the _f32 entry point): curve_fit's 1-D
                                            `sigma`, shared by all voxels -- the two keyword arguments the reference's solver
                                            names and forwards (curvefit.py:33, 295-306).  Residual and Jacobian rows are
                                            multiplied by 1 / sigma_i (_minpack_py.py:958-960, 545-562); cost is the weighted
                                            0.5 * sum.  A scalar sigma is n_b equal entries; a 2-D sigma (covariance matrix of
                                            the measurements) is not implemented */
    const int32_t *queue_order;          /* NULL, or n_vox int32 on the DEVICE, a permutation of 0 .. n_vox - 1 (PNX_MEM_DEVICE
                                            calls only, PNX_ERR_INVALID otherwise): the kernel's k-th queue pull fits voxel
                                            queue_order[k].  Results do not depend on the order (a voxel's arithmetic is its own);
                                            the run time does: a pass ends in the longest fits that were started last, so a caller
                                            with a predictor of the evaluation counts -- the nfev map of a previous fit of the
                                            same volume (refits, the second step of the reference's SegmentedFitter) or of the
                                            previous level of the IDEAL pyramid (fitters/ideal.py:150-190) -- passes them longest
                                            first (pnx_queue_order_f64 builds the permutation).  The array must stay valid until
                                            the call's work on `stream` has finished */
} pnx_curvefit_opts;

PNX_API int pnx_version(void);
/* Number of visible HIP devices (0 if none); never fails. */
PNX_API int pnx_device_count(void);
/* Copies the calling thread's last error message (NUL-terminated) into buf; returns its length. */
PNX_API int pnx_last_error(char *buf, int n);
/* Number of parameters of a model without the optional T1 parameter, or PNX_ERR_INVALID. */
PNX_API int pnx_model_n_params(int model);

/*
 * Device memory the library keeps between calls, per device: (i) PNX_MEM_HOST curve-fit calls keep one staging slab (the
 * volume's signal and results, up to PNX_STREAM_CACHE_MB, default 8192), a pinned control block and their streams for the next
 * call; (ii) the NNLS plans of the reference's regularised configurations share ONE set of slabs for the kernels behind their
 * first pass (1.55 GB, see pnx_nnls_plan_create), which stays allocated when the last plan is destroyed.
 * This frees both -- (i) unless such a call is running on the device (PNX_ERR_INVALID), (ii) unless a plan still exists.
 */
PNX_API int pnx_release_staging(int device);

/* order (n int32, device) = the voxel indices by descending key (n doubles, device), equal keys in index order: the queue order
 * for pnx_curvefit_opts::queue_order from a predicted evaluation count per voxel.  Synchronises `stream`. */
PNX_API int pnx_queue_order_f64(const double *key_device, int64_t n, int32_t *order_device, int device, void *stream);

/*
 * Batched bounded non-linear least squares, fp64, results matching SciPy 1.15 curve_fit(method="trf").
 * Replaces: CurveFitSolver._fit_data (curvefit.py:171-244) for all voxels at once.
 *
 *   b      (n_b,)                     host
 *   y      (n_vox, n_b) row-major     host|device    -- fitters/pixelwise.py:91-96 hands exactly this
 *   p0,lo,hi  (n_free,) host  or (n_free, n_vox) host|device when opts->per_voxel_p0_bounds
 *   fixed  (n_fixed,) host or (n_fixed, n_vox) host|device when opts->fixed_per_voxel; NULL if n_fixed==0
 *   popt   (n_free, n_vox)            out, host|device  (curvefit.py:235)
 *   pcov   (n_vox, n_free, n_free)    out or NULL       (curvefit.py:236-243; NaN on failure)
 *   status (n_vox) int8, nfev (n_vox) int32, cost (n_vox) = 0.5*sum(res^2): out, each may be NULL
 *          (PNX_MEM_DEVICE: status and cost are required when pcov is requested)
 */
PNX_API int pnx_curvefit_batch_f64(const pnx_curvefit_opts *opts, int64_t n_vox, const double *b, const double *y,
                           const double *p0, const double *lo, const double *hi, const double *fixed,
                           double *popt, double *pcov, int8_t *status, int32_t *nfev, double *cost,
                           int mem, int device, void *stream);

/*
 * The same fit with fp32 STORAGE: b, y, p0 / lo / hi, fixed in and popt, pcov, cost out are float; the arithmetic is
 * the fp64 of pnx_curvefit_batch_f64 (values are widened on the device, results narrowed there).  This is what the
 * reference computes for a float32 image -- curve_fit casts ydata to float64 (scipy:_minpack_py.py:930) -- without
 * the host-side float64 copy and with half the PCIe traffic.  SURVEY.md 8b: "T in {f32, f64} via suffix".
 */
PNX_API int pnx_curvefit_batch_f32(const pnx_curvefit_opts *opts, int64_t n_vox, const float *b, const float *y, const float *p0,
                           const float *lo, const float *hi, const float *fixed, float *popt, float *pcov,
                           int8_t *status, int32_t *nfev, float *cost, int mem, int device, void *stream);

/*
 * NNLS plan: everything that is shared by all voxels of one fit -- the regularised design matrix
 * A = [basis; reg] (nnls_solver.py:61-73) -- is uploaded and factored into its Gram form once.
 *   basis (n_meas, n_bins) row-major host;  reg (n_reg, n_bins) row-major host or NULL (n_reg = 0).
 * The plan picks the kernel; all of them walk the Lawson-Hanson path of scipy.optimize.nnls (same iteration counts):
 *   - reg = one of the reference's banded matrices (model_functions/nnls.py:46-85, orders 1-3) and n_meas <= 32 (every
 *     configuration the reference ships): basis resident in LDS, residual-form dual (csrc/pnx_nnls_blk.hip).  Its first
 *     instantiation holds 128 passive-set positions (twelve voxels in flight per CU); a voxel that wants more is solved again
 *     by a second instantiation with 256 positions (eight per CU) in one pass behind the call.  With a regulariser several
 *     times stronger than the reference's mu = 0.02 that concerns many voxels, so a call of >= 49 152 voxels solves a pilot of
 *     its first 12 288 and, when more than 15 % of those are handed over, the rest of the call goes to the second instantiation
 *     directly -- decided on the device, from the pilot's voxels only.  Device memory: a plan owns 0.2 GB of per-wave
 *     slabs (the first instantiation); the slabs of the two kernels behind it -- 0.55 GB the second instantiation, 1 GB the
 *     Gram-form kernel that remains the last resort -- exist ONCE per device and are shared by all such plans on it (N plans:
 *     1.55 + N x 0.2 GB; their uses are ordered on the device by events, calls stay asynchronous; pnx_release_staging frees the
 *     set once no plan is left).  Passive sets of up to 16 bins evaluate the dual in Gram form, w = A^T y - G[:, P] x_P (16 rows
 *     of G out of L2 instead of the 64 KB basis out of LDS), larger ones in residual form;
 *   - no regulariser / an all-zero one (reg_order = 0, the reference default): QR form (pnx_nnls_qr.hip) -- Q and R in LDS
 *     up to 32 measurements, in a per-wave global slab from 33 to 128 (264 KB per resident wave, allocated on the first
 *     solve of such a plan; up to round 3 more than 64 measurements were refused); the normal-equation kernel would pick other columns than SciPy on a rank-deficient basis, and
 *     this library never changes the algorithm silently;
 *   - anything else (dense regularisers, 33..128 measurements): Gram form (pnx_nnls.hip).
 * n_bins <= 512 (PNX_ERR_UNSUPPORTED beyond; the reference has no such limit, model_functions/nnls.py:37-77; up to round 3 the
 * limit was 256).  Up to 256 bins a lane of the wavefront that owns a voxel holds four bins, and every fast path is built on
 * that: the block kernel's LDS copy of the basis (32 x 258 doubles = 66 KB; at eight bins per lane 131 KB of a CU's 160 KB, two
 * voxels in flight per CU instead of twelve), the general kernel's 128 registers (sixteen waves per CU), the MFMA Gram step's
 * 256-column product.  257 .. 512 bins run on SEPARATE, SLOWER instantiations (a "wide" plan): the Gram-form kernel with six
 * (up to 384 bins) or eight bins per lane and the 256 passive-set positions of the narrow kernel (twelve waves per CU, A^T y
 * on the vector unit, no block kernel, no MFMA step; a voxel whose passive set wants a 257th position is handed to a
 * 512-position instantiation), and the QR-form kernels with a wider dual.  Same algorithm and decisions on the reference
 * fixtures g11_* and the oracle tests; NOT the same parity bar everywhere: a REGULARISED wide plan evaluates everything in Gram
 * form (condition number squared; the narrow plans' residual-form dual does not exist at eight bins per lane), and on random
 * ill-conditioned cases 2-3 % of them leave the oracle's path -- status flips at the iteration limit, coefficients off by up to
 * 1.5e-4 of the peak against the narrow plans' 1e-6 (profiles/r05_fuzz_nnls_wide.json: 7 of 200 cases; the listed cases are
 * the expected failures of `tests/fuzz_gpu_vs_oracle_nnls.py --wide`).  Measured on the C4 signal with 32 b-values: 3.4 M voxels/s at 300 bins, 2.7 M at
 * 384 and 1.7 M at 512 with the order-2 regulariser (10.3 M at 250 on 2^20 voxels: the step at 257 bins is a factor of 3), 8.8 M / 7.9 M
 * without (10.5 M at 250).
 * pnx_nnls_aty_f64 (the MFMA step on its own) stays a 256-column layout and refuses a wide plan.
 */
typedef struct pnx_nnls_plan pnx_nnls_plan;
PNX_API int pnx_nnls_plan_create(pnx_nnls_plan **plan, int n_meas, int n_bins, const double *basis, const double *reg,
                         int n_reg, int device);
PNX_API int pnx_nnls_plan_destroy(pnx_nnls_plan *plan);

/*
 * Batched NNLS  min ||A x - [y;0]||_2, x >= 0  per voxel, fp64, results matching scipy.optimize.nnls.
 * Replaces: NNLSSolver._fit_data (nnls_solver.py:129-180) incl. _extend_signal (never materialised).
 *   y (n_vox, n_meas) in;  coeff (n_vox, n_bins) out;  rnorm (n_vox) out = ||A x - y_ext||_2;
 *   status (n_vox) int8, iters (n_vox) int32: out, may be NULL.  max_iter: reference's max_iter
 *   (0 -> 3*n_bins like SciPy).
 */
PNX_API int pnx_nnls_solve_f64(pnx_nnls_plan *plan, int64_t n_vox, const double *y, int max_iter, double *coeff,
                       double *rnorm, int8_t *status, int32_t *iters, int mem, void *stream);
/* fp32 storage of y, coeff and rnorm, fp64 arithmetic (halves the 8.4 GB coefficient volume of the C4 workload). */
PNX_API int pnx_nnls_solve_f32(pnx_nnls_plan *plan, int64_t n_vox, const float *y, int max_iter, float *coeff, float *rnorm,
                       int8_t *status, int32_t *iters, int mem, void *stream);

/*
 * The MFMA Gram step of the NNLS path on its own: aty (n_vox, 256) = y (n_vox, n_meas) . basis (n_meas, n_bins),
 * columns >= n_bins zero (the layout the active-set kernel consumes).  n_bins <= 256, n_meas <= 64 (PNX_ERR_UNSUPPORTED otherwise).  Device pointers only; n_vox <= 2^20 per call.
 * This is the batched-GEMM part of what NNLSSolver._fit_single_pixel hands to scipy.optimize.nnls per voxel
 * (nnls_solver.py:195-197: A^T y of the normal equations); exposed so that it can be timed and checked by itself.
 */
PNX_API int pnx_nnls_aty_f64(pnx_nnls_plan *plan, int64_t n_vox, const double *y_dev, double *aty_dev, void *stream);

/* One-shot convenience: plan_create + solve + plan_destroy with host pointers. */
PNX_API int pnx_nnls_batch_f64(int64_t n_vox, int n_meas, int n_bins, const double *basis, const double *reg, int n_reg,
                       const double *y, int max_iter, double *coeff, double *rnorm, int8_t *status,
                       int32_t *iters, int device);

/*
 * Design-matrix builders (model_functions/nnls.py:17-85), host in / host out, computed on the device
 * in fp64 so that a host language without numpy can build the same matrices.
 */
PNX_API int pnx_nnls_bins(double d_min, double d_max, int n_bins, double *bins);
PNX_API int pnx_nnls_basis(int n_meas, const double *b, int n_bins, const double *bins, double *basis, int device);
PNX_API int pnx_nnls_regularization_matrix(int n_bins, int order, double mu, double *reg);

/*
 * NNLS spectrum post-processing on the device: what pyneapple.utility.spectrum does per voxel
 * (src/pyneapple/utility/spectrum.py:13-215) for all voxels at once.
 *   find_spectrum_peaks(spectrum, bins, height, regularized): scipy.signal.find_peaks(spectrum, height=height); fractions
 *     = peak heights, or -- regularized != 0 -- the Gaussian area height * FWHM / (2 sqrt(2 ln 2)) * sqrt(2 pi) with the FWHM
 *     of scipy.signal.peak_widths(rel_height) (calculate_peak_area); normalised to sum 1; d = bins[peak].
 *   apply_cutoffs(d, f, cutoffs): per range (lo, hi) no peak -> NaN, one -> kept, several -> geometric_mean_peak
 *     (log10 of the weighted geometric mean position -- the reference's own convention -- and the summed fraction);
 *     fractions renormalised over the ranges.
 *   spectrum (n_vox, n_bins) host|device, 3 <= n_bins <= 512; bins (n_bins,) host (up to 256 bins they travel in the kernel
 *     arguments, beyond in a stream-ordered device buffer: a device-mode call only enqueues either way); cutoffs (n_cut, 2) host.
 *   n_peaks (n_vox) int32: peaks found (may exceed max_peaks: the first max_peaks <= 64 are reported; a spectrum with more
 *     than 64 peaks -- 16 when it has a flat-topped rise, which takes SciPy's sequential scan on one lane -- gets NaN rows:
 *     its fractions would have to be normalised over peaks the table cannot hold; a 250-bin spectrum has at most 124 maxima);
 *   d_values / f_values (n_vox, max_peaks) NaN padded; d_cut / f_cut (n_vox, n_cut <= 8).  Outputs host|device as `mem`.
 */
PNX_API int pnx_nnls_spectrum_peaks_f64(int64_t n_vox, int n_bins, const double *spectrum, const double *bins_host, double height,
                                int regularized, double rel_height, int max_peaks, int32_t *n_peaks, double *d_values,
                                double *f_values, int n_cut, const double *cutoffs_host, double *d_cut, double *f_cut, int mem,
                                int device, void *stream);
/*
 * NNLS solve + spectrum post-processing in one call: the (n_vox, n_bins) spectra never leave the device (2 KB per voxel,
 * 8.4 GB for a 256 x 256 x 64 volume); per voxel only the peak table, the cutoff table, rnorm / status / iters come back.
 * Arguments as pnx_nnls_solve_f64 and pnx_nnls_spectrum_peaks_f64; y and every output host|device as `mem`.
 */
PNX_API int pnx_nnls_solve_peaks_f64(pnx_nnls_plan *plan, int64_t n_vox, const double *y, int max_iter, const double *bins_host,
                             double height, int regularized, double rel_height, int max_peaks, int32_t *n_peaks,
                             double *d_values, double *f_values, int n_cut, const double *cutoffs_host, double *d_cut,
                             double *f_cut, double *rnorm, int8_t *status, int32_t *iters, int mem, void *stream);
/*
 * float32 parameter maps (io/nifti.py:279-312 reconstruct_maps): out (n_spatial, k) zero filled, then
 * out[linear_index[i], :] = (float) values[i, :] for the n_px fitted voxels (linear_index = C-order index into the
 * (X, Y, Z) grid).  values (n_px, k) float64, linear_index (n_px) int64, out float32: host|device as `mem`.
 */
PNX_API int pnx_scatter_maps_f32(const double *values, const int64_t *linear_index, int64_t n_px, int k, int64_t n_spatial, float *out,
                         int mem, int device, void *stream);

/*
 * IDEAL level plumbing on the device.
 * pnx_resize2d_f64: resize the first two axes of an (X, Y, C) array (C = product of the trailing axes, contiguous)
 *   to (TX, TY, C) with OpenCV's INTER_LINEAR (method 0) / INTER_CUBIC (method 1) arithmetic -- half-pixel centres,
 *   cubic a = -0.75, replicated border, no anti-aliasing.  Replaces IDEALFitter._interpolate_array, which calls
 *   cv2.resize per slice and channel (fitters/ideal.py:299-320).  in / out: host or device (mem).
 * pnx_ideal_bounds_f64: from a parameter map (n_px, n_params) -- the previous level's result, resized -- the next
 *   level's start values and bounds, parameter-major (n_params, n_px): p0 = clip(map, lo, hi),
 *   lower = clip(p0 (1 - tol), lo, hi), upper = clip(p0 (1 + tol), lo, hi)  (ideal.py:167-198).  Device pointers;
 *   lo / hi / tol (n_params,) on the host.
 */
PNX_API int pnx_resize2d_f64(const double *in, int X, int Y, int64_t C, double *out, int TX, int TY, int method, int mem,
                     int device, void *stream);
PNX_API int pnx_ideal_bounds_f64(const double *map, int64_t n_px, int n_params, const double *lo_host, const double *hi_host,
                         const double *tol_host, double *p0, double *lower, double *upper, int device, void *stream);

/*
 * IDEAL level plumbing between the resize and the fit (fitters/ideal.py:199-254), device pointers:
 * pnx_mask_select_f64: C-order indices of mask[i] > threshold (ideal.py:199 thresholds the resized segmentation) into idx,
 *   their number into *n_selected (host) -- synchronises the stream, the caller sizes the level's arrays with it;
 * pnx_gather_rows_f64: dst (n_sel, c) = src[idx, :] (signal rows / start values of the fitted voxels, ideal.py:211-216);
 * pnx_scatter_rows_t_f64: the level's parameter map (n_total, k), zero outside the fitted voxels, from the solver's
 *   parameter-major estimates popt (k, n_sel) (ideal.py:243-252); idx NULL = every voxel in order;
 * pnx_row_ss_tot_f64: sum_j (y_ij - mean_i)^2 per row, the SS_tot of R^2 (fitters/base.py:179-183).
 * The last three only enqueue.
 */
PNX_API int pnx_mask_select_f64(const double *mask, int64_t n, double threshold, int64_t *idx, int64_t *n_selected, int device,
                        void *stream);
PNX_API int pnx_gather_rows_f64(const double *src, int64_t c, const int64_t *idx, int64_t n_sel, double *dst, int device, void *stream);
PNX_API int pnx_scatter_rows_t_f64(const double *popt, const int64_t *idx, int64_t n_sel, int k, int64_t n_total, double *pmap, int device,
                           void *stream);
PNX_API int pnx_row_ss_tot_f64(const double *y, int64_t n, int c, double *out, int device, void *stream);

/*
 * Bulk copies between pageable host arrays and device buffers -- what torch's tensor.to(device) / tensor.cpu() (or a plain
 * hipMemcpy) do for the callers that keep a volume resident across calls (the IDEAL level driver: fitters/ideal.py:226-259
 * holds image, masks and maps in host arrays; here they live in HBM and cross PCIe once each way).  The range is cut into
 * 32 MiB pieces (PNX_COPY_PIECE_MB) copied by `threads` host threads (0 = PNX_COPY_THREADS, default 4) on their own streams, a
 * download touches its destination pages first.  Both synchronise `stream` before they start and return when the bytes have
 * landed.
 */
PNX_API int pnx_upload(void *dst_device, const void *src_host, int64_t bytes, int device, void *stream, int threads);
PNX_API int pnx_download(void *dst_host, const void *src_device, int64_t bytes, int device, void *stream, int threads);

/*
 * Per-label column sums of an (n, c) row matrix and the rows per label: the reduction SegmentationWiseFitter performs with a
 * boolean gather + np.mean per label before it fits one row per label (fitters/segmentationwise.py:101-123).  labels[i] in
 * [0, n_labels) is the row's label POSITION (np.unique order); rows with a label outside that range are skipped.
 * sums (n_labels, c), counts (n_labels).  Deterministic (fixed summation order, no atomics).  mem: host pointers (the rows are
 * uploaded with pnx_upload's threaded copy) or device pointers; n_labels * (c + 1) <= 8192.  Synchronises the stream.
 */
PNX_API int pnx_label_sums_f64(const double *rows, const int32_t *labels, int64_t n, int c, int n_labels, double *sums, int64_t *counts,
                       int mem, int device, void *stream);

/*
 * Residual / Jacobian / normal-equation sweep at given parameters (one pass of the LM inner loop as a
 * standalone, HBM-streaming kernel): for every voxel reads y (n_b) and params (n_all), writes
 * cost = 0.5*||r||^2, g = J^T r (n_all) and the upper triangle of J^T J (n_all(n_all+1)/2).
 * Device pointers only.  T = float (f32) or double (f64).
 *   y (n_vox, n_b); params (n_all, n_vox); out_cost (n_vox); out_g (n_all, n_vox); out_jtj (n_tri, n_vox)
 */
PNX_API int pnx_sweep_f32(int model, int64_t n_vox, int n_b, const float *b_host, const float *y, const float *params,
                  float *out_cost, float *out_g, float *out_jtj, int device, void *stream);
PNX_API int pnx_sweep_f64(int model, int64_t n_vox, int n_b, const double *b_host, const double *y, const double *params,
                  double *out_cost, double *out_g, double *out_jtj, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PNX_H */
