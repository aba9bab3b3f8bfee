// pnx_api.hip -- the C ABI declared in include/pnx.h (host side: argument checks, staging, launches).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "pnx_curvefit_kernel.hpp"
#include "pnx_internal.hpp"
#include "pnx_nnls.hpp"

namespace pnx {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define PNX_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) return set_error(PNX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

struct DeviceInfo {
    bool ok = false;
    int cus = 0;
    // ring of work-queue counters so that asynchronous (device-mode) calls never share one
    unsigned long long *queues = nullptr;
    int next_queue = 0;
};
static constexpr int kQueueRing = 256;
static std::mutex g_mu;
static DeviceInfo g_dev[64];

static int get_device(int device, DeviceInfo **out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_error(PNX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n || device >= 64) return set_error(PNX_ERR_INVALID, "device %d out of range (%d visible)", device, n);
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceInfo &d = g_dev[device];
    if (!d.ok) {
        hipDeviceProp_t prop;
        PNX_HIP(hipGetDeviceProperties(&prop, device));
        d.cus = prop.multiProcessorCount;
        PNX_HIP(hipSetDevice(device));
        PNX_HIP(hipMalloc(&d.queues, sizeof(unsigned long long) * kQueueRing));
        d.ok = true;
    }
    *out = &d;
    return PNX_OK;
}

static unsigned long long *next_queue(DeviceInfo *d) {
    std::lock_guard<std::mutex> lk(g_mu);
    unsigned long long *q = d->queues + d->next_queue;
    d->next_queue = (d->next_queue + 1) % kQueueRing;
    return q;
}

// RAII device buffer for the host-staging path
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) return set_error(PNX_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return PNX_OK;
    }
};

static int model_n_params(int model) {
    switch (model) {
    case PNX_MODEL_MONO: return 2;
    case PNX_MODEL_BI_REDUCED: return 3;
    case PNX_MODEL_BI_S0: return 4;
    case PNX_MODEL_BI_FULL: return 4;
    case PNX_MODEL_TRI_REDUCED: return 5;
    case PNX_MODEL_TRI_S0: return 6;
    case PNX_MODEL_TRI_FULL: return 6;
    }
    return -1;
}

typedef int (*launch_fn)(int, int, const CurvefitArgs *, int, void *);
static launch_fn g_launch[7] = {pnx_launch_curvefit_m0, pnx_launch_curvefit_m1, pnx_launch_curvefit_m2,
                                pnx_launch_curvefit_m3, pnx_launch_curvefit_m4, pnx_launch_curvefit_m5,
                                pnx_launch_curvefit_m6};

static int curvefit_device(const pnx_curvefit_opts *o, int64_t n_vox, const double *b, const double *y_d,
                           const double *p0, const double *lo, const double *hi, const double *fixed, double *popt_d,
                           double *pcov_d, int8_t *status_d, int32_t *nfev_d, double *cost_d, DeviceInfo *dev,
                           hipStream_t stream) {
    CurvefitArgs a;
    memset(&a, 0, sizeof(a));
    a.y = y_d;
    a.popt = popt_d;
    a.pcov = pcov_d;
    a.status = status_d;
    a.nfev = nfev_d;
    a.cost = cost_d;
    a.n_vox = n_vox;
    a.n_b = o->n_b;
    a.per_voxel = o->per_voxel_p0_bounds;
    a.fixed_per_voxel = o->fixed_per_voxel;
    a.max_nfev = o->max_nfev > 0 ? o->max_nfev : 100 * o->n_free;  // least_squares: max_nfev=None -> 100*n
    a.n_fixed = o->n_fixed;
    a.t1_mode = o->t1_mode;
    a.tr = o->tr;
    a.tm = o->tm;
    a.ftol = o->ftol;
    a.xtol = o->xtol;
    a.gtol = o->gtol;
    for (int k = 0; k < o->n_free; ++k) a.free_idx[k] = o->free_idx[k];
    for (int k = 0; k < o->n_fixed; ++k) a.fixed_idx[k] = o->fixed_idx[k];
    for (int i = 0; i < o->n_b; ++i) a.b[i] = b[i];
    if (o->per_voxel_p0_bounds) {
        a.p0 = p0;
        a.lo = lo;
        a.hi = hi;
    } else {
        for (int k = 0; k < o->n_free; ++k) {
            a.p0s[k] = p0[k];
            a.los[k] = lo[k];
            a.his[k] = hi[k];
        }
    }
    if (o->n_fixed) {
        if (o->fixed_per_voxel)
            a.fixed = fixed;
        else
            for (int k = 0; k < o->n_fixed; ++k) a.fixeds[k] = fixed[k];
    }
    a.queue = next_queue(dev);
    PNX_HIP(hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), stream));
    return g_launch[o->model](o->n_free, o->jac_mode, &a, dev->cus, (void *)stream);
}

static int check_curvefit_opts(const pnx_curvefit_opts *o) {
    if (!o) return set_error(PNX_ERR_INVALID, "opts is NULL");
    int n_all = model_n_params(o->model);
    if (n_all < 0) return set_error(PNX_ERR_INVALID, "unknown model %d", o->model);
    if (o->t1_mode < 0 || o->t1_mode > 2) return set_error(PNX_ERR_INVALID, "t1_mode %d", o->t1_mode);
    if (o->t1_mode) n_all += 1;
    if (o->n_b < 1 || o->n_b > PNX_MAX_BVALUES) return set_error(PNX_ERR_INVALID, "n_b=%d out of range [1,%d]", o->n_b, PNX_MAX_BVALUES);
    if (o->n_free < 1 || o->n_fixed < 0 || o->n_free + o->n_fixed != n_all)
        return set_error(PNX_ERR_INVALID, "n_free=%d + n_fixed=%d != %d parameters of model %d", o->n_free, o->n_fixed, n_all, o->model);
    bool seen[PNX_MAX_PARAMS] = {false};
    for (int k = 0; k < o->n_free; ++k) {
        const int j = o->free_idx[k];
        if (j < 0 || j >= n_all || seen[j] || (k && j <= o->free_idx[k - 1]))
            return set_error(PNX_ERR_INVALID, "free_idx must be ascending, unique positions in [0,%d)", n_all);
        seen[j] = true;
    }
    for (int k = 0; k < o->n_fixed; ++k) {
        const int j = o->fixed_idx[k];
        if (j < 0 || j >= n_all || seen[j]) return set_error(PNX_ERR_INVALID, "fixed_idx overlaps free_idx or is out of range");
        seen[j] = true;
    }
    if (o->jac_mode != PNX_JAC_FD && o->jac_mode != PNX_JAC_ANALYTIC) return set_error(PNX_ERR_INVALID, "jac_mode %d", o->jac_mode);
    if (o->jac_mode == PNX_JAC_FD && o->n_fixed)
        return set_error(PNX_ERR_UNSUPPORTED,
                         "finite-difference Jacobian with fixed parameters: the reference uses the analytic Jacobian "
                         "there (curvefit.py:274-288); pass PNX_JAC_ANALYTIC");
    if (!(o->ftol >= 0) || !(o->xtol >= 0) || !(o->gtol >= 0)) return set_error(PNX_ERR_INVALID, "tolerances must be >= 0");
    return PNX_OK;
}

}  // namespace pnx

using namespace pnx;

extern "C" {

int pnx_version(void) { return PNX_VERSION_MAJOR * 100 + PNX_VERSION_MINOR; }

int pnx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pnx_last_error(char *buf, int n) {
    const int len = (int)strlen(g_err);
    if (buf && n > 0) {
        strncpy(buf, g_err, (size_t)n - 1);
        buf[n - 1] = 0;
    }
    return len;
}

int pnx_model_n_params(int model) {
    const int n = model_n_params(model);
    return n < 0 ? set_error(PNX_ERR_INVALID, "unknown model %d", model) : n;
}

int pnx_curvefit_batch_f64(const pnx_curvefit_opts *o, int64_t n_vox, const double *b, const double *y,
                           const double *p0, const double *lo, const double *hi, const double *fixed, double *popt,
                           double *pcov, int8_t *status, int32_t *nfev, double *cost, int mem, int device,
                           void *stream) {
    int rc = check_curvefit_opts(o);
    if (rc) return rc;
    if (n_vox < 0) return set_error(PNX_ERR_INVALID, "n_vox < 0");
    if (!b || !p0 || !lo || !hi || !popt || (n_vox && !y)) return set_error(PNX_ERR_INVALID, "NULL data pointer");
    if (o->n_fixed && !fixed) return set_error(PNX_ERR_INVALID, "fixed is NULL but n_fixed=%d", o->n_fixed);
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    if (n_vox == 0) return PNX_OK;
    DeviceInfo *dev;
    rc = get_device(device, &dev);
    if (rc) return rc;
    PNX_HIP(hipSetDevice(device));
    const int n = o->n_free;
    if (mem == PNX_MEM_DEVICE) {
        if (pcov && (!status || !cost))
            return set_error(PNX_ERR_INVALID, "device mode: pcov needs the status and cost outputs too (the covariance "
                                              "epilogue kernel reads them)");
    }
    if (mem == PNX_MEM_DEVICE)
        return curvefit_device(o, n_vox, b, y, p0, lo, hi, fixed, popt, pcov, status, nfev, cost, dev, (hipStream_t)stream);

    // ---- host staging (synchronous).  TODO(next): chunked double-buffered streams for volumes > HBM share.
    hipStream_t st = (hipStream_t)stream;
    DevBuf dy, dp0, dlo, dhi, dfx, dpopt, dpcov, dstat, dnfev, dcost;
    const size_t nv = (size_t)n_vox;
    if ((rc = dy.alloc(nv * o->n_b * sizeof(double)))) return rc;
    PNX_HIP(hipMemcpyAsync(dy.p, y, nv * o->n_b * sizeof(double), hipMemcpyHostToDevice, st));
    const double *p0_d = p0, *lo_d = lo, *hi_d = hi, *fx_d = fixed;
    if (o->per_voxel_p0_bounds) {
        const size_t bytes = nv * n * sizeof(double);
        if ((rc = dp0.alloc(bytes)) || (rc = dlo.alloc(bytes)) || (rc = dhi.alloc(bytes))) return rc;
        PNX_HIP(hipMemcpyAsync(dp0.p, p0, bytes, hipMemcpyHostToDevice, st));
        PNX_HIP(hipMemcpyAsync(dlo.p, lo, bytes, hipMemcpyHostToDevice, st));
        PNX_HIP(hipMemcpyAsync(dhi.p, hi, bytes, hipMemcpyHostToDevice, st));
        p0_d = (const double *)dp0.p;
        lo_d = (const double *)dlo.p;
        hi_d = (const double *)dhi.p;
    }
    if (o->n_fixed && o->fixed_per_voxel) {
        const size_t bytes = nv * o->n_fixed * sizeof(double);
        if ((rc = dfx.alloc(bytes))) return rc;
        PNX_HIP(hipMemcpyAsync(dfx.p, fixed, bytes, hipMemcpyHostToDevice, st));
        fx_d = (const double *)dfx.p;
    }
    if ((rc = dpopt.alloc(nv * n * sizeof(double)))) return rc;
    if (pcov && (rc = dpcov.alloc(nv * n * n * sizeof(double)))) return rc;
    if ((status || pcov) && (rc = dstat.alloc(nv))) return rc;
    if (nfev && (rc = dnfev.alloc(nv * sizeof(int32_t)))) return rc;
    if ((cost || pcov) && (rc = dcost.alloc(nv * sizeof(double)))) return rc;
    rc = curvefit_device(o, n_vox, b, (const double *)dy.p, p0_d, lo_d, hi_d, fx_d, (double *)dpopt.p,
                         pcov ? (double *)dpcov.p : nullptr, (status || pcov) ? (int8_t *)dstat.p : nullptr,
                         nfev ? (int32_t *)dnfev.p : nullptr, (cost || pcov) ? (double *)dcost.p : nullptr, dev, st);
    if (rc) return rc;
    PNX_HIP(hipMemcpyAsync(popt, dpopt.p, nv * n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (pcov) PNX_HIP(hipMemcpyAsync(pcov, dpcov.p, nv * n * n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (status) PNX_HIP(hipMemcpyAsync(status, dstat.p, nv, hipMemcpyDeviceToHost, st));
    if (nfev) PNX_HIP(hipMemcpyAsync(nfev, dnfev.p, nv * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (cost) PNX_HIP(hipMemcpyAsync(cost, dcost.p, nv * sizeof(double), hipMemcpyDeviceToHost, st));
    PNX_HIP(hipStreamSynchronize(st));
    return PNX_OK;
}

// ------------------------------------------------------------------------------------------- NNLS
struct pnx_nnls_plan {
    NnlsPlanData d;
};

int pnx_nnls_plan_create(pnx_nnls_plan **plan, int n_meas, int n_bins, const double *basis, const double *reg,
                         int n_reg, int device) {
    if (!plan || !basis) return set_error(PNX_ERR_INVALID, "NULL pointer");
    if (n_meas < 1 || n_bins < 1 || n_reg < 0 || (n_reg && !reg)) return set_error(PNX_ERR_INVALID, "bad NNLS sizes");
    if (n_bins > kNnlsMaxBins) return set_error(PNX_ERR_UNSUPPORTED, "n_bins=%d > %d", n_bins, kNnlsMaxBins);
    if (n_meas > kNnlsMaxMeas) return set_error(PNX_ERR_UNSUPPORTED, "n_meas=%d > %d", n_meas, kNnlsMaxMeas);
    DeviceInfo *dev;
    int rc = get_device(device, &dev);
    if (rc) return rc;
    PNX_HIP(hipSetDevice(device));
    pnx_nnls_plan *p = new pnx_nnls_plan();
    rc = nnls_plan_init(&p->d, n_meas, n_bins, basis, reg, n_reg, device, dev->cus);
    if (rc) {
        nnls_plan_free(&p->d);
        delete p;
        return rc;
    }
    *plan = p;
    return PNX_OK;
}

int pnx_nnls_plan_destroy(pnx_nnls_plan *plan) {
    if (!plan) return PNX_OK;
    nnls_plan_free(&plan->d);
    delete plan;
    return PNX_OK;
}

int pnx_nnls_solve_f64(pnx_nnls_plan *plan, int64_t n_vox, const double *y, int max_iter, double *coeff,
                       double *rnorm, int8_t *status, int32_t *iters, int mem, void *stream) {
    if (!plan) return set_error(PNX_ERR_INVALID, "plan is NULL");
    if (n_vox < 0 || (n_vox && (!y || !coeff || !rnorm))) return set_error(PNX_ERR_INVALID, "NULL data pointer");
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    if (n_vox == 0) return PNX_OK;
    NnlsPlanData &P = plan->d;
    PNX_HIP(hipSetDevice(P.device));
    hipStream_t st = (hipStream_t)stream;
    if (max_iter <= 0) max_iter = 3 * P.n_bins;  // scipy/optimize/_nnls.py:93-94
    if (mem == PNX_MEM_DEVICE) return nnls_solve_device(&P, n_vox, y, max_iter, coeff, rnorm, status, iters, st);
    const size_t nv = (size_t)n_vox;
    // chunked so that scratch + staged outputs stay bounded
    const size_t chunk = 1u << 18;
    DevBuf dy, dc, dr, ds, di;
    int rc;
    const size_t cn = nv < chunk ? nv : chunk;
    if ((rc = dy.alloc(cn * P.n_meas * sizeof(double))) || (rc = dc.alloc(cn * P.n_bins * sizeof(double))) ||
        (rc = dr.alloc(cn * sizeof(double))) || (rc = ds.alloc(cn)) || (rc = di.alloc(cn * sizeof(int32_t))))
        return rc;
    for (size_t off = 0; off < nv; off += chunk) {
        const size_t c = (nv - off) < chunk ? (nv - off) : chunk;
        PNX_HIP(hipMemcpyAsync(dy.p, y + off * P.n_meas, c * P.n_meas * sizeof(double), hipMemcpyHostToDevice, st));
        rc = nnls_solve_device(&P, (int64_t)c, (const double *)dy.p, max_iter, (double *)dc.p, (double *)dr.p,
                               (int8_t *)ds.p, (int32_t *)di.p, st);
        if (rc) return rc;
        PNX_HIP(hipMemcpyAsync(coeff + off * P.n_bins, dc.p, c * P.n_bins * sizeof(double), hipMemcpyDeviceToHost, st));
        PNX_HIP(hipMemcpyAsync(rnorm + off, dr.p, c * sizeof(double), hipMemcpyDeviceToHost, st));
        if (status) PNX_HIP(hipMemcpyAsync(status + off, ds.p, c, hipMemcpyDeviceToHost, st));
        if (iters) PNX_HIP(hipMemcpyAsync(iters + off, di.p, c * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PNX_HIP(hipStreamSynchronize(st));
    }
    return PNX_OK;
}

int pnx_nnls_batch_f64(int64_t n_vox, int n_meas, int n_bins, const double *basis, const double *reg, int n_reg,
                       const double *y, int max_iter, double *coeff, double *rnorm, int8_t *status, int32_t *iters,
                       int device) {
    pnx_nnls_plan *plan = nullptr;
    int rc = pnx_nnls_plan_create(&plan, n_meas, n_bins, basis, reg, n_reg, device);
    if (rc) return rc;
    rc = pnx_nnls_solve_f64(plan, n_vox, y, max_iter, coeff, rnorm, status, iters, PNX_MEM_HOST, nullptr);
    pnx_nnls_plan_destroy(plan);
    return rc;
}

// model_functions/nnls.py:17-28  np.logspace(log10(dmin), log10(dmax), n) = 10 ** linspace(...)
int pnx_nnls_bins(double d_min, double d_max, int n_bins, double *bins) {
    if (!bins || n_bins < 1 || !(d_min > 0) || !(d_max > 0)) return set_error(PNX_ERR_INVALID, "bad bins arguments");
    const double a = log10(d_min), b = log10(d_max);
    const double step = n_bins > 1 ? (b - a) / (double)(n_bins - 1) : 0.0;
    for (int i = 0; i < n_bins; ++i) {
        double e = a + (double)i * step;  // np.linspace: start + i*step, last point forced to `stop`
        if (i == n_bins - 1 && n_bins > 1) e = b;
        bins[i] = pow(10.0, e);
    }
    return PNX_OK;
}

// model_functions/nnls.py:46-85
int pnx_nnls_regularization_matrix(int n_bins, int order, double mu, double *reg) {
    if (!reg || n_bins < 1) return set_error(PNX_ERR_INVALID, "bad regularization arguments");
    if (order < 0 || order > 3) return set_error(PNX_ERR_UNSUPPORTED, "Regularization order %d not supported. Use 0-3.", order);
    const size_t n = (size_t)n_bins;
    for (size_t i = 0; i < n * n; ++i) reg[i] = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (order == 1) {
            reg[i * n + i] = -1.0 * mu;
            if (i + 1 < n) reg[i * n + i + 1] = 1.0 * mu;
        } else if (order == 2) {
            reg[i * n + i] = -2.0 * mu;
            if (i + 1 < n) reg[i * n + i + 1] = 1.0 * mu;
            if (i >= 1) reg[i * n + i - 1] = 1.0 * mu;
        } else if (order == 3) {
            reg[i * n + i] = -6.0 * mu;
            if (i + 1 < n) reg[i * n + i + 1] = 2.0 * mu;
            if (i >= 1) reg[i * n + i - 1] = 2.0 * mu;
            if (i + 2 < n) reg[i * n + i + 2] = 1.0 * mu;
            if (i >= 2) reg[i * n + i - 2] = 1.0 * mu;
        }
    }
    return PNX_OK;
}

int pnx_nnls_basis(int n_meas, const double *b, int n_bins, const double *bins, double *basis, int device) {
    if (!b || !bins || !basis || n_meas < 1 || n_bins < 1) return set_error(PNX_ERR_INVALID, "bad basis arguments");
    DeviceInfo *dev;
    int rc = get_device(device, &dev);
    if (rc) return rc;
    PNX_HIP(hipSetDevice(device));
    return nnls_build_basis(n_meas, b, n_bins, bins, basis);
}

}  // extern "C"
