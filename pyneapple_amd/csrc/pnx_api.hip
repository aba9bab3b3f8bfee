// pnx_api.hip -- the C ABI declared in include/pnx.h (host side: argument checks, staging, launches).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pnx_curvefit_kernel.hpp"
#include "pnx_host_pipeline.hpp"
#include "pnx_internal.hpp"
#include "pnx_nnls.hpp"

namespace pnx {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    char tmp[sizeof(g_err)];  // the format arguments may point into g_err (a message passed on from another thread)
    vsnprintf(tmp, sizeof(tmp), fmt, ap);
    va_end(ap);
    memcpy(g_err, tmp, sizeof(g_err));
    return code;
}
const char *last_error_text() { return g_err; }

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


// ---- host-staging pipeline: pnx_host_pipeline.hpp on the HIP runtime ------------------------------------------
struct HipBackend {
    typedef hipStream_t stream_t;
    typedef hipEvent_t event_t;
    static bool bind_device(int device) { return hipSetDevice(device) == hipSuccess; }
    // kernel streams at the lowest priority: hardware queues are pooled per priority, so a copy never sits in a queue
    // behind a chunk's kernel (see curvefit_streamed)
    static bool stream_create(stream_t *s, bool kernel) {
        int prio_least = 0, prio_greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) return false;
        const int prio_other = prio_least > 0 ? 0 : prio_greatest;
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, kernel ? prio_least : prio_other) == hipSuccess;
    }
    static void stream_destroy(stream_t s) { (void)hipStreamDestroy(s); }
    static bool stream_sync(stream_t s) { return hipStreamSynchronize(s) == hipSuccess; }
    static bool event_create(event_t *e) { return hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess; }
    static void event_destroy(event_t e) { (void)hipEventDestroy(e); }
    static bool event_record(event_t e, stream_t s) { return hipEventRecord(e, s) == hipSuccess; }
    static bool event_sync(event_t e) { return hipEventSynchronize(e) == hipSuccess; }
};
typedef PipeOpsT<hipStream_t> PipeOps;

static int run_pipeline(int n_chunks, int n_slots, int k_streams, int touchers, int device, hipStream_t user_stream,
                        const PipeOps &ops) {
    return run_pipeline_t<HipBackend>(n_chunks, n_slots, k_streams, touchers, device, user_stream, ops, dev_getenv("PNX_HOST_TRACE") != nullptr);
}

// write one byte per page of [p, p + bytes): first-touch faults taken here, in parallel with the running kernel,
// instead of serially inside the D2H copy.  Only bytes inside the range are written (the range is this chunk's own
// slice of a result array, about to be overwritten by its D2H copy).
static void touch_pages(void *p, size_t bytes) {
    if (!p || !bytes) return;
    volatile char *c = (volatile char *)p;
    const uintptr_t a = (uintptr_t)p;
    c[0] = 0;
    for (size_t off = (4096 - (a & 4095)) & 4095; off < bytes; off += 4096) c[off] = 0;
}

// the environment helpers are shared by the translation units (pnx_internal.hpp)
int env_int(const char *name, int dflt, int lo, int hi) {
    const char *e = getenv(name);
    if (!e) return dflt;
    const long v = atol(e);
    return v < lo ? lo : (v > hi ? hi : (int)v);
}
const char *dev_getenv(const char *name) {
    static const bool enabled = [] {
        const char *e = getenv("PNX_ENABLE_TEST_HOOKS");
        return e && e[0] == '1' && e[1] == 0;
    }();
    return enabled ? getenv(name) : nullptr;
}
int dev_env_int(const char *name, int dflt, int lo, int hi) { return dev_getenv(name) ? env_int(name, dflt, lo, hi) : dflt; }

// Helper threads of a host-array call.  One call uses an upload thread, one or two download threads and two page-touch
// helpers; the plugin's n_gpus = N (one call per device at once) and callers with several volumes in flight multiply that,
// on a CPU share of 16 cores per GPU.  The page-touch helpers only pay off while cores are idle: with two calls in flight each
// gets one, with three or more none (the download threads touch their own pages then); PNX_HOST_TOUCHERS / _OUT_THREADS
// set explicitly win.
static std::atomic<int> g_host_calls(0);
struct HostCallGuard {
    int in_flight;
    HostCallGuard() : in_flight(g_host_calls.fetch_add(1) + 1) {}
    ~HostCallGuard() { g_host_calls.fetch_sub(1); }
    int touchers() const { return getenv("PNX_HOST_TOUCHERS") ? env_int("PNX_HOST_TOUCHERS", 2, 0, 8) : (in_flight >= 3 ? 0 : (in_flight == 2 ? 1 : 2)); }
    int out_threads() const { return getenv("PNX_STREAM_OUT_THREADS") ? env_int("PNX_STREAM_OUT_THREADS", 2, 1, 4) : (in_flight >= 4 ? 1 : 2); }
};

struct Carver {  // hands out 256-byte aligned pieces of one device slab
    char *base = nullptr;
    size_t off = 0;
    void *take(size_t bytes) {
        void *p = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
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

// what a streamed launch adds to the kernel arguments (CurvefitArgs::ctl ...); phase 2 = covariance epilogue only
struct StreamLaunch {
    StreamCtl *ctl = nullptr;
    unsigned int *host_flags = nullptr;
    int granule_shift = 0;
    unsigned int spins = 0;
    int phase = 0;
};

static int curvefit_device(const pnx_curvefit_opts *o, int64_t n_vox, const double *b, const double *y_d,
                           const double *p0, const double *lo, const double *hi, const double *fixed, double *popt_d,
                           double *pcov_d, int8_t *status_d, int32_t *nfev_d, double *cost_d, DeviceInfo *dev,
                           hipStream_t stream, const StreamLaunch *sl = nullptr, const int32_t *order = nullptr) {
    CurvefitArgs a;
    memset(&a, 0, sizeof(a));
    if (sl) {
        a.ctl = sl->ctl;
        a.host_flags = sl->host_flags;
        a.granule_shift = sl->granule_shift;
        a.stream_spins = sl->spins;
        a.phase = sl->phase;
    }
    a.order = sl ? nullptr : order;  // device-mode calls only (pnx_curvefit_opts::queue_order): a streamed launch completes its granules in index order
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
    a.absolute_sigma = o->absolute_sigma != 0;
    a.use_sigma = o->sigma != nullptr;
    if (o->sigma)
        for (int i = 0; i < o->n_b; ++i) a.w[i] = 1.0 / o->sigma[i];  // transform = 1.0 / sigma (a zero sigma gives the reference's "Residuals are not finite" failure)
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
    if (a.phase != 2) {
        a.queue = next_queue(dev);
        PNX_HIP(hipMemsetAsync(a.queue, 0, sizeof(unsigned long long), stream));
    }
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

}  // extern "C"

// element-wise conversion between the storage type of an _f32 entry point and the fp64 the kernels compute in
template <typename S, typename D> __global__ void cvt_kernel(const S *__restrict__ src, D *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (D)src[i];
}
template <typename S, typename D> static int cvt(const S *src, D *dst, size_t n, hipStream_t st) {
    if (!n) return PNX_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((cvt_kernel<S, D>), dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n);
    PNX_HIP(hipGetLastError());
    return PNX_OK;
}

// stream-ordered scratch for the device-pointer _f32 path
struct AsyncBuf {
    void *p = nullptr;
    hipStream_t st = nullptr;
    int alloc(size_t bytes, hipStream_t s) {
        st = s;
        hipError_t e = hipMallocAsync(&p, bytes ? bytes : 8, s);
        if (e != hipSuccess) return set_error(PNX_ERR_NOMEM, "hipMallocAsync(%zu): %s", bytes, hipGetErrorString(e));
        return PNX_OK;
    }
    ~AsyncBuf() {
        if (p) (void)hipFreeAsync(p, st);
    }
};

// ---- host-pointer calls, streamed --------------------------------------------------------------------------------
// The chunk ring above launches one persistent kernel per chunk, and every one of them ends in a drain tail (lanes whose
// queue ran dry idle until the slowest voxel of their wave has converged): seven tails cost C3 about 10 ms of its 50.
// The whole call is ONE kernel instead (CurvefitArgs::ctl; per-voxel start values, bounds and fixed maps are uploaded piece by
// piece like the signal):
//   IN    uploads the signal piece by piece and moves the kernel's watermark behind each piece (an 8-byte copy on the
//         same stream, so the data is there before the watermark says so);
//   the kernel's lanes pull voxels in ascending order and wait at the watermark; each wave counts itself out of a granule
//         of voxels once it holds nothing below the granule's end, after a system-scope release of its stores; the wave
//         that completes a granule's count raises a flag in pinned host memory;
//   OUT   polls the flags and, granule by granule, runs the covariance epilogue and downloads the results while the
//         kernel is still fitting.
// The kernel never waits for another kernel, only for the watermark, and that wait is bounded (stream_spins polls, or the
// abort word) so the grid drains whatever happens on the host; a call whose kernel gave up is run again through the ring.
static constexpr int kStreamRetry = -1000;  // internal: "use the chunk ring"

// Device slab, pinned control block and streams of a streamed call.  One set per device is kept between calls (a 2 GB
// hipMalloc / hipFree pair and five stream creations cost 2-3 ms of a 42 ms call); a second call arriving while it is in use
// works on a private set.  pnx_release_staging() frees the kept set.
struct StreamRes {
    void *slab = nullptr;
    size_t slab_bytes = 0;
    void *pin = nullptr;
    size_t pin_bytes = 0;
    std::vector<hipStream_t> s;  // [0] uploads, [1] the persistent kernel (lowest priority), [2..] downloads
    int ensure_slab(size_t bytes) {
        if (slab_bytes >= bytes) return PNX_OK;
        if (slab) (void)hipFree(slab);
        slab = nullptr;
        slab_bytes = 0;
        hipError_t e = hipMalloc(&slab, bytes);
        if (e != hipSuccess) return set_error(PNX_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        slab_bytes = bytes;
        return PNX_OK;
    }
    int ensure_pin(size_t bytes) {
        if (pin_bytes >= bytes) return PNX_OK;
        if (pin) (void)hipHostFree(pin);
        pin = nullptr;
        pin_bytes = 0;
        bytes = (bytes + 65535) & ~(size_t)65535;
        PNX_HIP(hipHostMalloc(&pin, bytes, hipHostMallocCoherent | hipHostMallocMapped));
        pin_bytes = bytes;
        return PNX_OK;
    }
    int ensure_streams(int n) {
        // The runtime multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and a packet queued
        // behind the persistent kernel in ITS hardware queue would wait for the kernel's end -- which, for the upload, never
        // comes (measured: with torch's streams in the process the uploads sat behind the kernel until its poll limit).
        // Queues are pooled per priority, so the kernel's stream gets the lowest priority and a queue of its own; copies and
        // epilogues stay at the default priority (and win the dispatch arbitration against the fit, which is what one wants).
        int prio_least = 0, prio_greatest = 0;
        PNX_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        const int prio_other = prio_least > 0 ? 0 : prio_greatest;  // numerically lower = higher priority
        while ((int)s.size() < n) {
            hipStream_t q = nullptr;
            PNX_HIP(hipStreamCreateWithPriority(&q, hipStreamNonBlocking, s.size() == 1 ? prio_least : prio_other));
            s.push_back(q);
        }
        return PNX_OK;
    }
    void release() {
        if (slab) (void)hipFree(slab);
        if (pin) (void)hipHostFree(pin);
        for (auto q : s)
            if (q) (void)hipStreamDestroy(q);
        slab = pin = nullptr;
        slab_bytes = pin_bytes = 0;
        s.clear();
    }
};
static StreamRes g_sres[64];
static bool g_sres_busy[64] = {false};
// A streamed launch whose watermark did not move (another library's streams share the hardware queue of the upload, DESIGN
// section 5) is an environmental condition that will hold for the next call too: the device's host-array fits then go through
// the chunk ring for the next PNX_STREAM_COOLDOWN calls (or until pnx_release_staging) instead of paying the stall every time.
static std::atomic<int> g_stream_cooldown[64];

struct StreamLease {
    StreamRes *r = nullptr;
    int device = 0;
    bool kept = false;
    explicit StreamLease(int dev) : device(dev) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_sres_busy[dev]) {
            g_sres_busy[dev] = true;
            r = &g_sres[dev];
            kept = true;
        } else {
            r = new StreamRes();
        }
    }
    ~StreamLease() {
        if (kept) {
            // the slab stays for the next call (a 2 GB hipMalloc / hipFree pair costs 2-3 ms of a 40 ms call) unless it is larger than
            // PNX_STREAM_CACHE_MB or than a quarter of the HBM that would be free without it: on a device that is shared with a
            // framework's caching allocator the library does not sit on memory others are short of
            const size_t cap = (size_t)env_int("PNX_STREAM_CACHE_MB", 8192, 0, 1 << 20) << 20;
            size_t free_b = 0, total_b = 0;
            const bool crowded = r->slab_bytes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && r->slab_bytes > (free_b + r->slab_bytes) / 4;
            if (r->slab_bytes > cap || crowded) {
                (void)hipFree(r->slab);
                r->slab = nullptr;
                r->slab_bytes = 0;
            }
            std::lock_guard<std::mutex> lk(g_mu);
            g_sres_busy[device] = false;
        } else {
            r->release();
            delete r;
        }
    }
};

template <typename T>
static int curvefit_streamed(const pnx_curvefit_opts *o, size_t nv, const double *bd, const T *y, const double *p0d,
                             const double *lod, const double *hid, const T *p0_pv, const T *lo_pv, const T *hi_pv,
                             const double *fxd, const T *fixed_pv, T *popt, T *pcov, int8_t *status, int32_t *nfev, T *cost,
                             int gshift, DeviceInfo *dev, int device, hipStream_t user_stream, const HostCallGuard &hg) {
    constexpr bool F32 = sizeof(T) == 4;
    const int n = o->n_free, n_b = o->n_b;
    const int n_fpv = fixed_pv ? o->n_fixed : 0;  // per-voxel fixed maps (n_fixed, n_vox): uploaded piece by piece like the signal
    const bool pv = p0_pv != nullptr;             // per-voxel p0 / bounds (n_free, n_vox) each: likewise
    const size_t G = (size_t)1 << gshift;
    const int n_gran = (int)((nv + G - 1) >> gshift);
    // voxels per upload / watermark step.  Per-voxel p0 / bounds ride along as 3 n row slices per piece: at 128 Ki voxels those
    // are 1 MB copies and the upload (1.57 GB for C3) runs at 36 GB/s and holds the kernel back (53-58 ms, the ring's 55); at
    // 512 Ki 47-50 ms (profiles/stream_pv_probe.py)
    const size_t in_piece = (size_t)dev_env_int("PNX_STREAM_IN_CHUNK", p0_pv ? 1 << 19 : 1 << 17, 1024, 1 << 26);
    const int n_in = (int)((nv + in_piece - 1) / in_piece);
    const bool need_stat = status || pcov, need_cost = cost || pcov;
    const bool trace = dev_getenv("PNX_HOST_TRACE") != nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    auto now = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(); };
    if (user_stream) PNX_HIP(hipStreamSynchronize(user_stream));

    // device: the whole volume's fp64 working set (+ the T-typed transfer buffers of the float32 entry point) + control block
    StreamLease lease(device);
    StreamRes &res = *lease.r;
    double *dy = nullptr, *dpopt = nullptr, *dpcov = nullptr, *dcost = nullptr, *dfx = nullptr, *dp0 = nullptr, *dlo = nullptr, *dhi = nullptr;
    T *ty = nullptr, *tpopt = nullptr, *tpcov = nullptr, *tcost = nullptr, *tfx = nullptr, *tp0 = nullptr, *tlo = nullptr, *thi = nullptr;
    int8_t *dstat = nullptr;
    int32_t *dnfev = nullptr;
    StreamCtl *ctl = nullptr;
    const size_t ctl_bytes = sizeof(StreamCtl) + sizeof(unsigned int) * (size_t)n_gran;
    int rc;
    for (int pass = 0; pass < 2; ++pass) {
        Carver c;
        c.base = (char *)res.slab;
        auto both = [&](double *&d64, T *&t, size_t count) {
            d64 = (double *)c.take(count * sizeof(double));
            t = F32 ? (T *)c.take(count * sizeof(T)) : (T *)d64;
        };
        both(dy, ty, nv * n_b);
        if (n_fpv) both(dfx, tfx, nv * n_fpv);
        if (pv) {
            both(dp0, tp0, nv * n);
            both(dlo, tlo, nv * n);
            both(dhi, thi, nv * n);
        }
        both(dpopt, tpopt, nv * n);
        if (pcov) both(dpcov, tpcov, nv * n * n);
        if (need_stat) dstat = (int8_t *)c.take(nv);
        if (nfev) dnfev = (int32_t *)c.take(nv * sizeof(int32_t));
        if (need_cost) both(dcost, tcost, nv);
        ctl = (StreamCtl *)c.take(ctl_bytes);
        if (pass == 0 && res.ensure_slab(c.off)) return kStreamRetry;  // no room for the whole volume: the ring needs three chunks
    }
    // pinned host: watermark values (source of the 8-byte copies), granule flags (written by the kernel) and, behind them, the
    // abort word the kernel polls while it waits for the watermark (a plain host store raises it: no stream is involved)
    const size_t pin_bytes = sizeof(unsigned long long) * (size_t)(n_in + 1) + sizeof(unsigned int) * (size_t)(n_gran + 1);
    if (res.ensure_pin(pin_bytes)) return kStreamRetry;  // no pinned memory to be had: the ring does without
    memset(res.pin, 0, pin_bytes);
    unsigned long long *wm = (unsigned long long *)res.pin;  // [n_in] (+ one spare word)
    volatile unsigned int *flags = (volatile unsigned int *)(wm + n_in + 1);
    unsigned int *abort_word = (unsigned int *)(flags + n_gran);
    unsigned int *flags_dev = nullptr;
    PNX_HIP(hipHostGetDevicePointer((void **)&flags_dev, (void *)flags, 0));

    // the downloads are pageable copies (the runtime pins the destination pages piece by piece): two threads, each with its
    // own stream and every other granule, keep up with the kernel where one falls 13 ms behind (C3, profiles/stream_sweep.py)
    const int n_out = hg.out_threads();
    if ((rc = res.ensure_streams(2 + n_out))) return rc;
    hipStream_t s_in = res.s[0], s_main = res.s[1];
    PNX_HIP(hipMemsetAsync(ctl, 0, ctl_bytes, s_main));
    PNX_HIP(hipStreamSynchronize(s_main));  // the control block is clean before anybody moves the watermark

    StreamLaunch sl;
    sl.ctl = ctl;
    sl.host_flags = flags_dev;
    sl.granule_shift = gshift;
    sl.spins = (unsigned int)dev_env_int("PNX_STREAM_SPINS", 400000, 1000, 1 << 20);  // ~6 us per poll: 2.4 s, at most ~6 s (the host's own
                                                                                   // watchdog below gives up after PNX_STREAM_STALL_MS)
    sl.phase = 1;
    rc = curvefit_device(o, (int64_t)nv, bd, dy, pv ? dp0 : p0d, pv ? dlo : lod, pv ? dhi : hid, n_fpv ? dfx : fxd, dpopt, dpcov, dstat,
                         dnfev, dcost, dev, s_main, &sl);
    if (rc == PNX_ERR_UNSUPPORTED) return kStreamRetry;  // no streamed instantiation for this combination: the ring has one
    if (rc) return rc;
    const double t_launched = now();

    // the first watermark move, observed on the device: an event behind it on the upload stream
    hipEvent_t ev_first = nullptr;
    PNX_HIP(hipEventCreateWithFlags(&ev_first, hipEventDisableTiming));
    std::atomic<int> first_recorded(0);
    auto touch = [&](int g) {
        const size_t v0 = (size_t)g << gshift, c = std::min(G, nv - v0);
        for (int j = 0; j < n; ++j) touch_pages(popt + j * nv + v0, c * sizeof(T));
        if (pcov) touch_pages(pcov + v0 * n * n, c * n * n * sizeof(T));
        if (status) touch_pages(status + v0, c);
        if (nfev) touch_pages(nfev + v0, c * sizeof(int32_t));
        if (cost) touch_pages(cost + v0, c * sizeof(T));
    };
    StreamedOps ops;
    ops.bind_device = [&]() { return hipSetDevice(device) == hipSuccess; };
    ops.upload_piece = [&](int i) -> int {
        const size_t v0 = (size_t)i * in_piece, c = std::min(in_piece, nv - v0);
        PNX_HIP(hipMemcpyAsync(ty + v0 * n_b, y + v0 * n_b, c * n_b * sizeof(T), hipMemcpyHostToDevice, s_in));
        if constexpr (F32) {
            int r = cvt(ty + v0 * n_b, dy + v0 * n_b, c * n_b, s_in);
            if (r) return r;
        }
        if (pv) {  // parameter-major (n_free, n_vox) start values and bounds: one row slice per parameter and array
            const T *src[3] = {p0_pv, lo_pv, hi_pv};
            T *tdst[3] = {tp0, tlo, thi};
            double *ddst[3] = {dp0, dlo, dhi};
            for (int a3 = 0; a3 < 3; ++a3)
                for (int j = 0; j < n; ++j) {
                    PNX_HIP(hipMemcpyAsync(tdst[a3] + j * nv + v0, src[a3] + j * nv + v0, c * sizeof(T), hipMemcpyHostToDevice, s_in));
                    if constexpr (F32) {
                        int r = cvt(tdst[a3] + j * nv + v0, ddst[a3] + j * nv + v0, c, s_in);
                        if (r) return r;
                    }
                }
        }
        for (int j = 0; j < n_fpv; ++j) {  // parameter-major (n_fixed, n_vox): one row slice per fixed parameter
            PNX_HIP(hipMemcpyAsync(tfx + j * nv + v0, fixed_pv + j * nv + v0, c * sizeof(T), hipMemcpyHostToDevice, s_in));
            if constexpr (F32) {
                int r = cvt(tfx + j * nv + v0, dfx + j * nv + v0, c, s_in);
                if (r) return r;
            }
        }
        wm[i] = v0 + c;
        PNX_HIP(hipMemcpyAsync(&ctl->ready, &wm[i], sizeof(unsigned long long), hipMemcpyHostToDevice, s_in));
        if (i == 0) {
            PNX_HIP(hipEventRecord(ev_first, s_in));
            first_recorded.store(1);
        }
        return PNX_OK;
    };
    ops.upload_sync = [&]() -> int {
        PNX_HIP(hipStreamSynchronize(s_in));
        return PNX_OK;
    };
    ops.first_piece_landed = [&]() { return first_recorded.load() && hipEventQuery(ev_first) == hipSuccess; };
    ops.granule_ready = [&](int g) { return __atomic_load_n(&flags[g], __ATOMIC_ACQUIRE) != 0; };
    ops.download = [&](int g, int ot) -> int {
        hipStream_t s_out = res.s[2 + ot];
        StreamLaunch s2;
        s2.phase = 2;
        const size_t v0 = (size_t)g << gshift, c = std::min(G, nv - v0);
        if (pcov) {
            int r = curvefit_device(o, (int64_t)c, bd, nullptr, pv ? dp0 : p0d, pv ? dlo : lod, pv ? dhi : hid, n_fpv ? dfx : fxd, nullptr, dpcov + v0 * n * n,
                                    dstat + v0, nullptr, dcost + v0, dev, s_out, &s2);
            if (r) return r;
        }
        if constexpr (F32) {
            int r = PNX_OK;
            for (int j = 0; j < n && !r; ++j) r = cvt(dpopt + j * nv + v0, tpopt + j * nv + v0, c, s_out);
            if (!r && pcov) r = cvt(dpcov + v0 * n * n, tpcov + v0 * n * n, c * n * n, s_out);
            if (!r && cost) r = cvt(dcost + v0, tcost + v0, c, s_out);
            if (r) return r;
        }
        for (int j = 0; j < n; ++j)
            PNX_HIP(hipMemcpyAsync(popt + j * nv + v0, tpopt + j * nv + v0, c * sizeof(T), hipMemcpyDeviceToHost, s_out));
        if (pcov) PNX_HIP(hipMemcpyAsync(pcov + v0 * n * n, tpcov + v0 * n * n, c * n * n * sizeof(T), hipMemcpyDeviceToHost, s_out));
        if (status) PNX_HIP(hipMemcpyAsync(status + v0, dstat + v0, c, hipMemcpyDeviceToHost, s_out));
        if (nfev) PNX_HIP(hipMemcpyAsync(nfev + v0, dnfev + v0, c * sizeof(int32_t), hipMemcpyDeviceToHost, s_out));
        if (cost) PNX_HIP(hipMemcpyAsync(cost + v0, tcost + v0, c * sizeof(T), hipMemcpyDeviceToHost, s_out));
        PNX_HIP(hipStreamSynchronize(s_out));
        return PNX_OK;
    };
    ops.touch = touch;
    ops.abort_kernel = [&]() { __atomic_store_n(abort_word, 1u, __ATOMIC_RELEASE); };
    ops.kernel_state = [&]() -> int {
        const hipError_t e = hipStreamQuery(s_main);
        if (e == hipErrorNotReady) return 0;
        return e == hipSuccess ? 1 : set_error(PNX_ERR_HIP, "streamed curve fit kernel: %s", hipGetErrorString(e));
    };
    ops.kernel_wait = [&]() -> int {
        const hipError_t e = hipStreamSynchronize(s_main);
        return e == hipSuccess ? PNX_OK : set_error(PNX_ERR_HIP, "streamed curve fit kernel: %s", hipGetErrorString(e));
    };
    // a watermark that has not moved this long after the launch will not move: the first piece lands after 0.7 ms (C3), a
    // 512 Ki-voxel piece with per-voxel arrays after 3-4 ms
    const double stall_ms = dev_env_int("PNX_STREAM_STALL_MS", 50, 1, 60000);
    bool stalled = false;
    StreamedTimes times;
    rc = run_streamed(n_in, n_gran, n_out, hg.touchers(), stall_ms, dev_env_int("PNX_STREAM_TEST_DELAY_MS", 0, 0, 60000),
                      ops, &stalled, trace ? &times : nullptr, now);
    (void)hipStreamSynchronize(s_in);  // nothing of this call is left on the kept streams
    (void)hipEventDestroy(ev_first);
    if (rc) return rc;
    StreamCtl head;
    PNX_HIP(hipMemcpy(&head, ctl, sizeof(StreamCtl), hipMemcpyDeviceToHost));
    if (trace) {
        fprintf(stderr, "[pnx stream] %d granules of %zu voxels, %d upload pieces; launched %.2f kernel_done %.2f all_done %.2f ms%s%s\n",
                n_gran, G, n_in, t_launched, times.t_kernel, now(), head.timed_out ? " TIMED OUT" : "", stalled ? " STALLED (gave up)" : "");
        for (int i = 0; i < n_in; i += std::max(1, n_in / 8)) fprintf(stderr, "[pnx stream] upload piece %d enqueued by %.2f ms\n", i, times.t_in[i]);
        for (int g = 0; g < n_gran; g += std::max(1, n_gran / 8))
            fprintf(stderr, "[pnx stream] granule %d complete at %.2f, downloaded by %.2f ms\n", g, times.t_flag[g], times.t_out[g]);
    }
    if (stalled || head.timed_out) {
        g_stream_cooldown[device].store(dev_env_int("PNX_STREAM_COOLDOWN", 32, 0, 1 << 20));
        return kStreamRetry;
    }
    return PNX_OK;
}

// T = double: the fp64 entry point.  T = float: fp32 STORAGE (signal, p0 / bounds / fixed maps in, popt / pcov / cost
// out) with the same fp64 arithmetic -- what the reference does with a float32 image (curve_fit casts ydata to float64).
template <typename T>
static int curvefit_batch(const pnx_curvefit_opts *o, int64_t n_vox, const T *b, const T *y, const T *p0, const T *lo,
                          const T *hi, const T *fixed, T *popt, T *pcov, int8_t *status, int32_t *nfev, T *cost, int mem,
                          int device, void *stream) {
    constexpr bool F32 = sizeof(T) == 4;
    int rc = check_curvefit_opts(o);
    if (rc) return rc;
    if (n_vox < 0) return set_error(PNX_ERR_INVALID, "n_vox < 0");
    if (!b || !p0 || !lo || !hi || !popt || (n_vox && !y)) return set_error(PNX_ERR_INVALID, "NULL data pointer");
    if (o->n_fixed && !fixed) return set_error(PNX_ERR_INVALID, "fixed is NULL but n_fixed=%d", o->n_fixed);
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    if (o->queue_order && mem != PNX_MEM_DEVICE)
        return set_error(PNX_ERR_INVALID, "opts->queue_order is for PNX_MEM_DEVICE calls (a host-array call completes its pieces in index order)");
    if (n_vox == 0) return PNX_OK;
    DeviceInfo *dev;
    rc = get_device(device, &dev);
    if (rc) return rc;
    PNX_HIP(hipSetDevice(device));
    const int n = o->n_free;
    const size_t nv = (size_t)n_vox;
    const bool pv = o->per_voxel_p0_bounds != 0, fpv = o->n_fixed && o->fixed_per_voxel;
    // the small shared host arrays (b-values, shared p0 / bounds / fixed values) as fp64
    double bd[PNX_MAX_BVALUES], p0d[PNX_MAX_PARAMS], lod[PNX_MAX_PARAMS], hid[PNX_MAX_PARAMS], fxd[PNX_MAX_PARAMS];
    for (int i = 0; i < o->n_b; ++i) bd[i] = (double)b[i];
    if (!pv)
        for (int k = 0; k < n; ++k) {
            p0d[k] = (double)p0[k];
            lod[k] = (double)lo[k];
            hid[k] = (double)hi[k];
        }
    if (o->n_fixed && !fpv)
        for (int k = 0; k < o->n_fixed; ++k) fxd[k] = (double)fixed[k];
    auto as_d = [](const T *p) { return reinterpret_cast<const double *>(p); };  // only used when T is double

    if (mem == PNX_MEM_DEVICE) {
        if (pcov && (!status || !cost))
            return set_error(PNX_ERR_INVALID, "device mode: pcov needs the status and cost outputs too (the covariance "
                                              "epilogue kernel reads them)");
        hipStream_t st = (hipStream_t)stream;
        if constexpr (!F32) {
            return curvefit_device(o, n_vox, bd, as_d(y), pv ? as_d(p0) : p0d, pv ? as_d(lo) : lod, pv ? as_d(hi) : hid,
                                   fpv ? as_d(fixed) : fxd, (double *)popt, (double *)pcov, status, nfev, (double *)cost, dev, st, nullptr,
                                   o->queue_order);
        } else {
            AsyncBuf y64, p64, l64, h64, f64, o64, c64, k64;
            if ((rc = y64.alloc(nv * o->n_b * 8, st)) || (rc = cvt(y, (double *)y64.p, nv * o->n_b, st))) return rc;
            if (pv) {
                if ((rc = p64.alloc(nv * n * 8, st)) || (rc = l64.alloc(nv * n * 8, st)) || (rc = h64.alloc(nv * n * 8, st))) return rc;
                if ((rc = cvt(p0, (double *)p64.p, nv * n, st)) || (rc = cvt(lo, (double *)l64.p, nv * n, st)) ||
                    (rc = cvt(hi, (double *)h64.p, nv * n, st)))
                    return rc;
            }
            if (fpv && ((rc = f64.alloc(nv * o->n_fixed * 8, st)) || (rc = cvt(fixed, (double *)f64.p, nv * o->n_fixed, st)))) return rc;
            if ((rc = o64.alloc(nv * n * 8, st))) return rc;
            if (pcov && (rc = c64.alloc(nv * n * n * 8, st))) return rc;
            if ((cost || pcov) && (rc = k64.alloc(nv * 8, st))) return rc;
            rc = curvefit_device(o, n_vox, bd, (const double *)y64.p, pv ? (const double *)p64.p : p0d,
                                 pv ? (const double *)l64.p : lod, pv ? (const double *)h64.p : hid,
                                 fpv ? (const double *)f64.p : fxd, (double *)o64.p, pcov ? (double *)c64.p : nullptr, status,
                                 nfev, (double *)k64.p, dev, st, nullptr, o->queue_order);
            if (rc) return rc;
            if ((rc = cvt((const double *)o64.p, popt, nv * n, st))) return rc;
            if (pcov && (rc = cvt((const double *)c64.p, pcov, nv * n * n, st))) return rc;
            if (cost && (rc = cvt((const double *)k64.p, cost, nv, st))) return rc;
            return PNX_OK;  // AsyncBuf destructors enqueue the frees behind the conversions
        }
    }

    HostCallGuard hg;  // this call is in flight from here on (helper-thread budget)
    // ---- host staging, streamed: one persistent kernel for the whole volume (curvefit_streamed above)
    {
        // granule = unit of the download (and of the completion flags): 256 Ki voxels for volumes of 2 Mi voxels and more,
        // 128 Ki below (C3: 2^17 41.4-44 ms, 2^18 40.6-43.9 ms, 2^16 and 2^19 42-45 ms; profiles/stream_sweep.py)
        const int gshift = dev_env_int("PNX_STREAM_GRANULE_SHIFT", nv >= ((size_t)1 << 21) ? 18 : 17, 10, 24);
        const size_t per_vox = (size_t)(o->n_b + n + (pcov ? n * n : 0) + 2 + (fpv ? o->n_fixed : 0) + (pv ? 3 * n : 0)) * (F32 ? 12 : 8);
        size_t max_bytes = (size_t)dev_env_int("PNX_STREAM_MAX_MB", 65536, 1, 1 << 20) << 20;
        {   // never more than half of what is free now (plus the kept slab, which would be reused): a volume beyond that goes
            // through the ring's three chunk slots instead of one huge hipMalloc that fails or starves the process
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                size_t kept = 0;
                {
                    std::lock_guard<std::mutex> lk(g_mu);
                    if (!g_sres_busy[device]) kept = g_sres[device].slab_bytes;
                }
                max_bytes = std::min(max_bytes, free_b / 2 + kept);
            }
        }
        // not for the kernels that need (almost) every register of a lane (pnx_curvefit_inst.hip launch_pv): the copies that
        // feed a streamed kernel have to fit beside it
        const bool tight = n >= 6 || (n >= 4 && o->t1_mode);
        bool cooling = false;
        if (g_stream_cooldown[device].load() > 0) {  // a recent launch on this device stalled: the ring, without trying again
            g_stream_cooldown[device].fetch_sub(1);
            cooling = true;
        }
        if (!cooling && dev_env_int("PNX_HOST_STREAM", 1, 0, 1) && !(pv && o->n_fixed) && !tight && nv > ((size_t)1 << gshift) &&
            nv < ((size_t)1 << 31) && nv * per_vox <= max_bytes) {
            rc = curvefit_streamed<T>(o, nv, bd, y, p0d, lod, hid, pv ? p0 : nullptr, pv ? lo : nullptr, pv ? hi : nullptr, fxd,
                                      fpv ? fixed : nullptr, popt, pcov, status, nfev, cost, gshift, dev, device, (hipStream_t)stream, hg);
            if (rc != kStreamRetry) return rc;
            static std::atomic<bool> warned(false);
            if (dev_getenv("PNX_HOST_TRACE") || !warned.exchange(true))
                fprintf(stderr, "[pnx stream] the streamed launch could not be used (no room for the staging slab, or its upload did not "
                                "start within PNX_STREAM_STALL_MS); running the call through the chunk ring, and after a stall the next "
                                "PNX_STREAM_COOLDOWN calls of this device too. PNX_HOST_STREAM=0 skips the attempt.\n");
        }
    }

    // ---- host staging: chunk ring (run_pipeline above)
    // float32 transfers are half as long per voxel: a larger chunk (fewer drain tails of the persistent kernel) at the same
    // exposed transfer latency -- C3 float32: 86.3 M voxels/s at 768 Ki, 89.8 M at 1 Mi, 82.5 M at 2 Mi (profiles/host_chunk_sweep_f32.py)
    const size_t chunk = (size_t)dev_env_int("PNX_HOST_CHUNK", F32 ? 1 << 20 : 3 << 18, 1024, 1 << 26);
    // chunk boundaries: with three or more full chunks the first and the last piece are a quarter chunk -- the first kernel
    // starts after a quarter of an upload, and the serial tail (last kernel, last download) is a quarter as long
    std::vector<size_t> bounds;
    {
        const size_t ramp = dev_env_int("PNX_HOST_RAMP", 1, 0, 1) && nv >= 3 * chunk ? chunk / 4 : 0;
        size_t v = 0;
        bounds.push_back(0);
        if (ramp) bounds.push_back(v = ramp);
        const size_t body_end = nv - ramp;
        while (v < body_end) bounds.push_back(v = (body_end - v) < chunk ? body_end : v + chunk);
        if (ramp) bounds.push_back(nv);
    }
    const int n_chunks = (int)bounds.size() - 1;
    const int n_slots = n_chunks < 3 ? n_chunks : dev_env_int("PNX_HOST_SLOTS", 3, 2, 8);
    const size_t cap = nv < chunk ? nv : chunk;
    const bool need_stat = status || pcov, need_cost = cost || pcov;
    struct Slot {
        DevBuf slab;
        // fp64 working set of the kernels ...
        double *y = nullptr, *p0 = nullptr, *lo = nullptr, *hi = nullptr, *fx = nullptr, *popt = nullptr, *pcov = nullptr,
               *cost = nullptr;
        // ... and the T-typed transfer buffers (the same memory when T is double)
        T *ty = nullptr, *tp0 = nullptr, *tlo = nullptr, *thi = nullptr, *tfx = nullptr, *tpopt = nullptr, *tpcov = nullptr,
          *tcost = nullptr;
        int8_t *stat = nullptr;
        int32_t *nfev = nullptr;
    };
    std::vector<Slot> slots(n_slots);
    for (int w = 0; w < n_slots; ++w) {
        Slot &S = slots[w];
        for (int pass = 0; pass < 2; ++pass) {  // pass 0 sizes the slab, pass 1 carves it
            Carver c;
            c.base = (char *)S.slab.p;
            auto both = [&](double *&d64, T *&t, size_t count) {
                d64 = (double *)c.take(count * sizeof(double));
                t = F32 ? (T *)c.take(count * sizeof(T)) : (T *)d64;
            };
            both(S.y, S.ty, cap * o->n_b);
            if (pv) {
                both(S.p0, S.tp0, cap * n);
                both(S.lo, S.tlo, cap * n);
                both(S.hi, S.thi, cap * n);
            }
            if (fpv) both(S.fx, S.tfx, cap * o->n_fixed);
            both(S.popt, S.tpopt, cap * n);
            if (pcov) both(S.pcov, S.tpcov, cap * n * n);
            if (need_stat) S.stat = (int8_t *)c.take(cap);
            if (nfev) S.nfev = (int32_t *)c.take(cap * sizeof(int32_t));
            if (need_cost) both(S.cost, S.tcost, cap);
            if (pass == 0 && (rc = S.slab.alloc(c.off))) return rc;
        }
    }
    auto span = [&](int k, size_t &v0, size_t &c) {
        v0 = bounds[k];
        c = bounds[k + 1] - v0;
    };
    PipeOps ops;
    ops.h2d = [&](int k, int slot, hipStream_t st) -> int {
        Slot &S = slots[slot];
        size_t v0, c;
        span(k, v0, c);
        PNX_HIP(hipMemcpyAsync(S.ty, y + v0 * o->n_b, c * o->n_b * sizeof(T), hipMemcpyHostToDevice, st));
        // parameter-major (k, n_vox) arrays: one row slice per parameter, device stride = c
        if (pv)
            for (int j = 0; j < n; ++j) {
                PNX_HIP(hipMemcpyAsync(S.tp0 + j * c, p0 + j * nv + v0, c * sizeof(T), hipMemcpyHostToDevice, st));
                PNX_HIP(hipMemcpyAsync(S.tlo + j * c, lo + j * nv + v0, c * sizeof(T), hipMemcpyHostToDevice, st));
                PNX_HIP(hipMemcpyAsync(S.thi + j * c, hi + j * nv + v0, c * sizeof(T), hipMemcpyHostToDevice, st));
            }
        if (fpv)
            for (int j = 0; j < o->n_fixed; ++j)
                PNX_HIP(hipMemcpyAsync(S.tfx + j * c, fixed + j * nv + v0, c * sizeof(T), hipMemcpyHostToDevice, st));
        return PNX_OK;
    };
    ops.launch = [&](int k, int slot, hipStream_t st) -> int {
        Slot &S = slots[slot];
        size_t v0, c;
        span(k, v0, c);
        int r = PNX_OK;
        if constexpr (F32) {
            if ((r = cvt(S.ty, S.y, c * o->n_b, st))) return r;
            if (pv && ((r = cvt(S.tp0, S.p0, c * n, st)) || (r = cvt(S.tlo, S.lo, c * n, st)) || (r = cvt(S.thi, S.hi, c * n, st)))) return r;
            if (fpv && (r = cvt(S.tfx, S.fx, c * o->n_fixed, st))) return r;
        }
        r = curvefit_device(o, (int64_t)c, bd, S.y, pv ? S.p0 : p0d, pv ? S.lo : lod, pv ? S.hi : hid, fpv ? S.fx : fxd, S.popt,
                            pcov ? S.pcov : nullptr, S.stat, S.nfev, S.cost, dev, st);
        if (r) return r;
        if constexpr (F32) {
            if ((r = cvt(S.popt, S.tpopt, c * n, st))) return r;
            if (pcov && (r = cvt(S.pcov, S.tpcov, c * n * n, st))) return r;
            if (cost && (r = cvt(S.cost, S.tcost, c, st))) return r;
        }
        return PNX_OK;
    };
    ops.touch = [&](int k) {
        size_t v0, c;
        span(k, v0, c);
        for (int j = 0; j < n; ++j) touch_pages(popt + j * nv + v0, c * sizeof(T));
        if (pcov) touch_pages(pcov + v0 * n * n, c * n * n * sizeof(T));
        if (status) touch_pages(status + v0, c);
        if (nfev) touch_pages(nfev + v0, c * sizeof(int32_t));
        if (cost) touch_pages(cost + v0, c * sizeof(T));
    };
    ops.d2h = [&](int k, int slot, hipStream_t st) -> int {
        Slot &S = slots[slot];
        size_t v0, c;
        span(k, v0, c);
        for (int j = 0; j < n; ++j)
            PNX_HIP(hipMemcpyAsync(popt + j * nv + v0, S.tpopt + j * c, c * sizeof(T), hipMemcpyDeviceToHost, st));
        if (pcov) PNX_HIP(hipMemcpyAsync(pcov + v0 * n * n, S.tpcov, c * n * n * sizeof(T), hipMemcpyDeviceToHost, st));
        if (status) PNX_HIP(hipMemcpyAsync(status + v0, S.stat, c, hipMemcpyDeviceToHost, st));
        if (nfev) PNX_HIP(hipMemcpyAsync(nfev + v0, S.nfev, c * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (cost) PNX_HIP(hipMemcpyAsync(cost + v0, S.tcost, c * sizeof(T), hipMemcpyDeviceToHost, st));
        return PNX_OK;
    };
    return run_pipeline(n_chunks, n_slots, dev_env_int("PNX_HOST_KSTREAMS", 2, 1, 4), hg.touchers(), device, (hipStream_t)stream, ops);
}

extern "C" {
int pnx_release_staging(int device) {
    if (device < 0 || device >= 64) return set_error(PNX_ERR_INVALID, "device %d out of range", device);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_sres_busy[device]) return set_error(PNX_ERR_INVALID, "device %d: a streamed call is using the staging set", device);
    g_stream_cooldown[device].store(0);  // and the next host-array fit may try the streamed launch again
    int cur = 0;
    if (g_sres[device].slab || g_sres[device].pin || !g_sres[device].s.empty()) {
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        g_sres[device].release();
        (void)hipSetDevice(cur);
    }
    // the slab set the block-kernel NNLS plans of this device share (pnx_nnls.hpp): freed when no plan holds it, kept otherwise
    (void)nnls_shared_slabs_trim(device);
    return PNX_OK;
}

int pnx_curvefit_batch_f64(const pnx_curvefit_opts *o, int64_t n_vox, const double *b, const double *y,
                           const double *p0, const double *lo, const double *hi, const double *fixed, double *popt,
                           double *pcov, int8_t *status, int32_t *nfev, double *cost, int mem, int device,
                           void *stream) {
    return curvefit_batch<double>(o, n_vox, b, y, p0, lo, hi, fixed, popt, pcov, status, nfev, cost, mem, device, stream);
}

int pnx_curvefit_batch_f32(const pnx_curvefit_opts *o, int64_t n_vox, const float *b, const float *y, const float *p0,
                           const float *lo, const float *hi, const float *fixed, float *popt, float *pcov, int8_t *status,
                           int32_t *nfev, float *cost, int mem, int device, void *stream) {
    return curvefit_batch<float>(o, n_vox, b, y, p0, lo, hi, fixed, popt, pcov, status, nfev, cost, mem, device, stream);
}

// ------------------------------------------------------------------------------------------- NNLS
struct pnx_nnls_plan {
    NnlsPlanData d;
    std::mutex mu;  // the plan's device scratch (ATY chunk, M overflow, queue) serves one solve at a time
};

int pnx_nnls_plan_create(pnx_nnls_plan **plan, int n_meas, int n_bins, const double *basis, const double *reg,
                         int n_reg, int device) {
    if (!plan || !basis) return set_error(PNX_ERR_INVALID, "NULL pointer");
    if (n_meas < 1 || n_bins < 1 || n_reg < 0 || (n_reg && !reg)) return set_error(PNX_ERR_INVALID, "bad NNLS sizes");
    if (n_bins > kNnlsWideBins) return set_error(PNX_ERR_UNSUPPORTED, "n_bins=%d > %d", n_bins, kNnlsWideBins);
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

}  // extern "C"

// T = double: fp64 entry point.  T = float: fp32 storage of the signal, the coefficients and rnorm; fp64 arithmetic.
template <typename T>
static int nnls_solve(pnx_nnls_plan *plan, int64_t n_vox, const T *y, int max_iter, T *coeff, T *rnorm, int8_t *status,
                      int32_t *iters, int mem, void *stream) {
    constexpr bool F32 = sizeof(T) == 4;
    if (!plan) return set_error(PNX_ERR_INVALID, "plan is NULL");
    if (n_vox < 0 || (n_vox && (!y || !coeff || !rnorm))) return set_error(PNX_ERR_INVALID, "NULL data pointer");
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    if (n_vox == 0) return PNX_OK;
    NnlsPlanData &P = plan->d;
    PNX_HIP(hipSetDevice(P.device));
    hipStream_t st = (hipStream_t)stream;
    if (max_iter <= 0) max_iter = 3 * P.n_bins;  // scipy/optimize/_nnls.py:93-94
    const size_t nv = (size_t)n_vox;
    int rc;
    if (mem == PNX_MEM_DEVICE) {
        if constexpr (!F32) {
            return nnls_solve_device(&P, n_vox, (const double *)y, max_iter, (double *)coeff, (double *)rnorm, status, iters, st);
        } else {
            // converted in pieces of the kernel's own chunk so that the fp64 scratch stays at 2.4 GB
            const size_t piece = (size_t)kAtyChunk < nv ? (size_t)kAtyChunk : nv;
            AsyncBuf y64, c64, r64;
            if ((rc = y64.alloc(piece * P.n_meas * 8, st)) || (rc = c64.alloc(piece * P.n_bins * 8, st)) || (rc = r64.alloc(piece * 8, st)))
                return rc;
            for (size_t off = 0; off < nv; off += piece) {
                const size_t c = (nv - off) < piece ? (nv - off) : piece;
                if ((rc = cvt(y + off * P.n_meas, (double *)y64.p, c * P.n_meas, st))) return rc;
                rc = nnls_solve_device(&P, (int64_t)c, (const double *)y64.p, max_iter, (double *)c64.p, (double *)r64.p,
                                       status ? status + off : nullptr, iters ? iters + off : nullptr, st);
                if (rc) return rc;
                if ((rc = cvt((const double *)c64.p, coeff + off * P.n_bins, c * P.n_bins, st)) ||
                    (rc = cvt((const double *)r64.p, rnorm + off, c, st)))
                    return rc;
            }
            return PNX_OK;
        }
    }
    // host staging: the same chunk ring as the curve fit, with ONE kernel stream -- the plan's device scratch (ATY
    // chunk, M overflow, queue) serves one solve at a time, and in-order launches on one stream guarantee that.
    // The (n_vox, n_bins) coefficient array is 8.4 GB for the C4 volume: its D2H and first-touch faults hide behind
    // the solves of the following chunks.
    std::lock_guard<std::mutex> plan_lock(plan->mu);
    // every chunk is one launch of the solver plus (block kernel) one hand-over pass of ~8 ms: C4 from numpy arrays takes
    // 759 / 712 / 682 / 698 ms with chunks of 256 Ki / 512 Ki / 768 Ki / 1 Mi voxels (profiles/nnls_host_chunk.py) -- beyond
    // 768 Ki the last chunk's download (2 KB per voxel) is what grows
    const size_t chunk = (size_t)dev_env_int("PNX_NNLS_HOST_CHUNK", 3 << 18, 1024, 1 << 22);
    // chunk boundaries.  Block-kernel plans with three or more chunks (their hand-over pass is deferred, see below, so a
    // chunk more costs ~1.5 ms, not 8): the first and the last piece are a quarter chunk -- the first launch starts after a
    // quarter of an upload, and the download left exposed at the end is 0.4 GB instead of 1.6 (2 KB per voxel).
    std::vector<size_t> bounds;
    {
        const size_t ramp = (P.blk && dev_env_int("PNX_HOST_RAMP", 1, 0, 1) && nv >= 3 * chunk) ? chunk / 4 : 0;
        size_t v = 0;
        bounds.push_back(0);
        if (ramp) bounds.push_back(v = ramp);
        const size_t body_end = nv - ramp;
        while (v < body_end) bounds.push_back(v = (body_end - v) < chunk ? body_end : v + chunk);
        if (ramp) bounds.push_back(nv);
    }
    const int n_chunks = (int)bounds.size() - 1;
    const int n_slots = n_chunks < 3 ? n_chunks : 3;
    const size_t cap = nv < chunk ? nv : chunk;
    struct Slot {
        DevBuf slab;
        double *y = nullptr, *c = nullptr, *r = nullptr;  // fp64 working set
        T *ty = nullptr, *tc = nullptr, *tr = nullptr;    // transfer buffers (same memory when T is double)
        int8_t *s = nullptr;
        int32_t *i = nullptr;
    };
    std::vector<Slot> slots(n_slots);
    for (int w = 0; w < n_slots; ++w) {
        Slot &S = slots[w];
        for (int pass = 0; pass < 2; ++pass) {
            Carver c;
            c.base = (char *)S.slab.p;
            auto both = [&](double *&d64, T *&t, size_t count) {
                d64 = (double *)c.take(count * sizeof(double));
                t = F32 ? (T *)c.take(count * sizeof(T)) : (T *)d64;
            };
            both(S.y, S.ty, cap * P.n_meas);
            both(S.c, S.tc, cap * P.n_bins);
            both(S.r, S.tr, cap);
            S.s = (int8_t *)c.take(cap);
            S.i = (int32_t *)c.take(cap * sizeof(int32_t));
            if (pass == 0 && (rc = S.slab.alloc(c.off))) return rc;
        }
    }
    // Block-kernel plans hand a few voxels per chunk to the general kernel, and that pass costs ~8 ms per chunk whatever their
    // number (they are the longest solves there are): with several chunks the hand-over is deferred -- the chunks only
    // collect the voxels' indices and signal rows, ONE pass at the end of the call solves them, and their rows are patched
    // into the caller's arrays (C4 from numpy arrays: seven passes -> one).
    const int defer_cap = dev_env_int("PNX_NNLS_DEFER_CAP", 16384, 0, 1 << 22);
    const bool can_defer = P.blk && n_chunks >= 2 && defer_cap > 0 && nv < ((size_t)1 << 31);
    DevBuf dslab;
    NnlsDefer dctx{};
    double *d_sc = nullptr, *d_sr = nullptr;
    int8_t *d_ss = nullptr;
    int32_t *d_si = nullptr, *d_iota = nullptr;
    struct SideStream {  // the deferred pass runs beside the last chunk's download
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        ~SideStream() {
            if (s) (void)hipStreamDestroy(s);
            if (e) (void)hipEventDestroy(e);
        }
    } side;
    if (can_defer) {
        for (int pass = 0; pass < 2; ++pass) {
            Carver c;
            c.base = (char *)dslab.p;
            dctx.counters = (int32_t *)c.take(2 * sizeof(int32_t));
            dctx.bail = (int32_t *)c.take(nv * sizeof(int32_t));
            dctx.y_side = (double *)c.take((size_t)defer_cap * P.n_meas * sizeof(double));
            // results of the deferred pass, sized for the side buffer's capacity: the pass is launched behind the last chunk's
            // solve -- before the host knows how many voxels it holds -- so that it overlaps that chunk's download
            d_sc = (double *)c.take((size_t)defer_cap * P.n_bins * sizeof(double));
            d_sr = (double *)c.take((size_t)defer_cap * sizeof(double));
            d_ss = (int8_t *)c.take((size_t)defer_cap);
            d_si = (int32_t *)c.take((size_t)defer_cap * sizeof(int32_t));
            d_iota = (int32_t *)c.take((size_t)defer_cap * sizeof(int32_t));
            if (pass == 0 && (rc = dslab.alloc(c.off))) return rc;
        }
        dctx.cap = defer_cap;
        std::vector<int32_t> idx((size_t)defer_cap);
        for (int i = 0; i < defer_cap; ++i) idx[(size_t)i] = i;
        PNX_HIP(hipMemcpy(d_iota, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        {   // lowest priority, like every kernel stream of the host paths: hardware queues are pooled per priority, and a copy
            // must never sit in a queue behind a kernel (the download of the last chunk would wait for this pass)
            int prio_least = 0, prio_greatest = 0;
            PNX_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            PNX_HIP(hipStreamCreateWithPriority(&side.s, hipStreamNonBlocking, prio_least));
        }
        PNX_HIP(hipEventCreateWithFlags(&side.e, hipEventDisableTiming));
    }
    const bool overlap_pass = dev_env_int("PNX_NNLS_DEFER_OVERLAP", 1, 0, 1) != 0;
    auto run = [&](const bool defer) -> int {
        if (defer) PNX_HIP(hipMemset(dctx.counters, 0, 2 * sizeof(int32_t)));
        auto span = [&](int k, size_t &off, size_t &c) {
            off = bounds[(size_t)k];
            c = bounds[(size_t)k + 1] - off;
        };
        PipeOps ops;
        ops.h2d = [&](int k, int slot, hipStream_t s) -> int {
            size_t off, c;
            span(k, off, c);
            PNX_HIP(hipMemcpyAsync(slots[slot].ty, y + off * P.n_meas, c * P.n_meas * sizeof(T), hipMemcpyHostToDevice, s));
            return PNX_OK;
        };
        ops.launch = [&](int k, int slot, hipStream_t s) -> int {
            Slot &S = slots[slot];
            size_t off, c;
            span(k, off, c);
            int r = PNX_OK;
            if constexpr (F32)
                if ((r = cvt(S.ty, S.y, c * P.n_meas, s))) return r;
            if (defer) {
                NnlsDefer d = dctx;
                d.base = (int64_t)off;
                r = nnls_blk_solve_device(&P, (int64_t)c, S.y, max_iter, S.c, S.r, S.s, S.i, s, &d);
                if (!r && k == n_chunks - 1 && overlap_pass) {
                    // every chunk has appended its handed-over voxels: ONE pass of the general kernel over the side buffer, on a
                    // stream of its own behind this chunk's solve, while the chunk's spectra go home (8 ms of download, 9 ms of pass)
                    PNX_HIP(hipEventRecord(side.e, s));
                    PNX_HIP(hipStreamWaitEvent(side.s, side.e, 0));
                    r = nnls_blk_redo_device(&P, defer_cap, dctx.y_side, max_iter, d_sc, d_sr, d_ss, d_si, d_iota, dctx.counters, side.s);
                }
            } else {
                r = nnls_solve_device(&P, (int64_t)c, S.y, max_iter, S.c, S.r, S.s, S.i, s);
            }
            if (r) return r;
            if constexpr (F32)
                if ((r = cvt(S.c, S.tc, c * P.n_bins, s)) || (r = cvt(S.r, S.tr, c, s))) return r;
            return PNX_OK;
        };
        ops.touch = [&](int k) {
            size_t off, c;
            span(k, off, c);
            touch_pages(coeff + off * P.n_bins, c * P.n_bins * sizeof(T));
            touch_pages(rnorm + off, c * sizeof(T));
            if (status) touch_pages(status + off, c);
            if (iters) touch_pages(iters + off, c * sizeof(int32_t));
        };
        ops.d2h = [&](int k, int slot, hipStream_t s) -> int {
            Slot &S = slots[slot];
            size_t off, c;
            span(k, off, c);
            PNX_HIP(hipMemcpyAsync(coeff + off * P.n_bins, S.tc, c * P.n_bins * sizeof(T), hipMemcpyDeviceToHost, s));
            PNX_HIP(hipMemcpyAsync(rnorm + off, S.tr, c * sizeof(T), hipMemcpyDeviceToHost, s));
            if (status) PNX_HIP(hipMemcpyAsync(status + off, S.s, c, hipMemcpyDeviceToHost, s));
            if (iters) PNX_HIP(hipMemcpyAsync(iters + off, S.i, c * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            return PNX_OK;
        };
        HostCallGuard hg;
        int r = run_pipeline(n_chunks, n_slots, 1, hg.touchers(), P.device, st, ops);
        if (r || !defer) return r;
        if (!overlap_pass) {  // PNX_NNLS_DEFER_OVERLAP=0: the pass after the ring has drained (round 3's order)
            int rr = nnls_blk_redo_device(&P, defer_cap, dctx.y_side, max_iter, d_sc, d_sr, d_ss, d_si, d_iota, dctx.counters, side.s);
            if (rr) return rr;
        }
        PNX_HIP(hipStreamSynchronize(side.s));  // the deferred pass (launched behind the last chunk's solve)
        int32_t cnt[2] = {0, 0};
        PNX_HIP(hipMemcpy(cnt, dctx.counters, sizeof(cnt), hipMemcpyDeviceToHost));
        const int n = cnt[0];
        if (n == 0) return PNX_OK;
        if (n < 0 || (size_t)n > nv) return set_error(PNX_ERR_HIP, "deferred hand-over: %d voxels counted in a call of %zu", n, nv);
        // More handed-over voxels than the side buffer holds (stronger regularisers than the reference's: a few per cent of the
        // voxels): the pass behind the last chunk has solved the first defer_cap of them; the others follow in batches of
        // defer_cap, their signal rows gathered from the caller's array.  (Up to round 4 the whole call ran again.)
        std::vector<int32_t> where((size_t)n);
        PNX_HIP(hipMemcpy(where.data(), dctx.bail, where.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            if (where[(size_t)i] < 0 || (size_t)where[(size_t)i] >= nv)
                return set_error(PNX_ERR_HIP, "deferred hand-over: voxel index %d outside the call's %zu voxels", where[(size_t)i], nv);
        const int batch = n < defer_cap ? n : defer_cap;
        std::vector<int32_t> hi((size_t)batch);
        std::vector<double> hc((size_t)batch * P.n_bins), hr((size_t)batch), rows;
        std::vector<int8_t> hs((size_t)batch);
        for (int b0 = 0; b0 < n; b0 += defer_cap) {
            const int nb = (n - b0) < defer_cap ? (n - b0) : defer_cap;
            if (b0 > 0) {
                rows.resize((size_t)nb * P.n_meas);
                for (int i = 0; i < nb; ++i) {
                    const T *src = y + (size_t)where[(size_t)(b0 + i)] * P.n_meas;
                    for (int j = 0; j < P.n_meas; ++j) rows[(size_t)i * P.n_meas + j] = (double)src[j];
                }
                PNX_HIP(hipMemcpy(dctx.y_side, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
                // the list is 0 .. nb - 1 (d_iota) and the count on the device is n >= nb: the kernel stops at the nb it is given
                int rr = nnls_blk_redo_device(&P, nb, dctx.y_side, max_iter, d_sc, d_sr, d_ss, d_si, d_iota, dctx.counters, side.s);
                if (rr) return rr;
                PNX_HIP(hipStreamSynchronize(side.s));
            }
            PNX_HIP(hipMemcpy(hc.data(), d_sc, (size_t)nb * P.n_bins * sizeof(double), hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hr.data(), d_sr, (size_t)nb * sizeof(double), hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hs.data(), d_ss, (size_t)nb, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hi.data(), d_si, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int i = 0; i < nb; ++i) {  // (T) of a double rounds to nearest, as the device's narrowing copy does
                const size_t v = (size_t)where[(size_t)(b0 + i)];
                for (int j = 0; j < P.n_bins; ++j) coeff[v * P.n_bins + j] = (T)hc[(size_t)i * P.n_bins + j];
                rnorm[v] = (T)hr[(size_t)i];
                if (status) status[v] = hs[(size_t)i];
                if (iters) iters[v] = hi[(size_t)i];
            }
        }
        return PNX_OK;
    };
    rc = run(can_defer);
    return rc;
}

extern "C" {
int pnx_nnls_solve_f64(pnx_nnls_plan *plan, int64_t n_vox, const double *y, int max_iter, double *coeff,
                       double *rnorm, int8_t *status, int32_t *iters, int mem, void *stream) {
    return nnls_solve<double>(plan, n_vox, y, max_iter, coeff, rnorm, status, iters, mem, stream);
}

int pnx_nnls_solve_f32(pnx_nnls_plan *plan, int64_t n_vox, const float *y, int max_iter, float *coeff, float *rnorm,
                       int8_t *status, int32_t *iters, int mem, void *stream) {
    return nnls_solve<float>(plan, n_vox, y, max_iter, coeff, rnorm, status, iters, mem, stream);
}

// Host arrays in, peak tables out: the chunk ring of nnls_solve with the peak analysis behind every chunk's solve (the chunk's
// spectra never leave the device) and the same deferred hand-over -- the handed-over voxels' spectra are solved in one pass at
// the end, analysed, and their rows of the peak tables patched on the host.
static int nnls_solve_peaks_host(pnx_nnls_plan *plan, int64_t n_vox, const double *y, int max_iter, const double *bins_host,
                                 double height, int regularized, double rel_height, int max_peaks, int32_t *n_peaks,
                                 double *d_values, double *f_values, int n_cut, const double *cutoffs_host, double *d_cut,
                                 double *f_cut, double *rnorm, int8_t *status, int32_t *iters, hipStream_t st) {
    NnlsPlanData &P = plan->d;
    if (max_peaks > 0 && (!d_values || !f_values)) return set_error(PNX_ERR_INVALID, "d_values / f_values are NULL");
    if (n_cut > 0 && (!cutoffs_host || !d_cut || !f_cut)) return set_error(PNX_ERR_INVALID, "cutoffs / d_cut / f_cut are NULL");
    const size_t nv = (size_t)n_vox;
    const size_t chunk = (size_t)dev_env_int("PNX_NNLS_PEAKS_CHUNK", 3 << 18, 1024, 1 << 22);
    const int n_chunks = (int)((nv + chunk - 1) / chunk);
    const int n_slots = n_chunks < 3 ? n_chunks : 3;
    const size_t cap = nv < chunk ? nv : chunk;
    const size_t mp = (size_t)(max_peaks > 0 ? max_peaks : 1), nc = (size_t)(n_cut > 0 ? n_cut : 1);
    int rc;
    struct Out {  // device buffers of one batch of voxels: solver outputs + peak tables
        double *spec = nullptr, *r = nullptr, *d = nullptr, *f = nullptr, *dc = nullptr, *fc = nullptr;
        int8_t *s = nullptr;
        int32_t *i = nullptr, *np = nullptr;
        void carve(Carver &c, size_t n, int n_bins, size_t mp, size_t nc) {
            spec = (double *)c.take(n * n_bins * 8);
            r = (double *)c.take(n * 8);
            d = (double *)c.take(n * mp * 8);
            f = (double *)c.take(n * mp * 8);
            dc = (double *)c.take(n * nc * 8);
            fc = (double *)c.take(n * nc * 8);
            i = (int32_t *)c.take(n * 4);
            np = (int32_t *)c.take(n * 4);
            s = (int8_t *)c.take(n);
        }
    };
    struct Slot {
        DevBuf slab;
        double *y = nullptr;
        Out o;
    };
    std::vector<Slot> slots((size_t)n_slots);
    for (auto &S : slots)
        for (int pass = 0; pass < 2; ++pass) {
            Carver c;
            c.base = (char *)S.slab.p;
            S.y = (double *)c.take(cap * P.n_meas * 8);
            S.o.carve(c, cap, P.n_bins, mp, nc);
            if (pass == 0 && (rc = S.slab.alloc(c.off))) return rc;
        }
    auto analyse = [&](size_t n, const Out &o, hipStream_t s) -> int {
        return pnx_nnls_spectrum_peaks_f64((int64_t)n, P.n_bins, o.spec, bins_host, height, regularized, rel_height, max_peaks, o.np, o.d,
                                           o.f, n_cut, cutoffs_host, o.dc, o.fc, PNX_MEM_DEVICE, P.device, s);
    };
    const int defer_cap = dev_env_int("PNX_NNLS_DEFER_CAP", 16384, 0, 1 << 22);
    const bool can_defer = P.blk && n_chunks >= 2 && defer_cap > 0 && nv < ((size_t)1 << 31);
    DevBuf dslab;
    NnlsDefer dctx{};
    if (can_defer) {
        for (int pass = 0; pass < 2; ++pass) {
            Carver c;
            c.base = (char *)dslab.p;
            dctx.counters = (int32_t *)c.take(2 * sizeof(int32_t));
            dctx.bail = (int32_t *)c.take(nv * sizeof(int32_t));
            dctx.y_side = (double *)c.take((size_t)defer_cap * P.n_meas * sizeof(double));
            if (pass == 0 && (rc = dslab.alloc(c.off))) return rc;
        }
        dctx.cap = defer_cap;
    }
    auto run = [&](const bool defer) -> int {
        if (defer) PNX_HIP(hipMemset(dctx.counters, 0, 2 * sizeof(int32_t)));
        auto span = [&](int k, size_t &off, size_t &c) {
            off = (size_t)k * chunk;
            c = (nv - off) < chunk ? (nv - off) : chunk;
        };
        PipeOps ops;
        ops.h2d = [&](int k, int slot, hipStream_t s) -> int {
            size_t off, c;
            span(k, off, c);
            PNX_HIP(hipMemcpyAsync(slots[(size_t)slot].y, y + off * P.n_meas, c * P.n_meas * 8, hipMemcpyHostToDevice, s));
            return PNX_OK;
        };
        ops.launch = [&](int k, int slot, hipStream_t s) -> int {
            Slot &S = slots[(size_t)slot];
            size_t off, c;
            span(k, off, c);
            int r;
            if (defer) {
                NnlsDefer d = dctx;
                d.base = (int64_t)off;
                r = nnls_blk_solve_device(&P, (int64_t)c, S.y, max_iter, S.o.spec, S.o.r, S.o.s, S.o.i, s, &d);
            } else {
                r = nnls_solve_device(&P, (int64_t)c, S.y, max_iter, S.o.spec, S.o.r, S.o.s, S.o.i, s);
            }
            return r ? r : analyse(c, S.o, s);
        };
        ops.touch = [&](int k) {
            size_t off, c;
            span(k, off, c);
            touch_pages(rnorm + off, c * 8);
            if (status) touch_pages(status + off, c);
            if (iters) touch_pages(iters + off, c * 4);
            if (n_peaks) touch_pages(n_peaks + off, c * 4);
            if (max_peaks > 0) {
                touch_pages(d_values + off * max_peaks, c * max_peaks * 8);
                touch_pages(f_values + off * max_peaks, c * max_peaks * 8);
            }
            if (n_cut > 0) {
                touch_pages(d_cut + off * n_cut, c * n_cut * 8);
                touch_pages(f_cut + off * n_cut, c * n_cut * 8);
            }
        };
        ops.d2h = [&](int k, int slot, hipStream_t s) -> int {
            const Out &o = slots[(size_t)slot].o;
            size_t off, c;
            span(k, off, c);
            PNX_HIP(hipMemcpyAsync(rnorm + off, o.r, c * 8, hipMemcpyDeviceToHost, s));
            if (status) PNX_HIP(hipMemcpyAsync(status + off, o.s, c, hipMemcpyDeviceToHost, s));
            if (iters) PNX_HIP(hipMemcpyAsync(iters + off, o.i, c * 4, hipMemcpyDeviceToHost, s));
            if (n_peaks) PNX_HIP(hipMemcpyAsync(n_peaks + off, o.np, c * 4, hipMemcpyDeviceToHost, s));
            if (max_peaks > 0) {
                PNX_HIP(hipMemcpyAsync(d_values + off * max_peaks, o.d, c * max_peaks * 8, hipMemcpyDeviceToHost, s));
                PNX_HIP(hipMemcpyAsync(f_values + off * max_peaks, o.f, c * max_peaks * 8, hipMemcpyDeviceToHost, s));
            }
            if (n_cut > 0) {
                PNX_HIP(hipMemcpyAsync(d_cut + off * n_cut, o.dc, c * n_cut * 8, hipMemcpyDeviceToHost, s));
                PNX_HIP(hipMemcpyAsync(f_cut + off * n_cut, o.fc, c * n_cut * 8, hipMemcpyDeviceToHost, s));
            }
            return PNX_OK;
        };
        HostCallGuard hg;
        int r = run_pipeline(n_chunks, n_slots, 1, hg.touchers(), P.device, st, ops);
        if (r || !defer) return r;
        int32_t cnt[2] = {0, 0};
        PNX_HIP(hipMemcpy(cnt, dctx.counters, sizeof(cnt), hipMemcpyDeviceToHost));
        if (cnt[0] == 0) return PNX_OK;
        if (cnt[0] < 0 || (size_t)cnt[0] > nv) return set_error(PNX_ERR_HIP, "deferred hand-over: %d voxels counted in a call of %zu", cnt[0], nv);
        const size_t n_all = (size_t)cnt[0];
        const size_t n = n_all < (size_t)defer_cap ? n_all : (size_t)defer_cap;  // voxels per batch: what the side buffer holds
        DevBuf side;
        Out o;
        int32_t *iota = nullptr;
        for (int pass = 0; pass < 2; ++pass) {
            Carver c;
            c.base = (char *)side.p;
            o.carve(c, n, P.n_bins, mp, nc);
            iota = (int32_t *)c.take(n * 4);
            if (pass == 0 && (r = side.alloc(c.off))) return r;
        }
        std::vector<int32_t> idx(n), where(n_all), hi(n), hn(n);
        for (size_t i = 0; i < n; ++i) idx[i] = (int32_t)i;
        PNX_HIP(hipMemcpy(iota, idx.data(), n * 4, hipMemcpyHostToDevice));
        PNX_HIP(hipMemcpy(where.data(), dctx.bail, n_all * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n_all; ++i)
            if (where[i] < 0 || (size_t)where[i] >= nv)
                return set_error(PNX_ERR_HIP, "deferred hand-over: voxel index %d outside the call's %zu voxels", where[i], nv);
        std::vector<double> hr(n), hd(n * mp), hf(n * mp), hdc(n * nc), hfc(n * nc), rows;
        std::vector<int8_t> hs(n);
        // the side buffer holds the signal rows of the first batch (gathered chunk by chunk on the device); the rows of the
        // later ones come from the caller's array
        for (size_t b0 = 0; b0 < n_all; b0 += n) {
            const size_t nb = (n_all - b0) < n ? (n_all - b0) : n;
            if (b0 > 0) {
                rows.resize(nb * (size_t)P.n_meas);
                for (size_t i = 0; i < nb; ++i) {
                    const double *src = y + (size_t)where[b0 + i] * P.n_meas;
                    for (int j = 0; j < P.n_meas; ++j) rows[i * P.n_meas + j] = src[j];
                }
                PNX_HIP(hipMemcpy(dctx.y_side, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
            }
            if ((r = nnls_blk_redo_device(&P, (int64_t)nb, dctx.y_side, max_iter, o.spec, o.r, o.s, o.i, iota, dctx.counters, st))) return r;
            if ((r = analyse(nb, o, st))) return r;
            PNX_HIP(hipStreamSynchronize(st));
            PNX_HIP(hipMemcpy(hr.data(), o.r, nb * 8, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hs.data(), o.s, nb, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hi.data(), o.i, nb * 4, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hn.data(), o.np, nb * 4, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hd.data(), o.d, nb * mp * 8, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hf.data(), o.f, nb * mp * 8, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hdc.data(), o.dc, nb * nc * 8, hipMemcpyDeviceToHost));
            PNX_HIP(hipMemcpy(hfc.data(), o.fc, nb * nc * 8, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nb; ++i) {
                const size_t v = (size_t)where[b0 + i];
                rnorm[v] = hr[i];
                if (status) status[v] = hs[i];
                if (iters) iters[v] = hi[i];
                if (n_peaks) n_peaks[v] = hn[i];
                for (int j = 0; j < max_peaks; ++j) {
                    d_values[v * max_peaks + j] = hd[i * mp + j];
                    f_values[v * max_peaks + j] = hf[i * mp + j];
                }
                for (int j = 0; j < n_cut; ++j) {
                    d_cut[v * n_cut + j] = hdc[i * nc + j];
                    f_cut[v * n_cut + j] = hfc[i * nc + j];
                }
            }
        }
        return PNX_OK;
    };
    rc = run(can_defer);
    return rc;
}

int pnx_nnls_solve_peaks_f64(pnx_nnls_plan *plan, int64_t n_vox, const double *y, int max_iter, const double *bins_host,
                             double height, int regularized, double rel_height, int max_peaks, int32_t *n_peaks,
                             double *d_values, double *f_values, int n_cut, const double *cutoffs_host, double *d_cut,
                             double *f_cut, double *rnorm, int8_t *status, int32_t *iters, int mem, void *stream) {
    if (!plan) return set_error(PNX_ERR_INVALID, "plan is NULL");
    if (n_vox < 0 || (n_vox && (!y || !rnorm))) return set_error(PNX_ERR_INVALID, "NULL data pointer");
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    if (n_vox == 0) return PNX_OK;
    NnlsPlanData &P = plan->d;
    PNX_HIP(hipSetDevice(P.device));
    hipStream_t st = (hipStream_t)stream;
    if (max_iter <= 0) max_iter = 3 * P.n_bins;
    std::lock_guard<std::mutex> plan_lock(plan->mu);
    if (mem == PNX_MEM_HOST && dev_env_int("PNX_NNLS_PEAKS_RING", 1, 0, 1))
        return nnls_solve_peaks_host(plan, n_vox, y, max_iter, bins_host, height, regularized, rel_height, max_peaks, n_peaks, d_values,
                                     f_values, n_cut, cutoffs_host, d_cut, f_cut, rnorm, status, iters, st);
    // spectra of one chunk live in device scratch only: solve -> peak analysis -> next chunk
    const size_t chunk = (size_t)dev_env_int("PNX_NNLS_PEAKS_CHUNK", 1 << 20, 1024, 1 << 22);
    const size_t nv = (size_t)n_vox, cap = nv < chunk ? nv : chunk;
    DevBuf spec, stage;
    int rc;
    if ((rc = spec.alloc(cap * P.n_bins * sizeof(double)))) return rc;
    const bool host = mem == PNX_MEM_HOST;
    Carver c;
    double *sy = nullptr, *sr = nullptr, *sd = nullptr, *sf = nullptr, *sdc = nullptr, *sfc = nullptr;
    int8_t *ss = nullptr;
    int32_t *si = nullptr, *sn = nullptr;
    if (host) {
        for (int pass = 0; pass < 2; ++pass) {
            c = Carver();
            c.base = (char *)stage.p;
            sy = (double *)c.take(cap * P.n_meas * 8);
            sr = (double *)c.take(cap * 8);
            sd = (double *)c.take(cap * (max_peaks > 0 ? max_peaks : 1) * 8);
            sf = (double *)c.take(cap * (max_peaks > 0 ? max_peaks : 1) * 8);
            sdc = (double *)c.take(cap * (n_cut > 0 ? n_cut : 1) * 8);
            sfc = (double *)c.take(cap * (n_cut > 0 ? n_cut : 1) * 8);
            si = (int32_t *)c.take(cap * 4);
            sn = (int32_t *)c.take(cap * 4);
            ss = (int8_t *)c.take(cap);
            if (pass == 0 && (rc = stage.alloc(c.off))) return rc;
        }
    }
    for (size_t off = 0; off < nv; off += chunk) {
        const size_t n = (nv - off) < chunk ? (nv - off) : chunk;
        const double *yd = y + off * P.n_meas;
        if (host) {
            PNX_HIP(hipMemcpyAsync(sy, yd, n * P.n_meas * 8, hipMemcpyHostToDevice, st));
            yd = sy;
        }
        double *rd = host ? sr : rnorm + off;
        int8_t *std_ = host ? ss : (status ? status + off : nullptr);
        int32_t *itd = host ? si : (iters ? iters + off : nullptr);
        if ((rc = nnls_solve_device(&P, (int64_t)n, yd, max_iter, (double *)spec.p, rd, std_, itd, st))) return rc;
        rc = pnx_nnls_spectrum_peaks_f64((int64_t)n, P.n_bins, (const double *)spec.p, bins_host, height, regularized, rel_height,
                                         max_peaks, host ? sn : (n_peaks ? n_peaks + off : nullptr),
                                         host ? sd : (d_values ? d_values + off * max_peaks : nullptr),
                                         host ? sf : (f_values ? f_values + off * max_peaks : nullptr), n_cut, cutoffs_host,
                                         host ? sdc : (d_cut ? d_cut + off * n_cut : nullptr),
                                         host ? sfc : (f_cut ? f_cut + off * n_cut : nullptr), PNX_MEM_DEVICE, P.device, st);
        if (rc) return rc;
        if (host) {
            PNX_HIP(hipMemcpyAsync(rnorm + off, sr, n * 8, hipMemcpyDeviceToHost, st));
            if (status) PNX_HIP(hipMemcpyAsync(status + off, ss, n, hipMemcpyDeviceToHost, st));
            if (iters) PNX_HIP(hipMemcpyAsync(iters + off, si, n * 4, hipMemcpyDeviceToHost, st));
            if (n_peaks) PNX_HIP(hipMemcpyAsync(n_peaks + off, sn, n * 4, hipMemcpyDeviceToHost, st));
            if (max_peaks > 0) {
                PNX_HIP(hipMemcpyAsync(d_values + off * max_peaks, sd, n * max_peaks * 8, hipMemcpyDeviceToHost, st));
                PNX_HIP(hipMemcpyAsync(f_values + off * max_peaks, sf, n * max_peaks * 8, hipMemcpyDeviceToHost, st));
            }
            if (n_cut > 0) {
                PNX_HIP(hipMemcpyAsync(d_cut + off * n_cut, sdc, n * n_cut * 8, hipMemcpyDeviceToHost, st));
                PNX_HIP(hipMemcpyAsync(f_cut + off * n_cut, sfc, n * n_cut * 8, hipMemcpyDeviceToHost, st));
            }
        }
        // the spectrum scratch is reused by the next chunk: in-order on one stream
    }
    PNX_HIP(hipStreamSynchronize(st));  // the scratch buffers are freed on return
    return PNX_OK;
}

int pnx_nnls_aty_f64(pnx_nnls_plan *plan, int64_t n_vox, const double *y_dev, double *aty_dev, void *stream) {
    if (!plan) return set_error(PNX_ERR_INVALID, "plan is NULL");
    if (n_vox < 0 || (n_vox && !y_dev)) return set_error(PNX_ERR_INVALID, "NULL data pointer");
    if (n_vox == 0) return PNX_OK;
    PNX_HIP(hipSetDevice(plan->d.device));
    return nnls_aty_device(&plan->d, n_vox, y_dev, aty_dev, (hipStream_t)stream);
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

// ---- bulk copies between pageable host arrays and device buffers -----------------------------------------------
// A pageable hipMemcpy is staged by the calling thread (about 10 GB/s of memcpy into the runtime's pinned buffers) and
// a fresh destination array takes its first-touch page faults inside the copy (42 ms per GB).  Here the range is cut into
// 32 MiB pieces handed to a few threads, each with its own stream: staging copies, DMA and page faults overlap.
static int bulk_copy(void *dst, const void *src, size_t bytes, bool to_device, int device, hipStream_t after, int threads) {
    if (!bytes) return PNX_OK;
    if (!dst || !src) return set_error(PNX_ERR_INVALID, "NULL pointer");
    DeviceInfo *dev;
    int rc = get_device(device, &dev);
    if (rc) return rc;
    PNX_HIP(hipSetDevice(device));
    PNX_HIP(hipStreamSynchronize(after));  // the producer of a device source / the last reader of a device destination
    const size_t piece = (size_t)env_int("PNX_COPY_PIECE_MB", 32, 1, 1024) << 20;
    const size_t n_pieces = (bytes + piece - 1) / piece;
    int nt = threads > 0 ? threads : env_int("PNX_COPY_THREADS", 4, 1, 16);
    if ((size_t)nt > n_pieces) nt = (int)n_pieces;
    std::atomic<size_t> next(0);
    std::atomic<int> code(PNX_OK);
    std::mutex mu;
    std::string msg;
    auto work = [&]() {
        hipStream_t st = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            code.store(PNX_ERR_HIP);
            std::lock_guard<std::mutex> lk(mu);
            msg = "bulk copy: stream setup failed";
            return;
        }
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n_pieces || code.load() != PNX_OK) break;
            const size_t off = k * piece, len = (bytes - off) < piece ? (bytes - off) : piece;
            if (!to_device) touch_pages((char *)dst + off, len);
            hipError_t e = hipMemcpyAsync((char *)dst + off, (const char *)src + off, len,
                                          to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) {
                code.store(PNX_ERR_HIP);
                std::lock_guard<std::mutex> lk(mu);
                msg = std::string("bulk copy: ") + hipGetErrorString(e);
                break;
            }
        }
        (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (code.load() != PNX_OK) return set_error(code.load(), "%s", msg.c_str());
    return PNX_OK;
}

int pnx_upload(void *dst_device, const void *src_host, int64_t bytes, int device, void *stream, int threads) {
    if (bytes < 0) return set_error(PNX_ERR_INVALID, "negative byte count");
    return bulk_copy(dst_device, src_host, (size_t)bytes, true, device, (hipStream_t)stream, threads);
}

int pnx_download(void *dst_host, const void *src_device, int64_t bytes, int device, void *stream, int threads) {
    if (bytes < 0) return set_error(PNX_ERR_INVALID, "negative byte count");
    return bulk_copy(dst_host, src_device, (size_t)bytes, false, device, (hipStream_t)stream, threads);
}

}  // extern "C"

// ---- per-label sums of an (n, c) row matrix: the reduction in front of a segmentation-wise fit ---------------------
// Deterministic on purpose (no atomics): wave w owns a contiguous range of rows and adds them, in order, into its own LDS
// table [n_labels][c] (lane k owns column k); a second kernel adds the per-wave tables in wave order.
namespace pnx {
constexpr int kLabelWaves = 2048;
constexpr int kLabelUnroll = 8;

__global__ void __launch_bounds__(64) label_partial_kernel(const double *img, const int32_t *lab, long long n, int c, int n_lab,
                                                           double *part, long long *cnt_part) {
    extern __shared__ double tab[];  // [n_lab][c] doubles, then n_lab counts
    long long *cnt = reinterpret_cast<long long *>(tab + (size_t)n_lab * c);
    const int lane = threadIdx.x;
    for (int e = lane; e < n_lab * c; e += 64) tab[e] = 0.0;
    for (int e = lane; e < n_lab; e += 64) cnt[e] = 0;
    __syncthreads();
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long v0 = (long long)blockIdx.x * per;
    const long long v1 = (v0 + per) < n ? (v0 + per) : n;
    for (int k0 = 0; k0 < c; k0 += 64) {  // one pass per 64 columns (c <= 64: a single pass)
        const int k = k0 + lane;
        const bool on = k < c;
        long long v = v0;
        for (; v + kLabelUnroll <= v1; v += kLabelUnroll) {
            int l[kLabelUnroll];
            double x[kLabelUnroll];
#pragma unroll
            for (int u = 0; u < kLabelUnroll; ++u) {  // the loads of eight rows are in flight together
                l[u] = lab[v + u];
                x[u] = on ? img[(size_t)(v + u) * c + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kLabelUnroll; ++u) {
                if ((unsigned)l[u] < (unsigned)n_lab) {
                    if (on) tab[(size_t)l[u] * c + k] += x[u];
                    if (k0 == 0 && lane == 0) cnt[l[u]] += 1;
                }
            }
        }
        for (; v < v1; ++v) {
            const int l = lab[v];
            if ((unsigned)l < (unsigned)n_lab) {
                if (on) tab[(size_t)l * c + k] += img[(size_t)v * c + k];
                if (k0 == 0 && lane == 0) cnt[l] += 1;
            }
        }
    }
    __syncthreads();
    double *dst = part + (size_t)blockIdx.x * n_lab * c;
    for (int e = lane; e < n_lab * c; e += 64) dst[e] = tab[e];
    for (int e = lane; e < n_lab; e += 64) cnt_part[(size_t)blockIdx.x * n_lab + e] = cnt[e];
}

__global__ void label_reduce_kernel(const double *part, const long long *cnt_part, int waves, int n_lab, int c, double *sums,
                                    long long *counts) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_lab * c) {
        double s = 0.0;
        for (int w = 0; w < waves; ++w) s += part[(size_t)w * n_lab * c + e];
        sums[e] = s;
    }
    if (e < n_lab) {
        long long m = 0;
        for (int w = 0; w < waves; ++w) m += cnt_part[(size_t)w * n_lab + e];
        counts[e] = m;
    }
}
}  // namespace pnx

extern "C" int pnx_label_sums_f64(const double *rows, const int32_t *labels, int64_t n, int c, int n_labels, double *sums,
                                  int64_t *counts, int mem, int device, void *stream) {
    using namespace pnx;
    if (n < 0 || c < 1 || n_labels < 1) return set_error(PNX_ERR_INVALID, "bad label-sum sizes (n=%lld, c=%d, n_labels=%d)", (long long)n, c, n_labels);
    if (!sums || !counts || (n && (!rows || !labels))) return set_error(PNX_ERR_INVALID, "NULL pointer");
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem must be PNX_MEM_HOST or PNX_MEM_DEVICE");
    const size_t lds = ((size_t)n_labels * c + n_labels) * sizeof(double);
    if (lds > 64 * 1024) return set_error(PNX_ERR_UNSUPPORTED, "n_labels * (c + 1) = %zu entries do not fit the 64 KB LDS table", lds / 8);
    DeviceInfo *dev;
    int rc = get_device(device, &dev);
    if (rc) return rc;
    PNX_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const bool host = mem == PNX_MEM_HOST;
    const size_t tab = (size_t)n_labels * c;
    int waves = (int)((n + 63) / 64);
    if (waves > kLabelWaves) waves = kLabelWaves;
    if (waves < 1) waves = 1;
    DevBuf d_rows, d_lab, d_part, d_cnt, d_sums, d_counts;
    if ((rc = d_part.alloc((size_t)waves * tab * sizeof(double))) || (rc = d_cnt.alloc((size_t)waves * n_labels * sizeof(long long)))) return rc;
    const double *rows_d = rows;
    const int32_t *lab_d = labels;
    double *sums_d = sums;
    long long *counts_d = reinterpret_cast<long long *>(counts);
    if (host) {
        if ((rc = d_rows.alloc((size_t)n * c * sizeof(double))) || (rc = d_lab.alloc((size_t)n * sizeof(int32_t))) ||
            (rc = d_sums.alloc(tab * sizeof(double))) || (rc = d_counts.alloc((size_t)n_labels * sizeof(long long))))
            return rc;
        if ((rc = bulk_copy(d_rows.p, rows, (size_t)n * c * sizeof(double), true, device, st, 0))) return rc;
        if ((rc = bulk_copy(d_lab.p, labels, (size_t)n * sizeof(int32_t), true, device, st, 0))) return rc;
        rows_d = (const double *)d_rows.p;
        lab_d = (const int32_t *)d_lab.p;
        sums_d = (double *)d_sums.p;
        counts_d = (long long *)d_counts.p;
    }
    hipLaunchKernelGGL(label_partial_kernel, dim3(waves), dim3(64), lds, st, rows_d, lab_d, (long long)n, c, n_labels,
                       (double *)d_part.p, (long long *)d_cnt.p);
    PNX_HIP(hipGetLastError());
    hipLaunchKernelGGL(label_reduce_kernel, dim3((unsigned)((tab + 255) / 256)), dim3(256), 0, st, (const double *)d_part.p,
                       (const long long *)d_cnt.p, waves, n_labels, c, sums_d, counts_d);
    PNX_HIP(hipGetLastError());
    if (host) {
        PNX_HIP(hipMemcpyAsync(sums, sums_d, tab * sizeof(double), hipMemcpyDeviceToHost, st));
        PNX_HIP(hipMemcpyAsync(counts, counts_d, (size_t)n_labels * sizeof(long long), hipMemcpyDeviceToHost, st));
    }
    PNX_HIP(hipStreamSynchronize(st));  // the scratch tables are freed on return
    return PNX_OK;
}
