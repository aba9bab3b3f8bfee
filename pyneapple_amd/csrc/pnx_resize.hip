// pnx_resize.hip -- separable resize of the first two axes of an (X, Y, C) fp64 array with OpenCV's INTER_LINEAR /
// INTER_CUBIC arithmetic (half-pixel centres, cubic a = -0.75, replicated border, no anti-aliasing), and the
// element-wise step that turns a resized parameter map into the start values and bounds of the next IDEAL level.
// Reference: IDEALFitter._interpolate_array calls cv2.resize per slice and channel (fitters/ideal.py:299-320) and
// derives p0 / bounds per level (ideal.py:167-198).  One thread per output element, channel index fastest (the
// (Z, N) axes are contiguous), so loads and stores are coalesced; HBM bound.
#include <hip/hip_runtime.h>

#include <cmath>

#include <hipcub/hipcub.hpp>

#include "pnx_internal.hpp"

namespace pnx {

struct Taps {  // per output coordinate: 4 source indices and weights (linear: 2 used, rest weight 0)
    int idx[4];
    double w[4];
};

// The taps of output coordinate d, computed where they are used: a handful of flops per output element of an HBM-bound
// kernel -- and no tap table to build on the host, upload and wait for (a device-mode call only enqueues, and can be
// captured into a HIP graph).
__host__ __device__ inline Taps make_taps(int n_src, int n_dst, int method, int d) {
#pragma clang fp contract(off)  // the same roundings as a host evaluation: floor(f) must not depend on an FMA
    const double scale = (double)n_src / n_dst;
    const double f = (d + 0.5) * scale - 0.5;
    int s = (int)floor(f);
    double fr = f - s;
    Taps T;
    for (int k = 0; k < 4; ++k) {
        T.idx[k] = 0;
        T.w[k] = 0;
    }
    if (method == 0) {  // INTER_LINEAR
        if (s < 0) {
            s = 0;
            fr = 0;
        }
        if (s >= n_src - 1) {
            s = n_src - 1;
            fr = 0;
        }
        T.idx[0] = s;
        T.w[0] = 1.0 - fr;
        T.idx[1] = s + 1 < n_src ? s + 1 : n_src - 1;
        T.w[1] = fr;
    } else {  // INTER_CUBIC, A = -0.75
        const double A = -0.75, x = fr;
        const double c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        const double c1 = ((A + 2) * x - (A + 3)) * x * x + 1;
        const double c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        const double c[4] = {c0, c1, c2, 1.0 - c0 - c1 - c2};
        for (int k = 0; k < 4; ++k) {
            int i = s - 1 + k;
            i = i < 0 ? 0 : (i >= n_src ? n_src - 1 : i);
            T.idx[k] = i;
            T.w[k] = c[k];
        }
    }
    return T;
}

__global__ void resize2d_kernel(const double *__restrict__ in, double *__restrict__ out, int X, int Y, long long C, int TX,
                                int TY, int method) {
    const long long total = (long long)TX * TY * C;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long c = e % C;
        const long long xy = e / C;
        const int oy = (int)(xy % TY), ox = (int)(xy / TY);
        const Taps a = make_taps(X, TX, method, ox), b = make_taps(Y, TY, method, oy);
        // rows first, then columns -- the order of the numpy restatement (ideal.py resize2d: Wx, then Wy)
        double acc = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double col = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) col += a.w[i] * in[((long long)a.idx[i] * Y + b.idx[j]) * C + c];
            acc += b.w[j] * col;
        }
        out[e] = acc;
    }
}

struct BoundsArgs {  // by value: nothing to upload, nothing to wait for
    double lo[PNX_MAX_PARAMS], hi[PNX_MAX_PARAMS], tol[PNX_MAX_PARAMS];
};

// p0 = clip(map, lo, hi); lower = clip(p0 (1 - tol), lo, hi); upper = clip(p0 (1 + tol), lo, hi), written
// parameter-major (n_params, n_px) -- the layout the solver takes (ideal.py:182-189, validation.py:177-203).
// `map` is (n_px, n_params) (a parameter map with the parameter axis last).
__global__ void ideal_bounds_kernel(const double *__restrict__ map, long long n_px, int n_params, const BoundsArgs B,
                                    double *__restrict__ p0, double *__restrict__ lower, double *__restrict__ upper) {
    const long long total = n_px * n_params;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(e / n_px);
        const long long v = e - (long long)k * n_px;
        const double l = B.lo[k], h = B.hi[k], t = B.tol[k];
        double p = map[v * n_params + k];
        p = fmin(fmax(p, l), h);
        p0[e] = p;
        lower[e] = fmin(fmax(p * (1 - t), l), h);
        upper[e] = fmin(fmax(p * (1 + t), l), h);
    }
}

// ---- level plumbing between the resize and the fit (IDEALFitter.fit, fitters/ideal.py:199-254): which voxels of the level
// are fitted, their signal rows and start values gathered, the estimates scattered back into the level's map.
struct MaskAbove {  // idx -> mask[idx] > thr   (ideal.py:199: `_segmentation_interp[..., 0] > self.segmentation_threshold`)
    const double *mask;
    double thr;
    __host__ __device__ bool operator()(const long long &i) const { return mask[i] > thr; }
};

__global__ void gather_rows_kernel(const double *__restrict__ src, const long long *__restrict__ idx, long long n_sel, int c,
                                   double *__restrict__ dst) {
    const long long total = n_sel * c;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / c;
        const int j = (int)(e - i * c);
        dst[e] = src[idx[i] * c + j];
    }
}

// pmap (n_total, k) zero filled before; pmap[idx[i]][j] = popt[j][i]  (idx == nullptr: identity)
__global__ void scatter_rows_t_kernel(const double *__restrict__ popt, const long long *__restrict__ idx, long long n_sel, int k,
                                      double *__restrict__ pmap) {
    const long long total = n_sel * k;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e / n_sel);
        const long long i = e - (long long)j * n_sel;  // consecutive threads: consecutive voxels of one parameter row
        pmap[(idx ? idx[i] : i) * k + j] = popt[e];
    }
}

// sum_i (y_i - mean(y))^2 per row (fitters/base.py:179-181: SS_tot of R^2); one lane per row, rows are short (n_b values)
__global__ void row_ss_tot_kernel(const double *__restrict__ y, long long n, int c, double *__restrict__ out) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const double *row = y + r * c;
        double m = 0;
        for (int j = 0; j < c; ++j) m += row[j];
        m /= c;
        double s = 0;
        for (int j = 0; j < c; ++j) {
            const double d = row[j] - m;
            s = fma(d, d, s);
        }
        out[r] = s;
    }
}

}  // namespace pnx

using namespace pnx;

#define RS_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) return set_error(PNX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

extern "C" {

int pnx_resize2d_f64(const double *in, int X, int Y, int64_t C, double *out, int TX, int TY, int method, int mem, int device,
                     void *stream) {
    if (!in || !out) return set_error(PNX_ERR_INVALID, "NULL pointer");
    if (X < 1 || Y < 1 || C < 1 || TX < 1 || TY < 1) return set_error(PNX_ERR_INVALID, "bad resize shape");
    if (method != 0 && method != 1) return set_error(PNX_ERR_INVALID, "method %d (0 linear, 1 cubic)", method);
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(PNX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_error(PNX_ERR_INVALID, "device %d out of range", device);
    RS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t n_in = (size_t)X * Y * C, n_out = (size_t)TX * TY * C;
    const double *din = in;
    double *dout = out;
    struct Tmp {  // freed on every return path
        void *p = nullptr;
        ~Tmp() {
            if (p) (void)hipFree(p);
        }
    } tmp_in, tmp_out;
    if (mem == PNX_MEM_HOST) {
        RS_HIP(hipMalloc(&tmp_in.p, n_in * sizeof(double)));
        RS_HIP(hipMalloc(&tmp_out.p, n_out * sizeof(double)));
        RS_HIP(hipMemcpyAsync(tmp_in.p, in, n_in * sizeof(double), hipMemcpyHostToDevice, st));
        din = (const double *)tmp_in.p;
        dout = (double *)tmp_out.p;
    }
    size_t blocks = (n_out + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(resize2d_kernel, dim3((unsigned)blocks), dim3(256), 0, st, din, dout, X, Y, (long long)C, TX, TY, method);
    RS_HIP(hipGetLastError());
    if (mem == PNX_MEM_HOST) {
        RS_HIP(hipMemcpyAsync(out, dout, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
        RS_HIP(hipStreamSynchronize(st));
    }
    return PNX_OK;  // PNX_MEM_DEVICE: enqueued only
}

int pnx_ideal_bounds_f64(const double *map, int64_t n_px, int n_params, const double *lo_host, const double *hi_host,
                         const double *tol_host, double *p0, double *lower, double *upper, int device, void *stream) {
    if (!map || !lo_host || !hi_host || !tol_host || !p0 || !lower || !upper) return set_error(PNX_ERR_INVALID, "NULL pointer");
    if (n_px < 0 || n_params < 1 || n_params > PNX_MAX_PARAMS) return set_error(PNX_ERR_INVALID, "bad sizes");
    if (n_px == 0) return PNX_OK;
    RS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    BoundsArgs B;
    for (int k = 0; k < PNX_MAX_PARAMS; ++k) {
        B.lo[k] = k < n_params ? lo_host[k] : 0.0;
        B.hi[k] = k < n_params ? hi_host[k] : 0.0;
        B.tol[k] = k < n_params ? tol_host[k] : 0.0;
    }
    size_t blocks = ((size_t)n_px * n_params + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ideal_bounds_kernel, dim3((unsigned)blocks), dim3(256), 0, st, map, (long long)n_px, n_params, B, p0,
                       lower, upper);
    RS_HIP(hipGetLastError());
    return PNX_OK;
}

/* indices (C order) of the entries of `mask` (n values, device) that exceed `threshold`, written to idx (n int64, device);
 * returns their number in *n_selected (host).  Synchronises `stream`: the caller sizes the level's arrays with the count. */
int pnx_mask_select_f64(const double *mask, int64_t n, double threshold, int64_t *idx, int64_t *n_selected, int device,
                        void *stream) {
    if (!mask || !idx || !n_selected || n < 0) return set_error(PNX_ERR_INVALID, "bad argument");
    *n_selected = 0;
    if (n == 0) return PNX_OK;
    RS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    struct Tmp {
        void *p = nullptr;
        ~Tmp() {
            if (p) (void)hipFree(p);
        }
    } tmp, cnt;
    RS_HIP(hipMalloc(&cnt.p, sizeof(long long)));
    MaskAbove pred{mask, threshold};
    size_t bytes = 0;
    // hipCUB takes an int count: levels beyond 2^31 voxels are cut into pieces
    long long total = 0;
    const int64_t piece = (int64_t)1 << 30;
    for (int64_t off = 0; off < n; off += piece) {
        const int m = (int)((n - off) < piece ? (n - off) : piece);
        hipcub::CountingInputIterator<long long> it(off);
        size_t need = 0;
        RS_HIP(hipcub::DeviceSelect::If(nullptr, need, it, (long long *)idx + total, (long long *)cnt.p, m, pred, st));
        if (!tmp.p || need > bytes) {
            if (tmp.p) (void)hipFree(tmp.p);
            tmp.p = nullptr;
            RS_HIP(hipMalloc(&tmp.p, need ? need : 8));
            bytes = need;
        }
        RS_HIP(hipcub::DeviceSelect::If(tmp.p, need, it, (long long *)idx + total, (long long *)cnt.p, m, pred, st));
        long long c = 0;
        RS_HIP(hipMemcpyAsync(&c, cnt.p, sizeof(c), hipMemcpyDeviceToHost, st));
        RS_HIP(hipStreamSynchronize(st));
        total += c;
    }
    *n_selected = total;
    return PNX_OK;
}

/* dst (n_sel, c) = src[idx, :]  (src (n, c) row-major, idx (n_sel) int64; all device; enqueued only) */
int pnx_gather_rows_f64(const double *src, int64_t c, const int64_t *idx, int64_t n_sel, double *dst, int device, void *stream) {
    if (!src || !idx || !dst || c < 1 || c > (1 << 20) || n_sel < 0) return set_error(PNX_ERR_INVALID, "bad argument");
    if (n_sel == 0) return PNX_OK;
    RS_HIP(hipSetDevice(device));
    size_t blocks = ((size_t)n_sel * c + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (const long long *)idx,
                       (long long)n_sel, (int)c, dst);
    RS_HIP(hipGetLastError());
    return PNX_OK;
}

/* Parameter map of a level from the solver's estimates: pmap (n_total, k) = 0, then pmap[idx[i], j] = popt[j, i] for the n_sel
 * fitted voxels (idx NULL: all voxels in order).  popt (k, n_sel) parameter major.  All device; enqueued only.
 * (fitters/ideal.py:243-252: `param_map[xs, ys, zs, param_idx] = values`) */
int pnx_scatter_rows_t_f64(const double *popt, const int64_t *idx, int64_t n_sel, int k, int64_t n_total, double *pmap, int device,
                           void *stream) {
    if (!popt || !pmap || k < 1 || n_sel < 0 || n_total < n_sel) return set_error(PNX_ERR_INVALID, "bad argument");
    RS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    if (idx || n_sel < n_total) RS_HIP(hipMemsetAsync(pmap, 0, (size_t)n_total * k * sizeof(double), st));
    if (n_sel == 0) return PNX_OK;
    size_t blocks = ((size_t)n_sel * k + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(scatter_rows_t_kernel, dim3((unsigned)blocks), dim3(256), 0, st, popt, (const long long *)idx, (long long)n_sel, k,
                       pmap);
    RS_HIP(hipGetLastError());
    return PNX_OK;
}

/* out (n) = sum_j (y[i, j] - mean_j y[i, :])^2: SS_tot of R^2 (fitters/base.py:179-183).  Device pointers; enqueued only. */
int pnx_row_ss_tot_f64(const double *y, int64_t n, int c, double *out, int device, void *stream) {
    if (!y || !out || n < 0 || c < 1) return set_error(PNX_ERR_INVALID, "bad argument");
    if (n == 0) return PNX_OK;
    RS_HIP(hipSetDevice(device));
    size_t blocks = ((size_t)n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(row_ss_tot_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, (long long)n, c, out);
    RS_HIP(hipGetLastError());
    return PNX_OK;
}
}


// ---- queue order of a curve fit from a predictor of the evaluation counts (pnx_curvefit_queue_order) ------------
namespace pnx {
namespace {
__global__ void iota_kernel(int32_t *v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
}  // namespace
}  // namespace pnx

extern "C" int pnx_queue_order_f64(const double *key, int64_t n, int32_t *order, int device, void *stream) {
    using namespace pnx;
    if (!key || !order || n < 0) return set_error(PNX_ERR_INVALID, "bad argument");
    if (n == 0) return PNX_OK;
    if (n >= ((int64_t)1 << 31)) return set_error(PNX_ERR_INVALID, "n=%lld: at most 2^31 - 1 voxels", (long long)n);
    RS_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    struct Tmp {
        void *p = nullptr;
        ~Tmp() {
            if (p) (void)hipFree(p);
        }
    } tmp, keys_out, iota;
    RS_HIP(hipMalloc(&keys_out.p, (size_t)n * sizeof(double)));
    RS_HIP(hipMalloc(&iota.p, (size_t)n * sizeof(int32_t)));
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (int32_t *)iota.p, (int)n);
    RS_HIP(hipGetLastError());
    size_t need = 0;
    RS_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, need, key, (double *)keys_out.p, (const int32_t *)iota.p, order, (int)n, 0, 64, st));
    RS_HIP(hipMalloc(&tmp.p, need ? need : 8));
    // a radix sort is stable: voxels with equal keys keep their index order
    RS_HIP(hipcub::DeviceRadixSort::SortPairsDescending(tmp.p, need, key, (double *)keys_out.p, (const int32_t *)iota.p, order, (int)n, 0, 64, st));
    RS_HIP(hipStreamSynchronize(st));  // the temporaries are freed on return
    return PNX_OK;
}
