// pnx_spectrum.hip -- NNLS spectrum post-processing and parameter-map layout on the device.
//
// Replaces, for all voxels at once:
//   pyneapple.utility.spectrum.find_spectrum_peaks / calculate_peak_area / apply_cutoffs / geometric_mean_peak
//       (reference src/pyneapple/utility/spectrum.py:13-215), which call scipy.signal.find_peaks(height=...) and
//       scipy.signal.peak_widths(rel_height) per voxel;
//   pyneapple.io.nifti.reconstruct_maps (src/pyneapple/io/nifti.py:279-312): float32 (X, Y, Z[, k]) volumes, zero outside
//       the fitted voxels.
// With them a 250-bin NNLS fit returns O(10) numbers per voxel instead of a 2 KB spectrum (8.4 GB for the C4 volume).
//
// scipy.signal's kernels are compiled (no source on disk); what is restated is the published algorithm of SciPy 1.15:
// _local_maxima_1d (plateaus: midpoint of the flat top), the height condition, _peak_prominences (wlen = None) and
// _peak_widths with linear interpolation of the crossing points.  Pinned by fixtures generated from the reference
// (tests/golden/g9_spectrum_*).
//
// Mapping: HBM-bound compare / ballot work.  One wavefront owns one spectrum, four bins per lane: the row arrives with four
// coalesced 512-byte loads, neighbour samples come from a 2 KB LDS copy, and SciPy's sequential scans become wave ballots
// (nearest higher sample = highest / lowest set bit of a comparison mask, bases = masked wave minimum + ballot of the
// equal samples, crossing points = highest / lowest set bit of `x <= height`).  The peak table of a voxel lives in lanes
// 0..15.  A first version (one lane per spectrum walking an LDS tile of 64 spectra) ran at 101 GB/s: 1 500 dependent LDS
// reads per lane with one wave per CU; spectra with a flat-topped rise still take that sequential path (one lane).
#include <hip/hip_runtime.h>

#include <cmath>

#include "pnx_internal.hpp"

namespace pnx {
namespace {
constexpr int kW = 64;
constexpr int kNarrowBins = 256;  // four bins per lane, the bin centres travel in the kernel arguments
constexpr int kMaxBins = 512;     // eight bins per lane (the wide NNLS plans), the bin centres in a device buffer
constexpr int kMaxPeaks = 64;  // per voxel: peak k of the list lives in lane k of the wave that owns the spectrum
constexpr int kSeqPeaks = 16;  // the one-lane path of flat-topped spectra keeps its list in registers: 16 (NaN rows beyond)
constexpr int kMaxCut = 8;

struct PeakArgs {
    const double *spec;  // (n_vox, n_bins)
    long long n_vox;
    int n_bins, max_peaks, n_cut, regularized;
    double height, rel_height;
    double bins[kNarrowBins];  // by value: no device allocation, nothing to wait for in device mode
    const double *bins_wide;   // more than 256 bins (a kernel's arguments end at 4 KB): device copy, stream ordered
    __device__ double bin(int j) const { return bins_wide ? bins_wide[j] : bins[j]; }
    int32_t *n_peaks;    // (n_vox)
    double *d_values;    // (n_vox, max_peaks) NaN padded, may be null
    double *f_values;    // (n_vox, max_peaks)
    double cut[2 * kMaxCut];
    double *d_cut;  // (n_vox, n_cut), may be null
    double *f_cut;
};

// One voxel, one lane, SciPy's loops as they are written (plateaus included).  The wave-parallel kernel below falls back
// to this for the rare spectrum with a flat-topped rise (equal neighbouring samples after a rise), where the midpoint rule
// needs the sequential scan.
__device__ void peaks_sequential(const PeakArgs &A, const double *x, long long vox) {
    const int n = A.n_bins;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    // ---- scipy.signal._peak_finding_utils._local_maxima_1d + the height condition (hmin <= x[peak])
    int pk[kSeqPeaks];
    int m = 0, total = 0;
    {
        int i = 1;
        const int i_max = n - 1;
        while (i < i_max) {
            if (x[i - 1] < x[i]) {
                int ia = i + 1;
                while (ia < i_max && x[ia] == x[i]) ++ia;
                if (x[ia] < x[i]) {
                    const int mid = (i + ia - 1) / 2;
                    if (A.height <= x[mid]) {
                        if (m < kSeqPeaks) {
#pragma unroll
                            for (int k = 0; k < kSeqPeaks; ++k)
                                if (k == m) pk[k] = mid;  // static indexing: the list stays in registers
                            ++m;
                        }
                        ++total;
                    }
                    i = ia;
                }
            }
            ++i;
        }
    }
    if (total > kSeqPeaks) {  // more peaks than the table holds: NaN everywhere rather than fractions normalised over a part
        if (A.n_peaks) A.n_peaks[vox] = total;
        if (A.d_values)
            for (int k = 0; k < A.max_peaks; ++k) A.d_values[(size_t)vox * A.max_peaks + k] = A.f_values[(size_t)vox * A.max_peaks + k] = nan;
        if (A.d_cut)
            for (int c = 0; c < A.n_cut; ++c) A.d_cut[(size_t)vox * A.n_cut + c] = A.f_cut[(size_t)vox * A.n_cut + c] = nan;
        return;
    }
    // ---- fractions: raw heights, or the Gaussian area from the width at rel_height of the prominence
    double fv[kSeqPeaks], dv[kSeqPeaks];
    double fsum = 0;
#pragma unroll
    for (int k = 0; k < kSeqPeaks; ++k) {
        fv[k] = nan;
        dv[k] = nan;
        if (k < m) {
            const int peak = pk[k];
            const double xp = x[peak];
            double f = xp;
            if (A.regularized) {
                // _peak_prominences, wlen = None
                int i = peak, lb = peak, rb = peak;
                double lmin = xp, rmin = xp;
                while (0 <= i && x[i] <= xp) {
                    if (x[i] < lmin) {
                        lmin = x[i];
                        lb = i;
                    }
                    --i;
                }
                i = peak;
                while (i <= n - 1 && x[i] <= xp) {
                    if (x[i] < rmin) {
                        rmin = x[i];
                        rb = i;
                    }
                    ++i;
                }
                const double prom = xp - fmax(lmin, rmin);
                // _peak_widths
                const double h = xp - prom * A.rel_height;
                i = peak;
                while (lb < i && h < x[i]) --i;
                double lip = (double)i;
                if (x[i] < h) lip += (h - x[i]) / (x[i + 1] - x[i]);
                i = peak;
                while (i < rb && h < x[i]) ++i;
                double rip = (double)i;
                if (x[i] < h) rip -= (h - x[i]) / (x[i - 1] - x[i]);
                const double fwhm = rip - lip;
                // spectrum.py:44-47: height * fwhm / (2 sqrt(2 ln 2)) * sqrt(2 pi)
                f = xp * fwhm / (2.0 * sqrt(2.0 * log(2.0))) * sqrt(2.0 * 3.141592653589793);
            }
            fv[k] = f;
            dv[k] = A.bin(peak);
            fsum += f;
        }
    }
    if (fsum > 0) {
#pragma unroll
        for (int k = 0; k < kSeqPeaks; ++k)
            if (k < m) fv[k] = fv[k] / fsum;
    }
    if (A.n_peaks) A.n_peaks[vox] = total;
    if (A.d_values) {
        for (int k = 0; k < A.max_peaks; ++k) {
            double d = nan, f = nan;
#pragma unroll
            for (int j = 0; j < kSeqPeaks; ++j)
                if (j == k) {
                    d = dv[j];
                    f = fv[j];
                }
            A.d_values[(size_t)vox * A.max_peaks + k] = d;
            A.f_values[(size_t)vox * A.max_peaks + k] = f;
        }
    }
    // ---- apply_cutoffs (spectrum.py:142-215): none -> NaN, one -> as is, several -> geometric_mean_peak
    if (A.d_cut) {
        double dc[kMaxCut], fc[kMaxCut];
        double tot = 0;
        bool any = false;
#pragma unroll
        for (int c = 0; c < kMaxCut; ++c) {
            dc[c] = nan;
            fc[c] = nan;
            if (c < A.n_cut) {
                const double lo = A.cut[2 * c], hi = A.cut[2 * c + 1];
                int cnt = 0;
                double hs = 0, d1 = 0, f1 = 0;
#pragma unroll
                for (int k = 0; k < kSeqPeaks; ++k)
                    if (k < m && dv[k] >= lo && dv[k] <= hi) {
                        ++cnt;
                        hs += fv[k];
                        d1 = dv[k];
                        f1 = fv[k];
                    }
                if (cnt == 1) {
                    dc[c] = d1;
                    fc[c] = f1;
                } else if (cnt > 1) {
                    // geometric_mean_peak (spectrum.py:106-139): log10(prod(pos ** (h / sum h))), sum h
                    double prod = 1.0;
#pragma unroll
                    for (int k = 0; k < kSeqPeaks; ++k)
                        if (k < m && dv[k] >= lo && dv[k] <= hi) prod *= pow(dv[k], fv[k] / hs);
                    dc[c] = log10(prod);
                    fc[c] = hs;
                }
                if (cnt > 0) {
                    tot += fc[c];
                    any = true;
                }
            }
        }
        for (int c = 0; c < A.n_cut; ++c) {
            double d = nan, f = nan;
#pragma unroll
            for (int j = 0; j < kMaxCut; ++j)
                if (j == c) {
                    d = dc[j];
                    f = fc[j];
                }
            if (any && tot > 0 && !isnan(f)) f = f / tot;
            A.d_cut[(size_t)vox * A.n_cut + c] = d;
            A.f_cut[(size_t)vox * A.n_cut + c] = f;
        }
    }
}

// ---- wave-parallel version: one wavefront owns one spectrum, four (eight) bins per lane (bin = 64 s + lane) ---------------
__device__ inline double wmin(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ inline double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int NS> struct MaskN {  // one bit per bin
    unsigned long long m[NS];
};
template <int NS> __device__ inline MaskN<NS> ballot4(const bool (&c)[NS]) {
    MaskN<NS> r;
#pragma unroll
    for (int s = 0; s < NS; ++s) r.m[s] = __ballot(c[s] ? 1 : 0);
    return r;
}
template <int NS> __device__ inline int highest(const MaskN<NS> &k) {  // index of the highest set bit, -1 if none
#pragma unroll
    for (int s = NS - 1; s >= 0; --s)
        if (k.m[s]) return 64 * s + 63 - __builtin_clzll(k.m[s]);
    return -1;
}
template <int NS> __device__ inline int lowest(const MaskN<NS> &k, int none) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (k.m[s]) return 64 * s + __builtin_ctzll(k.m[s]);
    return none;
}

// NS: bins per lane (4 up to 256 bins, 8 up to 512)
template <int NS> __global__ void __launch_bounds__(256) spectrum_peaks_kernel(const PeakArgs A) {
    __shared__ double rows[4][64 * NS + 8];  // per wave: the spectrum, one pad sample on either side
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = A.n_bins;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    double *xr = rows[wave] + 1;  // xr[-1] and xr[n] exist
    for (long long vox = (long long)blockIdx.x * 4 + wave; vox < A.n_vox; vox += (long long)gridDim.x * 4) {
        const double *src = A.spec + (size_t)vox * n;
        double v[NS];
        int j[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            j[s] = 64 * s + lane;
            v[s] = j[s] < n ? src[j[s]] : 0.0;  // coalesced: 512 bytes per instruction
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < NS; ++s) xr[j[s]] = v[s];
        if (lane == 0) {
            xr[-1] = 0.0;
            xr[n] = 0.0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wave: its LDS instructions execute in order
        // ---- _local_maxima_1d without plateaus + the height condition
        bool pk[NS], flat[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool in = j[s] >= 1 && j[s] <= n - 2;
            const double xm = xr[j[s] - 1], xp = in ? xr[j[s] + 1] : 0.0;
            const bool rise = in && xm < v[s];
            pk[s] = rise && v[s] > xp && A.height <= v[s];
            flat[s] = rise && v[s] == xp;
        }
        bool anyflat = false;
#pragma unroll
        for (int s = 0; s < NS; ++s) anyflat = anyflat || flat[s];
        if (__any(anyflat ? 1 : 0)) {
            if (lane == 0) peaks_sequential(A, xr, vox);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            continue;
        }
        MaskN<NS> peaks = ballot4(pk);
        int total = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) total += __popcll(peaks.m[s]);
        if (total > kMaxPeaks) {  // more peaks than the table (one per lane) holds: NaN, never a partial normalisation
            if (A.n_peaks && lane == 0) A.n_peaks[vox] = total;
            if (A.d_values && lane < A.max_peaks) A.d_values[(size_t)vox * A.max_peaks + lane] = A.f_values[(size_t)vox * A.max_peaks + lane] = nan;
            if (A.d_cut && lane < A.n_cut) A.d_cut[(size_t)vox * A.n_cut + lane] = A.f_cut[(size_t)vox * A.n_cut + lane] = nan;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            continue;
        }
        // peak k of the list lives in lane k
        double dval = nan, fval = nan;
        int m = 0;
        for (int s = 0; s < NS; ++s) {
            unsigned long long bits = peaks.m[s];
            while (bits && m < kMaxPeaks) {
                const int p = 64 * s + __builtin_ctzll(bits);
                bits &= bits - 1;
                const double xp = xr[p];
                double f = xp;
                if (A.regularized) {
                    // _peak_prominences (wlen = None): nearest higher sample on either side bounds the scan; the base is the
                    // lowest sample in between, the one closest to the peak among equals
                    bool c[NS];
#pragma unroll
                    for (int t = 0; t < NS; ++t) c[t] = j[t] < p && v[t] > xp;
                    const int L = highest(ballot4(c));
#pragma unroll
                    for (int t = 0; t < NS; ++t) c[t] = j[t] > p && j[t] < n && v[t] > xp;
                    const int R = lowest(ballot4(c), n);
                    double a = INFINITY, b = INFINITY;
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        if (j[t] > L && j[t] <= p) a = fmin(a, v[t]);
                        if (j[t] >= p && j[t] < R) b = fmin(b, v[t]);
                    }
                    const double lmin = wmin(a), rmin = wmin(b);
#pragma unroll
                    for (int t = 0; t < NS; ++t) c[t] = j[t] > L && j[t] <= p && v[t] == lmin;
                    const int lb = highest(ballot4(c));
#pragma unroll
                    for (int t = 0; t < NS; ++t) c[t] = j[t] >= p && j[t] < R && v[t] == rmin;
                    const int rb = lowest(ballot4(c), p);
                    const double prom = xp - fmax(lmin, rmin);
                    // _peak_widths: first sample at or below the evaluation height on either side, not beyond the bases
                    const double h = xp - prom * A.rel_height;
#pragma unroll
                    for (int t = 0; t < NS; ++t) c[t] = j[t] > lb && j[t] <= p && !(h < v[t]);
                    int li = highest(ballot4(c));
                    if (li < 0) li = lb;
#pragma unroll
                    for (int t = 0; t < NS; ++t) c[t] = j[t] >= p && j[t] < rb && !(h < v[t]);
                    const int ri = lowest(ballot4(c), rb);
                    double lip = (double)li, rip = (double)ri;
                    const double xl = xr[li], xrr = xr[ri];
                    if (xl < h) lip += (h - xl) / (xr[li + 1] - xl);
                    if (xrr < h) rip -= (h - xrr) / (xr[ri - 1] - xrr);
                    const double fwhm = rip - lip;
                    f = xp * fwhm / (2.0 * sqrt(2.0 * log(2.0))) * sqrt(2.0 * 3.141592653589793);  // spectrum.py:44-47
                }
                if (lane == m) {
                    dval = A.bin(p);
                    fval = f;
                }
                ++m;
            }
        }
        {   // np.sum of the fractions in peak order, then the normalisation (spectrum.py:96-99)
            double fsum = 0;
            for (int k = 0; k < m; ++k) fsum += __shfl(fval, k);
            if (fsum > 0 && lane < m) fval = fval / fsum;
        }
        if (A.n_peaks && lane == 0) A.n_peaks[vox] = total;
        if (A.d_values && lane < A.max_peaks) {
            A.d_values[(size_t)vox * A.max_peaks + lane] = dval;
            A.f_values[(size_t)vox * A.max_peaks + lane] = fval;
        }
        // ---- apply_cutoffs (spectrum.py:142-215)
        if (A.d_cut) {
            double dc = nan, fc = nan;  // range c lives in lane c
            double tot = 0;
            bool any = false;
            for (int c = 0; c < A.n_cut; ++c) {
                const double lo = A.cut[2 * c], hi = A.cut[2 * c + 1];
                const bool in = lane < m && dval >= lo && dval <= hi;
                const unsigned long long bits = __ballot(in ? 1 : 0);
                const int cnt = __popcll(bits);
                double d = nan, f = nan;
                if (cnt == 1) {
                    const int k = __builtin_ctzll(bits);
                    d = __shfl(dval, k);
                    f = __shfl(fval, k);
                } else if (cnt > 1) {
                    // geometric_mean_peak (spectrum.py:106-139): sums and the product run over the peaks in order
                    double hs = 0;
                    for (unsigned long long b2 = bits; b2; b2 &= b2 - 1) hs += __shfl(fval, __builtin_ctzll(b2));
                    double prod = 1.0;
                    for (unsigned long long b2 = bits; b2; b2 &= b2 - 1) {
                        const int k = __builtin_ctzll(b2);
                        prod *= pow(__shfl(dval, k), __shfl(fval, k) / hs);
                    }
                    d = log10(prod);
                    f = hs;
                }
                if (cnt > 0) {
                    tot += f;
                    any = true;
                }
                if (lane == c) {
                    dc = d;
                    fc = f;
                }
            }
            if (any && tot > 0 && !isnan(fc)) fc = fc / tot;
            if (lane < A.n_cut) {
                A.d_cut[(size_t)vox * A.n_cut + lane] = dc;
                A.f_cut[(size_t)vox * A.n_cut + lane] = fc;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the row buffer is rewritten by the next voxel
    }
}

// out[(idx[i]) * k + c] = (float) values[i * k + c]; `out` was zero-filled
__global__ void scatter_maps_kernel(const double *values, const long long *idx, long long n_px, int k, float *out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_px * k) return;
    const long long i = e / k;
    const int c = (int)(e - i * k);
    out[(size_t)idx[i] * k + c] = (float)values[e];
}

struct DevTmp {
    void *p = nullptr;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
};
}  // namespace
}  // namespace pnx

#define PNX_HIPS(call)                                                                                    \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess) return pnx::set_error(PNX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

extern "C" {

int pnx_nnls_spectrum_peaks_f64(int64_t n_vox, int n_bins, const double *spectrum, const double *bins_host, double height,
                                int regularized, double rel_height, int max_peaks, int32_t *n_peaks, double *d_values,
                                double *f_values, int n_cut, const double *cutoffs_host, double *d_cut, double *f_cut, int mem,
                                int device, void *stream) {
    using namespace pnx;
    if (n_vox < 0 || n_bins < 3 || n_bins > kMaxBins) return set_error(PNX_ERR_INVALID, "n_vox=%lld, n_bins=%d (3..%d)", (long long)n_vox, n_bins, kMaxBins);
    if (max_peaks < 0 || max_peaks > kMaxPeaks) return set_error(PNX_ERR_INVALID, "max_peaks=%d (0..%d)", max_peaks, kMaxPeaks);
    if (n_cut < 0 || n_cut > kMaxCut) return set_error(PNX_ERR_INVALID, "n_cut=%d (0..%d)", n_cut, kMaxCut);
    if (!bins_host || (n_vox && !spectrum)) return set_error(PNX_ERR_INVALID, "NULL pointer");
    if (max_peaks && (!d_values || !f_values)) return set_error(PNX_ERR_INVALID, "d_values / f_values are NULL");
    if (n_cut && (!cutoffs_host || !d_cut || !f_cut)) return set_error(PNX_ERR_INVALID, "cutoffs / d_cut / f_cut are NULL");
    if (!(rel_height >= 0)) return set_error(PNX_ERR_INVALID, "rel_height must be at least 0");  // scipy: ValueError
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    if (n_vox == 0) return PNX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(PNX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_error(PNX_ERR_INVALID, "device %d out of range", device);
    PNX_HIPS(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    hipDeviceProp_t prop;
    PNX_HIPS(hipGetDeviceProperties(&prop, device));

    DevTmp dspec, dnp, dd, df, ddc, dfc;
    PeakArgs a;
    a.n_vox = n_vox;
    a.n_bins = n_bins;
    a.max_peaks = max_peaks;
    a.n_cut = n_cut;
    a.regularized = regularized;
    a.height = height;
    a.rel_height = rel_height;
    const bool wide = n_bins > kNarrowBins;
    for (int j = 0; j < kNarrowBins; ++j) a.bins[j] = (!wide && j < n_bins) ? bins_host[j] : 0.0;
    a.bins_wide = nullptr;
    for (int c = 0; c < 2 * n_cut; ++c) a.cut[c] = cutoffs_host[c];
    const size_t nv = (size_t)n_vox;
    if (mem == PNX_MEM_DEVICE) {
        a.spec = spectrum;
        a.n_peaks = n_peaks;
        a.d_values = max_peaks ? d_values : nullptr;
        a.f_values = f_values;
        a.d_cut = n_cut ? d_cut : nullptr;
        a.f_cut = f_cut;
    } else {
        PNX_HIPS(hipMalloc(&dspec.p, nv * n_bins * sizeof(double)));
        PNX_HIPS(hipMemcpyAsync(dspec.p, spectrum, nv * n_bins * sizeof(double), hipMemcpyHostToDevice, st));
        a.spec = (const double *)dspec.p;
        a.n_peaks = nullptr;
        if (n_peaks) {
            PNX_HIPS(hipMalloc(&dnp.p, nv * sizeof(int32_t)));
            a.n_peaks = (int32_t *)dnp.p;
        }
        a.d_values = a.f_values = a.d_cut = a.f_cut = nullptr;
        if (max_peaks) {
            PNX_HIPS(hipMalloc(&dd.p, nv * max_peaks * sizeof(double)));
            PNX_HIPS(hipMalloc(&df.p, nv * max_peaks * sizeof(double)));
            a.d_values = (double *)dd.p;
            a.f_values = (double *)df.p;
        }
        if (n_cut) {
            PNX_HIPS(hipMalloc(&ddc.p, nv * n_cut * sizeof(double)));
            PNX_HIPS(hipMalloc(&dfc.p, nv * n_cut * sizeof(double)));
            a.d_cut = (double *)ddc.p;
            a.f_cut = (double *)dfc.p;
        }
    }
    long long grid = (n_vox + 3) / 4;
    const long long cap = (long long)prop.multiProcessorCount * 8;
    if (grid > cap) grid = cap;
    if (wide) {  // stream ordered: allocated, filled (the copy from pageable memory is staged before the call returns), used, freed
        double *dbins = nullptr;
        PNX_HIPS(hipMallocAsync((void **)&dbins, (size_t)n_bins * sizeof(double), st));
        hipError_t e = hipMemcpyAsync(dbins, bins_host, (size_t)n_bins * sizeof(double), hipMemcpyHostToDevice, st);
        a.bins_wide = dbins;
        if (e == hipSuccess) {
            hipLaunchKernelGGL(spectrum_peaks_kernel<8>, dim3((unsigned)grid), dim3(256), 0, st, a);
            e = hipGetLastError();
        }
        const hipError_t e2 = hipFreeAsync(dbins, st);
        if (e != hipSuccess || e2 != hipSuccess)
            return pnx::set_error(PNX_ERR_HIP, "wide spectrum launch: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    } else {
        hipLaunchKernelGGL(spectrum_peaks_kernel<4>, dim3((unsigned)grid), dim3(256), 0, st, a);
        PNX_HIPS(hipGetLastError());
    }
    if (mem == PNX_MEM_HOST) {
        if (n_peaks) PNX_HIPS(hipMemcpyAsync(n_peaks, a.n_peaks, nv * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (max_peaks) {
            PNX_HIPS(hipMemcpyAsync(d_values, a.d_values, nv * max_peaks * sizeof(double), hipMemcpyDeviceToHost, st));
            PNX_HIPS(hipMemcpyAsync(f_values, a.f_values, nv * max_peaks * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        if (n_cut) {
            PNX_HIPS(hipMemcpyAsync(d_cut, a.d_cut, nv * n_cut * sizeof(double), hipMemcpyDeviceToHost, st));
            PNX_HIPS(hipMemcpyAsync(f_cut, a.f_cut, nv * n_cut * sizeof(double), hipMemcpyDeviceToHost, st));
        }
    }
    if (mem == PNX_MEM_HOST) PNX_HIPS(hipStreamSynchronize(st));  // the staging buffers are freed on return; device mode only enqueues
    return PNX_OK;
}

int pnx_scatter_maps_f32(const double *values, const int64_t *linear_index, int64_t n_px, int k, int64_t n_spatial, float *out,
                         int mem, int device, void *stream) {
    using namespace pnx;
    if (n_px < 0 || k < 1 || n_spatial < 0 || !out || (n_px && (!values || !linear_index))) return set_error(PNX_ERR_INVALID, "bad argument");
    if (mem != PNX_MEM_HOST && mem != PNX_MEM_DEVICE) return set_error(PNX_ERR_INVALID, "mem=%d", mem);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(PNX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_error(PNX_ERR_INVALID, "device %d out of range", device);
    PNX_HIPS(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nout = (size_t)n_spatial * k;
    if (mem == PNX_MEM_HOST) {
        for (int64_t i = 0; i < n_px; ++i)
            if (linear_index[i] < 0 || linear_index[i] >= n_spatial) return set_error(PNX_ERR_INVALID, "linear_index[%lld] out of range", (long long)i);
    }
    DevTmp dv, di, dout;
    const double *v = values;
    const long long *ix = (const long long *)linear_index;
    float *o = out;
    if (mem == PNX_MEM_HOST) {
        PNX_HIPS(hipMalloc(&dv.p, (size_t)n_px * k * sizeof(double) + 8));
        PNX_HIPS(hipMalloc(&di.p, (size_t)n_px * sizeof(long long) + 8));
        PNX_HIPS(hipMalloc(&dout.p, nout * sizeof(float) + 8));
        PNX_HIPS(hipMemcpyAsync(dv.p, values, (size_t)n_px * k * sizeof(double), hipMemcpyHostToDevice, st));
        PNX_HIPS(hipMemcpyAsync(di.p, linear_index, (size_t)n_px * sizeof(long long), hipMemcpyHostToDevice, st));
        v = (const double *)dv.p;
        ix = (const long long *)di.p;
        o = (float *)dout.p;
    }
    PNX_HIPS(hipMemsetAsync(o, 0, nout * sizeof(float), st));
    if (n_px) {
        const long long tot = (long long)n_px * k;
        hipLaunchKernelGGL(scatter_maps_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, v, ix, (long long)n_px, k, o);
        PNX_HIPS(hipGetLastError());
    }
    if (mem == PNX_MEM_HOST) {
        PNX_HIPS(hipMemcpyAsync(out, o, nout * sizeof(float), hipMemcpyDeviceToHost, st));
        PNX_HIPS(hipStreamSynchronize(st));
    }
    return PNX_OK;
}

}  // extern "C"
