// pnx_nnls.hip -- batched Tikhonov-regularised NNLS (Lawson-Hanson active set) for gfx950, fp64.
//
// Hot path replaced: NNLSSolver._fit_single_pixel -> scipy.optimize.nnls(A, y_ext, maxiter)
// (reference src/pyneapple/solvers/nnls_solver.py:182-210) for every voxel, with A = [basis; reg]
// (nnls_solver.py:61-73) shared by all voxels and y_ext = [y | 0] never materialised (nnls_solver.py:75-86).
//
// Mapping to the machine:
//   * ONE WAVEFRONT OWNS ONE VOXEL (workgroup = 1 wave, persistent, pulling voxels from an atomic queue).
//     The 250-bin vectors (dual w, A^T y, x) are spread 4 bins per lane; the passive set is a position list.
//   * A is shared, so the per-voxel QR of SciPy's kernel is replaced by the Gram form: G = A^T A is built
//     once per plan (fp64), each voxel only needs A^T y = B^T y and sub-blocks of G.  The passive-set
//     normal equations G_PP z = (A^T y)_P are solved through M = L^{-1} (L = chol(G_PP)) kept explicitly:
//     with M every step of the active-set iteration is a mat-vec that parallelises over the wavefront
//     (no serial triangular-solve chain):   entering column: l = M g, new row of M = [-(l^T M)/lam, 1/lam];
//     leaving column k: Givens rotations on adjacent rows of M (column k removed) that annihilate M[:,k].
//     Same selection rule, 0.01 independence test, ztest rejection, alpha interpolation, round-off clean-up
//     loop and `iteration == maxiter` failure as Lawson-Hanson / SciPy 1.15 -> identical iteration counts.
//   * Rows 0..63 of M live in LDS (packed lower triangle, 16.6 KB per wave; triangular-number row offsets make
//     the row-wise ds_read_b64 conflict free), rows >= 64 in a per-wave HBM/L2 scratch.
//   * G (n_bins^2 fp64 = 0.5 MB) and B stay L2 resident; the dual update streams p columns of G per iteration.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "pnx_internal.hpp"
#include "pnx_nnls.hpp"

namespace pnx {

constexpr int kW = 64;
constexpr int kSlots = kNnlsMaxBins / kW;  // 4
constexpr int kLdsRows = 64;
constexpr int kLdsTri = kLdsRows * (kLdsRows + 1) / 2;                         // 2080 doubles
constexpr int kGlobTri = kNnlsMaxBins * (kNnlsMaxBins + 1) / 2 - kLdsTri;      // doubles per wave
constexpr int kNone = 1 << 30;

struct NnlsArgs {
    const double *y;
    double *coeff;
    double *rnorm;
    int8_t *status;
    int32_t *iters;
    const double *G;
    const double *B;
    const double *RT;
    double *Mglob;
    unsigned long long *queue;
    long long n_vox;
    int n_meas, n_bins, n_reg, max_iter;
};

__device__ inline int tri(int i) { return i * (i + 1) / 2; }

// Cross-lane reductions on the DPP data path (row_shr 1/2/4/8, row_bcast 15/31): a handful of VALU ops with
// register-file latency instead of 12 ds_bpermute round trips through the LDS crossbar per fp64 reduction.
template <int CTRL, int ROW_MASK, bool ZERO_FILL> __device__ inline double dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = ZERO_FILL ? 0 : lo, ohi = ZERO_FILL ? 0 : hi;
    const int rlo = __builtin_amdgcn_update_dpp(olo, lo, CTRL, ROW_MASK, 0xf, ZERO_FILL);
    const int rhi = __builtin_amdgcn_update_dpp(ohi, hi, CTRL, ROW_MASK, 0xf, ZERO_FILL);
    return __hiloint2double(rhi, rlo);
}
template <int CTRL, int ROW_MASK> __device__ inline int dpp_mov_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
constexpr int kShr1 = 0x111, kShr2 = 0x112, kShr4 = 0x114, kShr8 = 0x118, kBc15 = 0x142, kBc31 = 0x143;

// inclusive prefix sum over the 64 lanes (lane 63 holds the total)
__device__ inline double wave_incl_scan(double v, int) {
    v += dpp_mov<kShr1, 0xf, true>(v);
    v += dpp_mov<kShr2, 0xf, true>(v);
    v += dpp_mov<kShr4, 0xf, true>(v);
    v += dpp_mov<kShr8, 0xf, true>(v);
    v += dpp_mov<kBc15, 0xa, true>(v);
    v += dpp_mov<kBc31, 0xc, true>(v);
    return v;
}
__device__ inline double bcast_lane63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v) { return bcast_lane63(wave_incl_scan(v, 0)); }
__device__ inline double wave_max(double v) {
    v = fmax(v, dpp_mov<kShr1, 0xf, false>(v));
    v = fmax(v, dpp_mov<kShr2, 0xf, false>(v));
    v = fmax(v, dpp_mov<kShr4, 0xf, false>(v));
    v = fmax(v, dpp_mov<kShr8, 0xf, false>(v));
    v = fmax(v, dpp_mov<kBc15, 0xa, false>(v));
    v = fmax(v, dpp_mov<kBc31, 0xc, false>(v));
    return bcast_lane63(v);
}
__device__ inline double wave_min(double v) { return -wave_max(-v); }
__device__ inline int wave_min_i(int v) {
    int t;
    t = dpp_mov_i<kShr1, 0xf>(v); v = t < v ? t : v;
    t = dpp_mov_i<kShr2, 0xf>(v); v = t < v ? t : v;
    t = dpp_mov_i<kShr4, 0xf>(v); v = t < v ? t : v;
    t = dpp_mov_i<kShr8, 0xf>(v); v = t < v ? t : v;
    t = dpp_mov_i<kBc15, 0xa>(v); v = t < v ? t : v;
    t = dpp_mov_i<kBc31, 0xc>(v); v = t < v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}

// Bin ownership: lane l holds bins {2l, 2l+1, 128+2l, 128+2l+1} so that one row of G (leading dimension 256,
// zero padded) is fetched with two 16-byte loads per lane.
__device__ inline int binof(int lane, int s) { return ((s >> 1) << 7) + 2 * lane + (s & 1); }

// Column pass over the packed lower-triangular M: out[s] (k = lane + 64 s) = sum_{i >= k} v[i] * M[i][k].
// v is an LDS vector (broadcast reads).  Two vectors at once (va, vb) so one sweep of M serves both.
template <bool TWO>
__device__ inline void col_pass(const double *Mlds, const double *Mg, int p, int lane, const double *va,
                                const double *vb, double (&oa)[kSlots], double (&ob)[kSlots]) {
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        oa[s] = 0;
        ob[s] = 0;
    }
    const int p_lds = p < kLdsRows ? p : kLdsRows;
    for (int i = 0; i < p_lds; ++i) {
        const double a = va[i];
        const double b = TWO ? vb[i] : 0.0;
        if (lane <= i) {
            const double m = Mlds[tri(i) + lane];
            oa[0] += a * m;
            if (TWO) ob[0] += b * m;
        }
    }
    for (int i = kLdsRows; i < p; ++i) {
        const double a = va[i];
        const double b = TWO ? vb[i] : 0.0;
        const double *row = Mg + (tri(i) - kLdsTri);
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int k = lane + kW * s;
            if (k <= i) {
                const double m = row[k];
                oa[s] += a * m;
                if (TWO) ob[s] += b * m;
            }
        }
    }
}

__global__ void __launch_bounds__(64) nnls_kernel(const NnlsArgs A) {
    extern __shared__ double sm[];
    double *Mlds = sm;                     // kLdsTri
    double *qv = sm + kLdsTri;             // q = M (A^T y)_P, by position
    double *xv = qv + kNnlsMaxBins;        // x on the passive set, by position
    double *t1 = xv + kNnlsMaxBins;        // scratch vectors
    double *t2 = t1 + kNnlsMaxBins;
    unsigned short *pidx = (unsigned short *)(t2 + kNnlsMaxBins);  // position -> bin
    const int lane = threadIdx.x;
    double *Mg = A.Mglob + (size_t)blockIdx.x * kGlobTri;
    const int n = A.n_bins, nm = A.n_meas, nreg = A.n_reg;
    const int m_total = nm + nreg;

    for (;;) {
        unsigned long long vq = 0;
        if (lane == 0) vq = atomicAdd(A.queue, 1ULL);
        vq = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(vq >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)vq);
        if (vq >= (unsigned long long)A.n_vox) break;
        const long long vox = (long long)vq;
        const double *yv = A.y + (size_t)vox * nm;

        // ---- load y, A^T y
        bool finite = true;
        double yn2 = 0;
        __syncthreads();
        for (int k = lane; k < nm; k += kW) {
            const double v = yv[k];
            t1[k] = v;
            finite = finite && isfinite(v);
            yn2 += v * v;
        }
        yn2 = wave_sum(yn2);
        finite = __all(finite ? 1 : 0) != 0;
        __syncthreads();
        double aty[kSlots], w[kSlots], z[kSlots];
        bool inP[kSlots];
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const int j = binof(lane, s);
            double acc = 0;
            if (j < n && finite)
                for (int k = 0; k < nm; ++k) acc += A.B[(size_t)k * n + j] * t1[k];
            aty[s] = acc;
            inP[s] = false;
            z[s] = 0;
        }
        int p = 0, iteration = 0, status = finite ? 1 : -2;

        while (status == 1 && p < n && p < m_total) {
            // ---- dual w = A^T y - G[:,P] x_P on the zero set
#pragma unroll
            for (int s = 0; s < kSlots; ++s) w[s] = aty[s];
            // p rows of G (L2 resident), 8 in flight: the rows are independent, only the four accumulators chain
            for (int pos0 = 0; pos0 < p; pos0 += 8) {
                double2 ga[8], gb[8];
                double xs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int pos = pos0 + u < p ? pos0 + u : p - 1;
                    const int col = __builtin_amdgcn_readfirstlane((int)pidx[pos]);
                    xs[u] = pos0 + u < p ? xv[pos] : 0.0;
                    const double *gc = A.G + (size_t)col * kNnlsMaxBins + 2 * lane;
                    ga[u] = *reinterpret_cast<const double2 *>(gc);
                    gb[u] = *reinterpret_cast<const double2 *>(gc + 128);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    w[0] -= ga[u].x * xs[u];
                    w[1] -= ga[u].y * xs[u];
                    w[2] -= gb[u].x * xs[u];
                    w[3] -= gb[u].y * xs[u];
                }
            }
#pragma unroll
            for (int s = 0; s < kSlots; ++s)
                if (inP[s] || binof(lane, s) >= n) w[s] = -INFINITY;

            bool accepted = false;
            int jmax = 0;
            double lam = 0, qn = 0;
            double l[kSlots];
            for (;;) {
                // ---- largest positive w_j (ties: lowest bin)
                double best = -INFINITY;
#pragma unroll
                for (int s = 0; s < kSlots; ++s) best = fmax(best, w[s]);
                best = wave_max(best);
                int bj = kNone;
#pragma unroll
                for (int s = kSlots - 1; s >= 0; --s)
                    if (w[s] == best) bj = binof(lane, s);
                bj = wave_min_i(bj);  // ties: lowest bin
                if (!(best > 0)) break;  // KKT satisfied
                jmax = bj;
                // ---- g = G[P, jmax] -> t1 ; l = M g
                __syncthreads();
                for (int pos = lane; pos < p; pos += kW) t1[pos] = A.G[(size_t)jmax * kNnlsMaxBins + pidx[pos]];
                __syncthreads();
                const double Gjj = A.G[(size_t)jmax * kNnlsMaxBins + jmax];
                double atyj;
                {
                    const int ol = (jmax & 127) >> 1, os = ((jmax >> 7) << 1) | (jmax & 1);  // owner lane / slot
                    const double av = os == 0 ? aty[0] : (os == 1 ? aty[1] : (os == 2 ? aty[2] : aty[3]));
                    atyj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(av), ol),
                                            __builtin_amdgcn_readlane(__double2loint(av), ol));
                }
                double ll = 0, lq = 0;
#pragma unroll
                for (int s = 0; s < kSlots; ++s) {
                    l[s] = 0;
                    const int i = lane + kW * s;
                    if (kW * s < p) {
                        if (i < p) {
                            double acc = 0;
                            if (s == 0) {
                                const double *row = Mlds + tri(i);
                                for (int k = 0; k <= i; ++k) acc += row[k] * t1[k];
                            } else {
                                const double *row = Mg + (tri(i) - kLdsTri);
                                for (int k = 0; k <= i; ++k) acc += row[k] * t1[k];
                            }
                            l[s] = acc;
                            ll += acc * acc;
                            lq += acc * qv[i];
                        }
                    }
                }
                ll = wave_sum(ll);
                lq = wave_sum(lq);
                const double lam2 = Gjj - ll;
                // Gram form: lam^2 = G_jj - |l|^2 carries an absolute rounding error of a few eps*G_jj; below that
                // floor the column is numerically dependent on the passive set (only happens without regulariser)
                lam = lam2 > 64.0 * 2.220446049250313e-16 * Gjj ? sqrt(lam2) : 0.0;
                const double un = sqrt(ll);
                bool ok = ((un + lam * 0.01) - un) > 0;  // Lawson-Hanson linear-independence test
                if (ok) {
                    qn = (atyj - lq) / lam;
                    const double ztest = qn / lam;
                    ok = ztest > 0;
                }
                if (ok) {
                    accepted = true;
                    break;
                }
                // reject: w[j] = 0 and look for the next largest
#pragma unroll
                for (int s = 0; s < kSlots; ++s)
                    if (binof(lane, s) == jmax) w[s] = 0.0;
            }
            if (!accepted) break;
#ifdef PNX_NNLS_TRACE
            if (lane == 0 && blockIdx.x == 0) printf("A it=%d p=%d j=%d lam=%.17g qn=%.17g\n", iteration, p, jmax, lam, qn);
#endif

            // ---- column jmax enters: new row of M and z = M^T q in one sweep over M
            __syncthreads();
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                const int i = lane + kW * s;
                if (i < p) t2[i] = l[s];
            }
            __syncthreads();
            {
                double a1[kSlots], a2[kSlots];
                col_pass<true>(Mlds, Mg, p, lane, t2, qv, a1, a2);
                const double inv = 1.0 / lam;
                double *rowp = (p < kLdsRows) ? (Mlds + tri(p)) : (Mg + (tri(p) - kLdsTri));
#pragma unroll
                for (int s = 0; s < kSlots; ++s) {
                    const int k = lane + kW * s;
                    if (k < p) {
                        const double r = -a1[s] * inv;
                        rowp[k] = r;
                        z[s] = a2[s] + r * qn;
                    } else if (k == p) {
                        rowp[k] = inv;
                        z[s] = qn * inv;
                    }
                    if (binof(lane, s) == jmax) inP[s] = true;
                }
                if (lane == 0) {
                    qv[p] = qn;
                    xv[p] = 0.0;
                    pidx[p] = (unsigned short)jmax;
                }
                p += 1;
            }
            __syncthreads();

            // ---- inner loop: keep the passive-set solution feasible
            for (;;) {
                iteration += 1;
                if (iteration == A.max_iter) {
                    status = 0;
                    break;
                }
                double bestT = INFINITY;
                int bpos = kNone;
#pragma unroll
                for (int s = 0; s < kSlots; ++s) {
                    const int i = lane + kW * s;
                    if (i < p && z[s] <= 0) {
                        const double xi = xv[i];
                        const double T = -xi / (z[s] - xi);
                        if (T < bestT) {
                            bestT = T;
                            bpos = i;
                        }
                    }
                }
                {
                    const double gmin = wave_min(bestT);
                    bpos = (bestT == gmin && bpos != kNone) ? bpos : kNone;
                    bpos = wave_min_i(bpos);  // ties: first position (Lawson-Hanson keeps the first minimum)
                    bestT = gmin;
                }
                __syncthreads();
                if (bpos == kNone) {
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) {
                        const int i = lane + kW * s;
                        if (i < p) xv[i] = z[s];
                    }
                    __syncthreads();
                    break;
                }
                const double alpha = bestT;
#pragma unroll
                for (int s = 0; s < kSlots; ++s) {
                    const int i = lane + kW * s;
                    if (i < p) {
                        const double xi = xv[i];
                        xv[i] = xi + alpha * (z[s] - xi);
                    }
                }
                __syncthreads();
                int jj = bpos;
                for (;;) {
#ifdef PNX_NNLS_TRACE
                    if (lane == 0 && blockIdx.x == 0) printf("R it=%d p=%d jj=%d alpha=%.17g\n", iteration, p, jj, alpha);
#endif
                    // ---- position jj leaves the passive set
                    double mv[kSlots], pre[kSlots];
                    double carry = 0;
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) {
                        const int i = lane + kW * s;
                        mv[s] = 0;
                        if (kW * s < p) {
                            if (i >= jj && i < p) {
                                mv[s] = (s == 0) ? Mlds[tri(i) + jj] : Mg[tri(i) - kLdsTri + jj];
                            }
                            const double sc = wave_incl_scan(mv[s] * mv[s], lane);
                            pre[s] = sc + carry;
                            carry += bcast_lane63(sc);
                            if (i < p) {
                                t1[i] = pre[s];
                                t2[i] = mv[s];
                            }
                        } else
                            pre[s] = carry;
                    }
                    __syncthreads();
                    double cs[kSlots], sn[kSlots];
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) {
                        const int i = lane + kW * s;
                        cs[s] = 1.0;
                        sn[s] = 0.0;
                        if (i >= jj && i < p - 1) {
                            const double a = (i == jj) ? mv[s] : sqrt(pre[s]);  // first carried value keeps its sign
                            const double b = t2[i + 1];
                            const double r = sqrt(t1[i + 1]);
                            if (r > 0) {
                                cs[s] = b / r;
                                sn[s] = a / r;
                            }
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) {
                        const int i = lane + kW * s;
                        if (i < p) {
                            t1[i] = cs[s];
                            t2[i] = sn[s];
                        }
                    }
                    const int bin_out = pidx[jj];
                    __syncthreads();
                    {
                        double car[kSlots];
                        const double *rowj = (jj < kLdsRows) ? (Mlds + tri(jj)) : (Mg + (tri(jj) - kLdsTri));
#pragma unroll
                        for (int s = 0; s < kSlots; ++s) {
                            const int c = lane + kW * s;
                            car[s] = (c < jj) ? rowj[c] : 0.0;
                        }
                        double carq = qv[jj];
                        for (int i = jj; i < p - 1; ++i) {
                            const double c_ = t1[i], s_ = t2[i];
                            const double qnx = qv[i + 1];
                            const double *rown = (i + 1 < kLdsRows) ? (Mlds + tri(i + 1)) : (Mg + (tri(i + 1) - kLdsTri));
                            double *rowo = (i < kLdsRows) ? (Mlds + tri(i)) : (Mg + (tri(i) - kLdsTri));
                            double outv[kSlots];
#pragma unroll
                            for (int s = 0; s < kSlots; ++s) {
                                const int c = lane + kW * s;
                                outv[s] = 0;
                                if (kW * s <= i && c <= i) {
                                    const double nxt = rown[c < jj ? c : c + 1];
                                    outv[s] = c_ * car[s] - s_ * nxt;
                                    car[s] = s_ * car[s] + c_ * nxt;
                                }
                            }
                            // row i+1 has been read by every lane before row i (<= i entries) is overwritten:
                            // rows are distinct, so no ordering issue inside the step
#pragma unroll
                            for (int s = 0; s < kSlots; ++s) {
                                const int c = lane + kW * s;
                                if (kW * s <= i && c <= i) rowo[c] = outv[s];
                            }
                            const double oq = c_ * carq - s_ * qnx;
                            carq = s_ * carq + c_ * qnx;
                            if (lane == 0) qv[i] = oq;
                        }
                    }
                    __syncthreads();
                    // ---- drop position jj from x / pidx
                    double xs[kSlots];
                    unsigned short ps[kSlots];
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) {
                        const int i = lane + kW * s;
                        xs[s] = 0;
                        ps[s] = 0;
                        if (i >= jj && i < p - 1) {
                            xs[s] = xv[i + 1];
                            ps[s] = pidx[i + 1];
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) {
                        const int i = lane + kW * s;
                        if (i >= jj && i < p - 1) {
                            xv[i] = xs[s];
                            pidx[i] = ps[s];
                        }
                        if (binof(lane, s) == bin_out) inP[s] = false;
                    }
                    p -= 1;
                    __syncthreads();
#ifdef PNX_NNLS_TRACE
                    if (lane == 0 && blockIdx.x == 0 && iteration == 21) {
                        for (int i = 0; i < p; ++i) printf("  q[%d]=%.15g Mii=%.15g Mi0=%.15g x=%.15g pidx=%d\n", i, qv[i], Mlds[tri(i)+i], Mlds[tri(i)], xv[i], (int)pidx[i]);
                    }
#endif
                    // ---- round-off clean-up: any remaining x <= 0 leaves too (first position first)
                    int bad = kNone;
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) {
                        const int i = lane + kW * s;
                        if (i < p && xv[i] <= 0 && i < bad) bad = i;
                    }
                    bad = wave_min_i(bad);
                    if (bad == kNone) break;
                    jj = bad;
                }
                // ---- z = M^T q
                {
                    double dummy[kSlots];
                    col_pass<false>(Mlds, Mg, p, lane, qv, qv, z, dummy);
                }
            }
        }

        // ---- outputs
        double *cv = A.coeff + (size_t)vox * n;
        __syncthreads();
        for (int j = lane; j < n; j += kW) t1[j] = 0.0;
        __syncthreads();
        if (status == 1)
            for (int pos = lane; pos < p; pos += kW) t1[pidx[pos]] = xv[pos];
        __syncthreads();
        for (int j = lane; j < n; j += kW) cv[j] = t1[j];
        double rn;
        if (status == 1) {
            // rnorm = || [B; reg] x - [y; 0] ||_2 evaluated directly (no ||y||^2 - ||q||^2 cancellation)
            double acc = 0;
            for (int k = lane; k < nm; k += kW) {
                double r = -yv[k];
                for (int pos = 0; pos < p; ++pos) r += A.B[(size_t)k * n + pidx[pos]] * xv[pos];
                acc += r * r;
            }
            for (int i = lane; i < nreg; i += kW) {
                double r = 0;
                for (int pos = 0; pos < p; ++pos) r += A.RT[(size_t)pidx[pos] * nreg + i] * xv[pos];
                acc += r * r;
            }
            rn = sqrt(wave_sum(acc));
        } else
            rn = sqrt(yn2);  // reference failure path: zeros, ||y_ext|| (nnls_solver.py:205-210)
        if (lane == 0) {
            A.rnorm[vox] = rn;
            if (A.status) A.status[vox] = (int8_t)status;
            if (A.iters) A.iters[vox] = iteration;
        }
    }
}

// ---- plan-time kernels ------------------------------------------------------------------------
// G = B^T B + reg^T reg in fp64 (one-off, 2*n^2*(n_meas+n_reg) flop = 35 MFLOP for 250 bins).
__global__ void gram_kernel(const double *B, const double *RT, int nm, int n, int nreg, double *G) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n || i >= n) return;
    double acc = 0;
    for (int k = 0; k < nm; ++k) acc += B[(size_t)k * n + i] * B[(size_t)k * n + j];
    for (int r = 0; r < nreg; ++r) acc += RT[(size_t)i * nreg + r] * RT[(size_t)j * nreg + r];
    G[(size_t)i * kNnlsMaxBins + j] = acc;
}

__global__ void basis_kernel(const double *b, const double *bins, int nm, int n, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n || i >= nm) return;
    out[(size_t)i * n + j] = exp(-b[i] * bins[j]);  // model_functions/nnls.py:41-43
}

#define PNX_HIPN(call)                                                                             \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) return set_error(PNX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

static size_t nnls_lds_bytes() { return sizeof(double) * (kLdsTri + 4 * kNnlsMaxBins) + sizeof(unsigned short) * kNnlsMaxBins; }

int nnls_plan_init(NnlsPlanData *P, int n_meas, int n_bins, const double *basis, const double *reg, int n_reg,
                   int device, int cus) {
    P->device = device;
    P->cus = cus;
    P->n_meas = n_meas;
    P->n_bins = n_bins;
    P->n_reg = n_reg;
    const size_t nb = (size_t)n_meas * n_bins, nr = (size_t)n_reg * n_bins, ng = (size_t)n_bins * kNnlsMaxBins;
    for (size_t i = 0; i < nb; ++i)
        if (!std::isfinite(basis[i])) return set_error(PNX_ERR_INVALID, "basis contains non-finite values");
    PNX_HIPN(hipMalloc(&P->B, nb * sizeof(double)));
    PNX_HIPN(hipMalloc(&P->RT, (nr ? nr : 1) * sizeof(double)));
    PNX_HIPN(hipMalloc(&P->G, ng * sizeof(double)));
    PNX_HIPN(hipMemset(P->G, 0, ng * sizeof(double)));  // rows padded to 256 columns
    PNX_HIPN(hipMalloc(&P->queue, sizeof(unsigned long long)));
    PNX_HIPN(hipMemcpy(P->B, basis, nb * sizeof(double), hipMemcpyHostToDevice));
    if (nr) {
        std::vector<double> rt(nr);
        for (int i = 0; i < n_reg; ++i)
            for (int j = 0; j < n_bins; ++j) {
                const double v = reg[(size_t)i * n_bins + j];
                if (!std::isfinite(v)) return set_error(PNX_ERR_INVALID, "reg contains non-finite values");
                rt[(size_t)j * n_reg + i] = v;
            }
        PNX_HIPN(hipMemcpy(P->RT, rt.data(), nr * sizeof(double), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(gram_kernel, dim3((n_bins + 63) / 64, n_bins), dim3(64), 0, 0, P->B, P->RT, n_meas, n_bins,
                       n_reg, P->G);
    PNX_HIPN(hipGetLastError());
    // persistent grid: as many single-wave workgroups as fit (LDS bound), one scratch slab each
    PNX_HIPN(hipFuncSetAttribute((const void *)nnls_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nnls_lds_bytes()));
    int occ = 0;
    PNX_HIPN(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nnls_kernel, kW, nnls_lds_bytes()));
    if (occ < 1) return set_error(PNX_ERR_HIP, "nnls kernel does not fit on a CU");
    P->n_waves = occ * cus;
    P->mglob_stride = kGlobTri;
    PNX_HIPN(hipMalloc(&P->Mglob, (size_t)P->n_waves * kGlobTri * sizeof(double)));
    PNX_HIPN(hipDeviceSynchronize());
    return PNX_OK;
}

void nnls_plan_free(NnlsPlanData *P) {
    if (P->B) (void)hipFree(P->B);
    if (P->RT) (void)hipFree(P->RT);
    if (P->G) (void)hipFree(P->G);
    if (P->Mglob) (void)hipFree(P->Mglob);
    if (P->queue) (void)hipFree(P->queue);
    *P = NnlsPlanData();
}

int nnls_solve_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d,
                      double *rnorm_d, int8_t *status_d, int32_t *iters_d, hipStream_t stream) {
    NnlsArgs a;
    a.y = y_d;
    a.coeff = coeff_d;
    a.rnorm = rnorm_d;
    a.status = status_d;
    a.iters = iters_d;
    a.G = P->G;
    a.B = P->B;
    a.RT = P->RT;
    a.Mglob = P->Mglob;
    a.queue = P->queue;
    a.n_vox = n_vox;
    a.n_meas = P->n_meas;
    a.n_bins = P->n_bins;
    a.n_reg = P->n_reg;
    a.max_iter = max_iter;
    PNX_HIPN(hipMemsetAsync(P->queue, 0, sizeof(unsigned long long), stream));
    long long grid = n_vox < P->n_waves ? n_vox : P->n_waves;
    hipLaunchKernelGGL(nnls_kernel, dim3((unsigned)grid), dim3(kW), nnls_lds_bytes(), stream, a);
    PNX_HIPN(hipGetLastError());
    return PNX_OK;
}

int nnls_build_basis(int n_meas, const double *b, int n_bins, const double *bins, double *basis) {
    double *db = nullptr, *dbin = nullptr, *dout = nullptr;
    PNX_HIPN(hipMalloc(&db, n_meas * sizeof(double)));
    PNX_HIPN(hipMalloc(&dbin, n_bins * sizeof(double)));
    PNX_HIPN(hipMalloc(&dout, (size_t)n_meas * n_bins * sizeof(double)));
    PNX_HIPN(hipMemcpy(db, b, n_meas * sizeof(double), hipMemcpyHostToDevice));
    PNX_HIPN(hipMemcpy(dbin, bins, n_bins * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(basis_kernel, dim3((n_bins + 63) / 64, n_meas), dim3(64), 0, 0, db, dbin, n_meas, n_bins, dout);
    PNX_HIPN(hipGetLastError());
    PNX_HIPN(hipMemcpy(basis, dout, (size_t)n_meas * n_bins * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(db);
    (void)hipFree(dbin);
    (void)hipFree(dout);
    return PNX_OK;
}

}  // namespace pnx
