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
//   * The kernel is latency bound per wave (throughput scales linearly with resident waves), so the per-voxel
//     on-chip footprint is what matters: ONLY the first 48 rows of M live in LDS (packed lower triangle,
//     9.4 KB per wave, <= 128 VGPRs -> 16 waves per CU; triangular-number row offsets keep ds_read_b64 conflict
//     free); rows >= 48 go to a per-wave L2/HBM scratch and are always touched row-contiguously.  Every
//     position-indexed vector (q, x, z, g, l, rotation coefficients, position -> bin map) lives in registers,
//     4 positions per lane, and is broadcast with v_readlane / moved with DPP instead of LDS round trips.
//   * G (n_bins x 256 fp64 = 0.5 MB, zero padded) and B stay L2 resident; the dual update streams p rows of G per
//     iteration, two 16-byte loads per lane and row, 4 rows in flight.
#include <hip/hip_runtime.h>

#include <mutex>

#include <cmath>
#include <cstdlib>
#include <vector>

#include "pnx_internal.hpp"
#include "pnx_nnls.hpp"
#include "pnx_nnls_dev.hpp"

namespace pnx {

#ifndef PNX_NNLS_LDS_ROWS
#define PNX_NNLS_LDS_ROWS 48
#endif
#ifndef PNX_NNLS_GBATCH
#define PNX_NNLS_GBATCH 4
#endif
#ifndef PNX_NNLS_W6_WPS
#define PNX_NNLS_W6_WPS 3  // waves per SIMD of nnls_kernel<6, 4>
#endif
#ifndef PNX_NNLS_WAVES_PER_SIMD
#define PNX_NNLS_WAVES_PER_SIMD 4
#endif
constexpr int kLdsRows = PNX_NNLS_LDS_ROWS;  // rows of M kept in LDS (<= 64)
constexpr int kLdsTri = kLdsRows * (kLdsRows + 1) / 2;
template <int KS> constexpr int glob_tri() { return (kW * KS) * (kW * KS + 1) / 2 - kLdsTri; }  // doubles of overflow scratch per wave
// Row access that is "LDS or slab" by a (uniform or per-lane) row index.  Typed by address space: with plain pointers the compiler
// folds the two cases into ONE flat access through a selected base pointer, and a flat access waits for both the LDS and the
// vector-memory counter (round 4, found in the block kernel first: pnx_nnls_blk.hip; 15 flat accesses here: + 5-7 %).
// (Also tried here and not kept: the block kernel's closed-form rotation of q and v_rsq-based rotation coefficients --
// 5.09 -> 4.34 M voxels/s at 33 b-values: this kernel sits at exactly 128 registers, the extra scans spill.)
typedef __attribute__((address_space(3))) double gen_lds_double;
typedef __attribute__((address_space(1))) double gen_glb_double;
constexpr int kGBatch = PNX_NNLS_GBATCH;  // rows of G in flight per lane in the dual update
static_assert(kLdsRows <= kW, "LDS rows are owned by the first slot");

struct NnlsArgs {
    const double *y;
    double *coeff;
    double *rnorm;
    int8_t *status;
    int32_t *iters;
    const double *G;   // (n_bins, 256) zero padded
    const double *Bp;  // (n_meas, 256) zero padded
    const double *RT;  // (n_bins, n_reg)
    const double *aty; // (n_vox, 256) = Y * Bp from the MFMA Gram step, or null (computed per wave on the VALU)
    double *Mglob;
    unsigned long long *queue;
    long long n_vox;
    int n_meas, n_bins, n_reg, max_iter;
    double rc[5];  // banded Toeplitz regulariser: R[i][j] = rc[j - i + 2] for |j - i| <= 2 (mu included), zero outside
    int rhb;       // its half bandwidth (1 or 2); 0: general regulariser, rows of RT are used instead
    const int32_t *redo_list, *redo_count;  // non-null: only the voxels redo_list[0 .. *redo_count) (handed over by pnx_nnls_blk.hip, or by <8, 4> to <8, 8>)
    int32_t *bail;  // <KB, KP < KB> only: [0] number of voxels whose passive set outgrew 64 KP positions, [1 ..] their indices
    const int32_t *route = nullptr;  // non-null: the launch only runs when *route == 1 (the pilot of a block-kernel plan chose the Gram form, pnx_nnls_blk.hip)
};

__device__ inline int tri(int i) { return i * (i + 1) / 2; }
constexpr int kGenBail = 2;  // internal status of nnls_kernel<KB, KP < KB>: hand the voxel over

// out[s] (k = lane + 64 s) = sum_{i >= k} va_i * M[i][k]  (and the same with vb when TWO): one sweep over the
// packed lower-triangular M, every row read contiguously; va_i / vb_i are broadcast with v_readlane.
template <bool TWO, int KS>
__device__ inline void col_pass(const double *Mlds, const double *Mg, int p, int lane, const double (&va)[KS],
                                const double (&vb)[KS], double (&oa)[KS], double (&ob)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        oa[s] = 0;
        ob[s] = 0;
    }
    const int p_lds = p < kLdsRows ? p : kLdsRows;
    // branch-free body (clamped address, masked product) so that the unrolled steps keep their LDS reads in flight
    extern __shared__ double dyn_lds[];  // the kernel's dynamic LDS (same base as Mlds): rows of M, then bc[64]
    double *bc = dyn_lds + kLdsTri;
    bc[lane] = va[0];  // broadcast reads (one LDS instruction) instead of two v_readlane per value: the kernel is VALU bound
    for (int i = 0; i < p_lds; ++i) {
        const double a = bc[i];
        const double b = TWO ? rl(vb[0], i) : 0.0;
        // lanes beyond the row read on into the following rows / the broadcast buffer (inside the allocation: tri(47) + 63 <
        // kLdsTri + 64); their product is masked
        const double m0 = Mlds[tri(i) + lane];
        const unsigned long long on = (2ull << i) - 1;  // lanes 0 .. i
        fma_on(oa[0], a, m0, on);
        if (TWO) fma_on(ob[0], b, m0, on);
    }
    // overflow rows (global scratch): four rows' loads are issued before any of them is consumed
    auto one = [&](int i, auto S) {
        constexpr int si = decltype(S)::value;
        const double a = rl(va[si], i & 63);
        const double b = TWO ? rl(vb[si], i & 63) : 0.0;
        const double *row = Mg + (tri(i) - kLdsTri);
#pragma unroll
        for (int s = 0; s <= si; ++s) {
            const int k = lane + kW * s;
            const bool on = k <= i;
            const double m0 = row[k];  // past the row end: the following rows of this wave's slab (masked below)
            const double m = on ? m0 : 0.0;
            oa[s] += a * m;
            if (TWO) ob[s] += b * m;
        }
    };
    auto four = [&](int i, auto S) {
        constexpr int si = decltype(S)::value;
        double m[4][si + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double *row = Mg + (tri(i + r) - kLdsTri);
#pragma unroll
            for (int s = 0; s <= si; ++s) {
                const int k = lane + kW * s;
                m[r][s] = row[k];  // past the row end: the following rows of this wave's slab (masked below)
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double a = rl(va[si], (i + r) & 63);
            const double b = TWO ? rl(vb[si], (i + r) & 63) : 0.0;
#pragma unroll
            for (int s = 0; s <= si; ++s) {
                const double mm = (lane + kW * s <= i + r) ? m[r][s] : 0.0;
                oa[s] += a * mm;
                if (TWO) ob[s] += b * mm;
            }
        }
    };
    for_pos4n<KS>(kLdsRows, p, four, one);
}


#ifdef PNX_NNLS_STAMP
#define STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

// KB: bins per lane, KP: passive-set positions per lane (position i lives in lane i & 63, slot i >> 6).  <4, 4>: up to 256 bins,
// 16 waves per CU.  Wide plans (257 .. 512 bins, A^T y on the VALU): <8, 4> (<6, 4> up to 384 bins) -- twice the bin-indexed registers (dual, A^T y,
// passive flags, rows of G) but the position-indexed ones of the narrow kernel, so three waves per SIMD; a voxel whose passive set
// wants a 257th position is handed to <8, 8> (two waves per SIMD) through the bail list, as the block kernel hands over to <4, 4>.
template <int KB, int KP> __global__ void __launch_bounds__(64, KB == 4 ? PNX_NNLS_WAVES_PER_SIMD : (KP == 4 ? (KB == 6 ? PNX_NNLS_W6_WPS : 3) : 2)) nnls_kernel(const NnlsArgs A) {
    constexpr int kBS = KB == 4 ? kNnlsMaxBins : kNnlsWideBins;  // row stride of G, Bp (and of the MFMA product, KB == 4 only); <6, 4> reads 384 bins of 512-bin rows
#ifdef PNX_NNLS_STAMP
    unsigned long long seg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ double Mlds[];  // packed rows 0..kLdsRows-1 of M, then bc[64]
    double *bc = Mlds + kLdsTri;      // broadcast buffer: position-indexed values of slot 0, read with a uniform address
    const int lane = threadIdx.x;
    double *Mg = A.Mglob + (size_t)blockIdx.x * glob_tri<KP>();
    gen_lds_double *MldsT = (gen_lds_double *)Mlds;
    gen_glb_double *MgT = (gen_glb_double *)Mg;
    const int n = A.n_bins, nm = A.n_meas, nreg = A.n_reg;
    const int m_total = nm + nreg;
    if (A.route && *A.route != 1) return;  // wave uniform: the pilot kept the block kernel, nothing to do here

    for (;;) {
        unsigned long long vq = 0;
        if (lane == 0) vq = atomicAdd(A.queue, 1ULL);
        vq = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(vq >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)vq);
        long long vox = (long long)vq;
        if (A.redo_list) {
            {   // the list never holds more than n_vox entries the caller has room for (a deferred pass is launched with its
                // side buffers' capacity before the count is known to the host)
                const unsigned long long nr = (unsigned long long)*A.redo_count;
                if (vq >= (nr < (unsigned long long)A.n_vox ? nr : (unsigned long long)A.n_vox)) break;
            }
            vox = A.redo_list[vq];
        } else if (vq >= (unsigned long long)A.n_vox)
            break;
        const double *yv = A.y + (size_t)vox * nm;

        // ---- y (by measurement: lane k & 63, slot k >> 6), A^T y = B^T y (by bin)
        double yreg[2] = {0.0, 0.0};
        bool finite = true;
        double yn2 = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = lane + kW * s;
            if (k < nm) {
                yreg[s] = yv[k];
                finite = finite && isfinite(yreg[s]);
                yn2 += yreg[s] * yreg[s];
            }
        }
        yn2 = wave_sum(yn2);
        finite = __all(finite ? 1 : 0) != 0;
        double aty[KB] = {}, w[KB], z[KP] = {};
        bool inP[KB] = {};
        if (KB == 4 && finite && A.aty) {
            const double *ar = A.aty + (size_t)vox * kBS + 2 * lane;
            const double2 a0 = *reinterpret_cast<const double2 *>(ar);
            const double2 a1 = *reinterpret_cast<const double2 *>(ar + 128);
            aty[0] = a0.x;
            aty[1] = a0.y;
            aty[2] = a1.x;
            aty[3] = a1.y;
        } else if (finite) {
#pragma unroll 4
            for (int k = 0; k < nm; ++k) {
                const double yk = k < kW ? rl(yreg[0], k & 63) : rl(yreg[1], k & 63);
                const double *br = A.Bp + (size_t)k * kBS + 2 * lane;
                double2 bh[KB / 2];
#pragma unroll
                for (int h = 0; h < KB / 2; ++h) bh[h] = *reinterpret_cast<const double2 *>(br + 128 * h);
#pragma unroll
                for (int h = 0; h < KB / 2; ++h) {
                    aty[2 * h] += bh[h].x * yk;
                    aty[2 * h + 1] += bh[h].y * yk;
                }
            }
        }
        // position-indexed state
        double q[KP] = {}, x[KP] = {};
        int pidx[KP] = {};
        int p = 0, iteration = 0, status = finite ? 1 : -2;
        STAMP(0);

        while (status == 1 && p < n && p < m_total) {
            if (KP < KB && p >= kW * KP) {  // every position slot is in use: the <KB, KB> instantiation redoes this voxel
                status = kGenBail;
                break;
            }
            // ---- dual w = A^T y - G[:,P] x_P on the zero set: p rows of G (L2 resident), kGBatch in flight
#pragma unroll
            for (int s = 0; s < KB; ++s) w[s] = aty[s];
            // positions 64 sl .. 64 sl + 63 live in register slot sl: one loop per slot keeps the slot index a
            // compile-time constant (a run-time slot select costs ~35 scalar instructions per row)
            bc[lane] = x[0];  // the kernel is VALU-issue bound: x_pos comes back through an LDS broadcast read, not 2 readlanes
#pragma unroll
            for (int sl = 0; sl < KP; ++sl) {
                if (p <= sl * kW) break;  // wave-uniform
                const int cnt = (p - sl * kW) < kW ? (p - sl * kW) : kW;
                int l0 = 0;
                for (; l0 + kGBatch <= cnt; l0 += kGBatch) {  // full batches: no clamping, no masking
                    double2 gg[kGBatch][KB / 2];
                    double xs[kGBatch];
#pragma unroll
                    for (int u = 0; u < kGBatch; ++u) {
                        const int col = __builtin_amdgcn_readlane(pidx[sl], l0 + u);
                        xs[u] = sl == 0 ? bc[l0 + u] : rl(x[sl], l0 + u);
                        const double *gc = A.G + (size_t)col * kBS + 2 * lane;
#pragma unroll
                        for (int h = 0; h < KB / 2; ++h) gg[u][h] = *reinterpret_cast<const double2 *>(gc + 128 * h);
                    }
#pragma unroll
                    for (int u = 0; u < kGBatch; ++u) {
#pragma unroll
                        for (int h = 0; h < KB / 2; ++h) {
                            w[2 * h] -= gg[u][h].x * xs[u];
                            w[2 * h + 1] -= gg[u][h].y * xs[u];
                        }
                    }
                }
                if (l0 < cnt) {  // ragged last batch
                    double2 gg[kGBatch][KB / 2];
                    double xs[kGBatch];
#pragma unroll
                    for (int u = 0; u < kGBatch; ++u) {
                        const bool on = l0 + u < cnt;
                        const int ll_ = on ? l0 + u : cnt - 1;
                        const int col = __builtin_amdgcn_readlane(pidx[sl], ll_);
                        const double xv = rl(x[sl], ll_);
                        xs[u] = on ? xv : 0.0;
                        const double *gc = A.G + (size_t)col * kBS + 2 * lane;
#pragma unroll
                        for (int h = 0; h < KB / 2; ++h) gg[u][h] = *reinterpret_cast<const double2 *>(gc + 128 * h);
                    }
#pragma unroll
                    for (int u = 0; u < kGBatch; ++u) {
#pragma unroll
                        for (int h = 0; h < KB / 2; ++h) {
                            w[2 * h] -= gg[u][h].x * xs[u];
                            w[2 * h + 1] -= gg[u][h].y * xs[u];
                        }
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < KB; ++s)
                if (inP[s] || binof(lane, s) >= n) w[s] = -INFINITY;
            STAMP(1);

            bool accepted = false;
            int jmax = 0;
            double lam = 0, qn = 0, inv_lam = 0;
            double l[KP];
            for (;;) {
                // ---- largest positive w_j (ties: lowest bin)
                double best = -INFINITY;
#pragma unroll
                for (int s = 0; s < KB; ++s) best = fmax(best, w[s]);
                best = wave_max(best);
                if (!(best > 0)) break;  // KKT satisfied
                int bj = kNone;
#pragma unroll
                for (int s = KB - 1; s >= 0; --s)
                    if (w[s] == best) bj = binof(lane, s);
                jmax = wave_min_i(bj);
                // ---- g = G[P, jmax] (by position), l = M g
                const double *grow = A.G + (size_t)jmax * kBS;
                double g[KP];
#pragma unroll
                for (int s = 0; s < KP; ++s) g[s] = (lane + kW * s < p) ? grow[pidx[s]] : 0.0;
                const double Gjj = grow[jmax];
                double atyj;
                {
                    const int ol = (jmax & 127) >> 1, os = ((jmax >> 7) << 1) | (jmax & 1);  // owner lane / slot
                    double av = aty[KB - 1];
#pragma unroll
                    for (int s = KB - 2; s >= 0; --s) av = os == s ? aty[s] : av;
                    atyj = rl(av, ol);
                }
#pragma unroll
                for (int s = 0; s < KP; ++s) l[s] = 0;
                {
                    // LDS rows as a column sweep: uniform k, lane i accumulates M[i][k] g_k for i >= k
                    const int plim = p < kLdsRows ? p : kLdsRows;
                    const bool mine = lane < plim;
                    const double *row0 = Mlds + tri(mine ? lane : 0);
                    // unrolled by hand (the compiler does not unroll loops around v_readlane): four LDS reads in flight,
                    // two accumulators, a quarter of the loop control
                    double acc0 = 0, acc1 = 0;
                    int k = 0;
                    bc[lane] = g[0];
                    for (; k + 4 <= plim; k += 4) {
                        const double g0 = bc[k], g1 = bc[k + 1], g2 = bc[k + 2], g3 = bc[k + 3];
                        const double m0 = row0[k], m1 = row0[k + 1], m2 = row0[k + 2], m3 = row0[k + 3];
                        const unsigned long long on = ~0ull << k;  // lanes >= k; lanes >= plim accumulate garbage they never use
                        fma_on(acc0, m0, g0, on);
                        fma_on(acc1, m1, g1, on << 1);
                        fma_on(acc0, m2, g2, on << 2);
                        fma_on(acc1, m3, g3, on << 3);
                    }
                    for (; k < plim; ++k) {
                        const double gk = bc[k];
                        const double m0 = row0[k];  // tri(lane) + k < tri(kLdsRows): always inside the LDS rows
                        fma_on(acc0, m0, gk, ~0ull << k);
                    }
                    l[0] = mine ? acc0 + acc1 : 0.0;
                }
                // overflow rows: contiguous row read (lanes over k), DPP reduction, result to the owner of i
                {
                    auto one = [&](int i, auto S) {
                        constexpr int si = decltype(S)::value;
                        const double *row = Mg + (tri(i) - kLdsTri);
                        double part = 0;
#pragma unroll
                        for (int s = 0; s <= si; ++s) {
                            const int k = lane + kW * s;
                            const bool on = k <= i;
                            const double m0 = row[k];  // past the row end: the following rows of this wave's slab (masked below)
                            part += on ? m0 * g[s] : 0.0;
                        }
                        const double li = wave_sum(part);
                        if (lane == (i & 63)) l[si] = li;
                    };
                    auto four = [&](int i, auto S) {  // four rows in flight, then four independent reductions
                        constexpr int si = decltype(S)::value;
                        double m[4][si + 1];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double *row = Mg + (tri(i + r) - kLdsTri);
#pragma unroll
                            for (int s = 0; s <= si; ++s) {
                                const int k = lane + kW * s;
                                m[r][s] = row[k];  // past the row end: the following rows of this wave's slab (masked below)
                            }
                        }
                        double part[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            part[r] = 0;
#pragma unroll
                            for (int s = 0; s <= si; ++s) part[r] += (lane + kW * s <= i + r) ? m[r][s] * g[s] : 0.0;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double li = wave_sum(part[r]);
                            if (lane == ((i + r) & 63)) l[si] = li;
                        }
                    };
                    for_pos4n<KP>(kLdsRows, p, four, one);
                }
                double ll = 0, lq = 0;
#pragma unroll
                for (int s = 0; s < KP; ++s) {
                    if (lane + kW * s < p) {
                        ll += l[s] * l[s];
                        lq += l[s] * q[s];
                    }
                }
                ll = wave_sum(ll);
                lq = wave_sum(lq);
                const double lam2 = Gjj - ll;
                // Gram form: lam^2 = G_jj - |l|^2 carries an absolute rounding error of a few eps*G_jj; below that
                // floor the column is numerically dependent on the passive set (only happens without regulariser)
                // wave-uniform scalar algebra on v_rsq_f64 + Newton instead of the IEEE sqrt / divide expansions
                const bool indep = lam2 > 64.0 * 2.220446049250313e-16 * Gjj;
                const double ilam = indep ? rsqrt_nr(lam2) : 0.0;
                lam = lam2 * ilam;
                lam = fma(0.5 * ilam, fma(-lam, lam, lam2), lam);  // sqrt(lam2) to within an ulp
                const double un = ll > 0 ? ll * rsqrt_nr(ll) : 0.0;
                bool ok = ((un + lam * 0.01) - un) > 0;  // Lawson-Hanson linear-independence test
                if (ok) {
                    qn = (atyj - lq) * ilam;
                    ok = qn > 0;  // ztest = qn / lam with lam > 0
                }
                inv_lam = ilam;
                if (ok) {
                    accepted = true;
                    break;
                }
                // reject: w[j] = 0 and look for the next largest
#pragma unroll
                for (int s = 0; s < KB; ++s)
                    if (binof(lane, s) == jmax) w[s] = 0.0;
            }
            STAMP(2);
            if (!accepted) break;
#ifdef PNX_NNLS_TRACE
            if (lane == 0 && blockIdx.x == 0) printf("A it=%d p=%d j=%d lam=%.17g qn=%.17g\n", iteration, p, jmax, lam, qn);
#endif

            // ---- column jmax enters: new row of M and z = M^T q in one sweep over M
            {
                // one sweep over M gives the new row r = -(l^T M) / lam.  z = M^T q is then a rank-one update of the
                // current solution (x == z = M_old^T q_old whenever a column enters): z_k = x_k + r_k * qn.  After
                // every removal z is recomputed from scratch (second sweep below), so nothing drifts.
                double a1[KP], a2[KP];
                col_pass<false, KP>(Mlds, Mg, p, lane, l, l, a1, a2);
                const double inv = inv_lam;
#pragma unroll
                for (int s = 0; s < KP; ++s) {
                    const int k = lane + kW * s;
                    if (k <= p) {
                        const double r = k < p ? -a1[s] * inv : inv;
                        if (p < kLdsRows)  // wave uniform
                            MldsT[tri(p) + k] = r;
                        else
                            MgT[tri(p) - kLdsTri + k] = r;
                        z[s] = k < p ? x[s] + r * qn : qn * inv;
                    }
                }
#pragma unroll
                for (int s = 0; s < KB; ++s) {
                    if (binof(lane, s) == jmax) inP[s] = true;
                }
                put(q, p, qn, lane);
                put(x, p, 0.0, lane);
                put_i(pidx, p, jmax, lane);
                p += 1;
            }
            __syncthreads();
            STAMP(3);

            // ---- inner loop: keep the passive-set solution feasible
            for (;;) {
                iteration += 1;
                if (iteration == A.max_iter) {
                    status = 0;
                    break;
                }
                {
                    // most inner iterations (46 of 64 on the C4 workload) find no z <= 0: one ballot instead of four fp64
                    // divisions and two wave reductions
                    bool viol = false;
#pragma unroll
                    for (int s = 0; s < KP; ++s) viol = viol || (lane + kW * s < p && z[s] <= 0);
                    if (!__any(viol ? 1 : 0)) {
#pragma unroll
                        for (int s = 0; s < KP; ++s)
                            if (lane + kW * s < p) x[s] = z[s];
                        break;
                    }
                }
                double bestT = INFINITY;
                int bpos = kNone;
#pragma unroll
                for (int s = 0; s < KP; ++s) {
                    const int i = lane + kW * s;
                    if (i < p && z[s] <= 0) {
                        const double T = -x[s] / (z[s] - x[s]);
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
                if (bpos == kNone) {
#pragma unroll
                    for (int s = 0; s < KP; ++s)
                        if (lane + kW * s < p) x[s] = z[s];
                    break;
                }
                const double alpha = bestT;
#pragma unroll
                for (int s = 0; s < KP; ++s)
                    if (lane + kW * s < p) x[s] = x[s] + alpha * (z[s] - x[s]);
                STAMP(4);
                int jj = bpos;
                for (;;) {
#ifdef PNX_NNLS_TRACE
                    if (lane == 0 && blockIdx.x == 0) printf("R it=%d p=%d jj=%d alpha=%.17g\n", iteration, p, jj, alpha);
#endif
                    // ---- position jj leaves the passive set: Givens rotations on adjacent rows of M (column jj
                    // removed) that annihilate m = M[:, jj]; coefficients from the prefix norms of m
                    double mv[KP], pre[KP];
                    double carry = 0;
#pragma unroll
                    for (int s = 0; s < KP; ++s) {
                        const int i = lane + kW * s;
                        mv[s] = 0;
                        pre[s] = carry;
                        if (kW * s < p) {
                            if (i >= jj && i < p) {
                                if (i < kLdsRows)
                                    mv[s] = MldsT[tri(i) + jj];
                                else
                                    mv[s] = MgT[tri(i) - kLdsTri + jj];
                            }
                            const double sc = wave_incl_scan(mv[s] * mv[s]);
                            pre[s] = sc + carry;
                            carry += rl(sc, 63);
                        }
                    }
                    double mnext[KP], prenext[KP];
                    shift_down(mv, mnext, lane);
                    shift_down(pre, prenext, lane);
                    double cs[KP], sn[KP];
#pragma unroll
                    for (int s = 0; s < KP; ++s) {
                        const int i = lane + kW * s;
                        cs[s] = 1.0;
                        sn[s] = 0.0;
                        if (i >= jj && i < p - 1) {
                            const double a = (i == jj) ? mv[s] : sqrt(pre[s]);  // first carried value keeps its sign
                            const double b = mnext[s];
                            const double r = sqrt(prenext[s]);
                            if (r > 0) {
                                cs[s] = b / r;
                                sn[s] = a / r;
                            }
                        }
                    }
                    const int bin_out = get_at_i(pidx, jj);
                    {
                        double car[KP];
#pragma unroll
                        for (int s = 0; s < KP; ++s) {
                            const int c = lane + kW * s;
                            car[s] = 0.0;
                            if (c < jj) {
                                if (jj < kLdsRows)  // wave uniform
                                    car[s] = MldsT[tri(jj) + c];
                                else
                                    car[s] = MgT[tri(jj) - kLdsTri + c];
                            }
                        }
                        double carq = get_at(q, jj);
                        double qsh[KP];
                        shift_down(q, qsh, lane);  // qsh[i] = q[i + 1]
                        for_posn<KP, 2>(jj, p - 1, [&](int i, auto S) {
                            constexpr int si = decltype(S)::value;
                            const double c_ = rl(cs[si], i & 63), s_ = rl(sn[si], i & 63);
                            const double qnx = rl(qsh[si], i & 63);
#pragma unroll
                            for (int s = 0; s <= si; ++s) {
                                const int c = lane + kW * s;
                                if (c <= i) {
                                    const int cn = c < jj ? c : c + 1;
                                    double nxt;
                                    if (i + 1 < kLdsRows)  // wave uniform
                                        nxt = MldsT[tri(i + 1) + cn];
                                    else
                                        nxt = MgT[tri(i + 1) - kLdsTri + cn];
                                    const double outv = c_ * car[s] - s_ * nxt;
                                    car[s] = s_ * car[s] + c_ * nxt;
                                    if (i < kLdsRows)
                                        MldsT[tri(i) + c] = outv;
                                    else
                                        MgT[tri(i) - kLdsTri + c] = outv;
                                }
                            }
                            const double oq = c_ * carq - s_ * qnx;
                            carq = s_ * carq + c_ * qnx;
                            if (lane == (i & 63)) q[si] = oq;
                        });
                    }
                    // ---- drop position jj from x / pidx
                    {
                        double xsh[KP];
                        int psh[KP];
                        shift_down(x, xsh, lane);
                        shift_down_i(pidx, psh, lane);
#pragma unroll
                        for (int s = 0; s < KP; ++s) {
                            const int i = lane + kW * s;
                            if (i >= jj && i < p - 1) {
                                x[s] = xsh[s];
                                pidx[s] = psh[s];
                            }
                        }
#pragma unroll
                        for (int s = 0; s < KB; ++s) {
                            if (binof(lane, s) == bin_out) inP[s] = false;
                        }
                    }
                    p -= 1;
                    __syncthreads();
                    // ---- round-off clean-up: any remaining x <= 0 leaves too (first position first)
                    int bad = kNone;
#pragma unroll
                    for (int s = KP - 1; s >= 0; --s) {
                        const int i = lane + kW * s;
                        if (i < p && x[s] <= 0) bad = i;
                    }
                    bad = wave_min_i(bad);
                    if (bad == kNone) break;
                    jj = bad;
                }
                STAMP(5);
                // ---- z = M^T q
                {
                    double dummy[KP];
                    col_pass<false, KP>(Mlds, Mg, p, lane, q, q, z, dummy);
                }
                STAMP(6);
            }
        }
        STAMP(7);

        // ---- outputs: x by bin, rnorm = || [B; reg] x - [y; 0] ||_2 evaluated directly.  M is dead now: its LDS rows serve as
        // the bin-ordered scratch that turns x by position into x by bin (one scatter instead of p x 14 select instructions)
        // and, for the reference's banded regularisers, carries the stencil R x (the generic loop over rows of RT is p
        // dependent L2 round trips per 64 rows of R: 110 k cycles per voxel, 6 % of the kernel, measured with the stamps)
        if (KP < KB && status == kGenBail) {
            if (lane == 0) A.bail[1 + atomicAdd(A.bail, 1)] = (int32_t)vox;
            continue;
        }
        double xb[KB] = {};
        double rn;
        if (status == 1) {
            wave_sync();
            double *xbuf = Mlds;  // [2 + 64 KB + 2] <= kLdsTri
            double tt = 0, dummy[KB];
            reg_terms<false>(xbuf, A.rc, A.rhb, A.rhb ? nreg : 0, p, lane, x, pidx, dummy, &tt);  // leaves x in bin order in xbuf
#pragma unroll
            for (int s = 0; s < KB; ++s) xb[s] = xbuf[2 + binof(lane, s)];
            wave_sync();
            double acc = 0;
            for (int k = 0; k < nm; ++k) {
                const double *br = A.Bp + (size_t)k * kBS + 2 * lane;
                double2 bh[KB / 2];
#pragma unroll
                for (int h = 0; h < KB / 2; ++h) bh[h] = *reinterpret_cast<const double2 *>(br + 128 * h);
                double part = bh[0].x * xb[0] + bh[0].y * xb[1];
#pragma unroll
                for (int h = 1; h < KB / 2; ++h) {
                    part += bh[h].x * xb[2 * h];
                    part += bh[h].y * xb[2 * h + 1];
                }
                const double fit = wave_sum(part);
                const double yk = k < kW ? rl(yreg[0], k & 63) : rl(yreg[1], k & 63);
                if (lane == 0) acc += (fit - yk) * (fit - yk);
            }
            acc = rl(acc, 0);
            double racc = tt;
            if (!A.rhb) {  // general regulariser: rows of reg, 64 at a time
                for (int i0 = 0; i0 < nreg; i0 += kW) {
                    const int i = i0 + lane;
                    double r = 0;
                    for_posn<KP, 4>(0, p, [&](int pos, auto S) {
                        constexpr int si = decltype(S)::value;
                        const int b = __builtin_amdgcn_readlane(pidx[si], pos & 63);
                        const double xv = rl(x[si], pos & 63);
                        if (i < nreg) r += A.RT[(size_t)b * nreg + i] * xv;
                    });
                    racc += r * r;
                }
            }
            rn = sqrt(acc + wave_sum(racc));
        } else
            rn = sqrt(yn2);  // reference failure path: zeros, ||y_ext|| (nnls_solver.py:205-210)
        double *cv = A.coeff + (size_t)vox * n;
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            const int j = binof(lane, s);
            if (j < n) cv[j] = xb[s];
        }
        if (lane == 0) {
            A.rnorm[vox] = rn;
            if (A.status) A.status[vox] = (int8_t)status;
            if (A.iters) A.iters[vox] = iteration;
        }
        STAMP(8);
    }
#ifdef PNX_NNLS_STAMP
    if (lane == 0 && blockIdx.x == 7)
        printf("STAMP setup=%llu w=%llu cand=%llu append=%llu alpha=%llu removal=%llu colpass2=%llu tail=%llu out=%llu\n", seg[0], seg[1], seg[2], seg[3], seg[4], seg[5], seg[6], seg[7], seg[8]);
#endif
}

// ---- Gram step on the matrix cores -----------------------------------------------------------------
// ATY (n_vox, 256) = Y (n_vox, n_meas) * Bp (n_meas, 256) in fp64 on v_mfma_f64_16x16x4_f64 (the right-hand side
// of every voxel's normal equations must be fp64: with an fp32 Gram the NNLS solution is off by O(1), SURVEY.md
// section 7).  One wave owns a strip of 16 voxels x 256 bins: 16 column tiles x (n_meas / 4) MFMAs; Bp is staged
// once per block in LDS ([k][256], fragment reads are 128-byte contiguous per k), Y fragments come straight from
// L2/HBM (4 KB per strip).  Lane maps (cdna_hip_programming.md section 3): A[l & 15][l >> 4], B[l >> 4][l & 15],
// D: col = l & 15, row = (l >> 4) + 4 * reg.
using f64x4 = __attribute__((ext_vector_type(4))) double;

// The product is streamed out once and read once, 2 GB later: nontemporal stores, +9 % (0.503 -> 0.460 ms per 2^20 voxels).
// Regrouping two column tiles with v_permlane32_swap so that a store instruction covers 2 rows x 256 B instead of
// 4 rows x 128 B was measured too: no gain (0.524 ms / 0.478 ms with nontemporal stores).
#ifndef PNX_ATY_NT
#define PNX_ATY_NT 1
#endif
#ifndef PNX_ATY_WAVES
#define PNX_ATY_WAVES 8
#endif
// waves sharing one LDS copy of the basis (64 KB at 32 measurements: 2 workgroups per CU); 4 / 8 / 16 waves:
// 0.477 / 0.439 / 0.436 ms per 2^20 voxels -- more stores in flight per CU, the step is bound by its 2 KB per voxel of output
constexpr int kAtyWaves = PNX_ATY_WAVES;
__device__ __forceinline__ void aty_store(double *p, double v) {
#if PNX_ATY_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__global__ void __launch_bounds__(kAtyWaves * 64) nnls_aty_mfma_kernel(const double *Y, const double *Bp, double *ATY,
                                                            long long n_vox, int nm) {
    extern __shared__ double bsm[];  // [kpad][256]
    const int kpad = (nm + 3) & ~3;
    for (int e = threadIdx.x; e < kpad * kNnlsMaxBins; e += blockDim.x)
        bsm[e] = (e / kNnlsMaxBins) < nm ? Bp[e] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const long long n_strips = (n_vox + 15) / 16;
    for (long long st = (long long)blockIdx.x * kAtyWaves + wave; st < n_strips; st += (long long)gridDim.x * kAtyWaves) {
        const long long v0 = st * 16;
        const long long va = v0 + r16;
        // A fragments for all k-steps: Y[v0 + r16][4 s + kq]
        double afrag[kNnlsMaxMeas / 4];
#pragma unroll
        for (int s4 = 0; s4 < kNnlsMaxMeas / 4; ++s4) {
            const int k = 4 * s4 + kq;
            afrag[s4] = (s4 * 4 < kpad && k < nm && va < n_vox) ? Y[(size_t)va * nm + k] : 0.0;
        }
        for (int tile = 0; tile < kNnlsMaxBins / 16; ++tile) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < kNnlsMaxMeas / 4; ++s4) {
                if (s4 * 4 < kpad) {
                    const double bfrag = bsm[(4 * s4 + kq) * kNnlsMaxBins + tile * 16 + r16];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[s4], bfrag, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long vrow = v0 + kq + 4 * r;
                if (vrow < n_vox) aty_store(ATY + (size_t)vrow * kNnlsMaxBins + tile * 16 + r16, acc[r]);
            }
        }
    }
}

// ---- plan-time kernels ------------------------------------------------------------------------
// G = B^T B + reg^T reg in fp64 (one-off, 2*n^2*(n_meas+n_reg) flop = 35 MFLOP for 250 bins).
__global__ void gram_kernel(const double *B, const double *RT, int nm, int n, int nreg, double *G, int gstride) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n || i >= n) return;
    double acc = 0;
    for (int k = 0; k < nm; ++k) acc += B[(size_t)k * n + i] * B[(size_t)k * n + j];
    for (int r = 0; r < nreg; ++r) acc += RT[(size_t)i * nreg + r] * RT[(size_t)j * nreg + r];
    G[(size_t)i * gstride + j] = acc;
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

// rows 0..kLdsRows-1 of M plus a 64-entry broadcast buffer (a uniform-address ds_read_b64 replaces two v_readlane)
static size_t nnls_lds_bytes() {
    static const size_t pad = dev_getenv("PNX_NNLS_LDS_PAD") ? (size_t)atoi(dev_getenv("PNX_NNLS_LDS_PAD")) : 0;  // occupancy experiments
    return sizeof(double) * (kLdsTri + kW) + pad;
}
static_assert(kLdsTri >= 2 + kNnlsWideBins + 2, "the epilogue's bin-ordered scratch aliases the LDS rows of M");

// Is reg what model_functions/nnls.py:46-85 builds for orders 1-3: square, R[i][j] = c[j - i] inside a band of half width
// <= 2 and exactly zero outside?  Then the regulariser is five numbers and the fast kernel applies it as a stencil.
static bool toeplitz_band(const double *reg, int n_reg, int n_bins, double (&c)[5], int *hb) {
    if (!reg || n_reg != n_bins || n_bins < 5) return false;
    for (int d = -2; d <= 2; ++d) c[d + 2] = reg[(size_t)2 * n_bins + 2 + d];  // row 2 has the whole band
    bool any = false;
    for (int i = 0; i < n_reg; ++i)
        for (int j = 0; j < n_bins; ++j) {
            const int d = j - i;
            const double want = (d >= -2 && d <= 2) ? c[d + 2] : 0.0;
            if (reg[(size_t)i * n_bins + j] != want) return false;
            any = any || want != 0.0;
        }
    *hb = (c[0] != 0.0 || c[4] != 0.0) ? 2 : 1;
    return any;
}


// ---- slabs shared by the block-kernel plans of a device (pnx_nnls.hpp) ---------------------------------------------
namespace {
struct SharedSlabs {
    std::mutex mu;
    int refs = 0;
    double *blk4 = nullptr, *gram = nullptr;
    size_t blk4_bytes = 0, gram_bytes = 0;
    hipEvent_t last = nullptr;       // recorded behind the last launch that used the slabs ...
    hipStream_t last_stream = nullptr;  // ... on this stream
    bool used = false;
};
constexpr int kMaxSharedDevices = 64;
SharedSlabs g_shared[kMaxSharedDevices];
}  // namespace

int nnls_shared_slabs_get(int device, size_t blk4_bytes, size_t gram_bytes, double **blk4, double **gram) {
    if (device < 0 || device >= kMaxSharedDevices) return set_error(PNX_ERR_INVALID, "device %d", device);
    SharedSlabs &S = g_shared[device];
    std::lock_guard<std::mutex> lk(S.mu);
    if (S.blk4 && (S.blk4_bytes < blk4_bytes || S.gram_bytes < gram_bytes)) {
        if (S.refs) return set_error(PNX_ERR_NOMEM, "device %d: shared NNLS slabs are in use at another size", device);
        (void)hipFree(S.blk4);
        (void)hipFree(S.gram);
        S.blk4 = S.gram = nullptr;
    }
    if (!S.blk4) {
        hipError_t e = hipMalloc(&S.blk4, blk4_bytes);
        if (e == hipSuccess) e = hipMemset(S.blk4, 0, blk4_bytes);  // the block sweeps read whole blocks, also rows nobody has written yet: finite
        if (e == hipSuccess) e = hipMalloc(&S.gram, gram_bytes);
        if (e != hipSuccess) {  // all or nothing: a half-made set must not be handed to the next plan
            if (S.blk4) (void)hipFree(S.blk4);
            S.blk4 = S.gram = nullptr;
            return set_error(PNX_ERR_NOMEM, "shared NNLS slabs (%zu + %zu bytes) on device %d: %s", blk4_bytes, gram_bytes, device, hipGetErrorString(e));
        }
        S.blk4_bytes = blk4_bytes;
        S.gram_bytes = gram_bytes;
        if (!S.last) PNX_HIPN(hipEventCreateWithFlags(&S.last, hipEventDisableTiming));
        S.used = false;
    }
    S.refs += 1;
    *blk4 = S.blk4;
    *gram = S.gram;
    return PNX_OK;
}
void nnls_shared_slabs_put(int device) {
    if (device < 0 || device >= kMaxSharedDevices) return;
    std::lock_guard<std::mutex> lk(g_shared[device].mu);
    if (g_shared[device].refs > 0) g_shared[device].refs -= 1;
}
int nnls_shared_slabs_trim(int device) {
    if (device < 0 || device >= kMaxSharedDevices) return 1;
    SharedSlabs &S = g_shared[device];
    std::lock_guard<std::mutex> lk(S.mu);
    if (S.refs) return 0;  // a plan holds them: kept
    if (S.blk4) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        (void)hipFree(S.blk4);  // synchronises the device: nothing enqueued still uses them
        (void)hipFree(S.gram);
        (void)hipSetDevice(cur);
        S.blk4 = S.gram = nullptr;
        S.blk4_bytes = S.gram_bytes = 0;
        S.used = false;
    }
    return 1;
}
NnlsSharedUse::NnlsSharedUse(const NnlsPlanData *P_, hipStream_t stream_) : P(P_ && P_->shared_slabs ? P_ : nullptr), stream(stream_) {
    if (!P) return;
    SharedSlabs &S = g_shared[P->device];
    S.mu.lock();
    if (S.used && S.last_stream != stream) (void)hipStreamWaitEvent(stream, S.last, 0);
}
NnlsSharedUse::~NnlsSharedUse() {
    if (!P) return;
    SharedSlabs &S = g_shared[P->device];
    (void)hipEventRecord(S.last, stream);
    S.last_stream = stream;
    S.used = true;
    S.mu.unlock();
}

int nnls_plan_init(NnlsPlanData *P, int n_meas, int n_bins, const double *basis, const double *reg, int n_reg,
                   int device, int cus) {
    P->device = device;
    P->cus = cus;
    P->n_meas = n_meas;
    P->n_bins = n_bins;
    P->n_reg = n_reg;
    if (!toeplitz_band(reg, n_reg, n_bins, P->rc, &P->rhb) || dev_getenv("PNX_NNLS_GENERIC_REG")) P->rhb = 0;
    {   // reg_order = 0 (the reference's default) is an all-zero matrix: without regulariser rows the Gram form squares a
        // condition number of ~1e16 -- those fits go through the QR-based kernel
        bool zero = true;
        for (size_t i = 0; i < (size_t)n_reg * n_bins && zero; ++i) zero = reg[i] == 0.0;
        // up to 64 measurements with Q and R in LDS, 65 .. 128 with both in a per-wave global slab (pnx_nnls_qr.hip): never a
        // silently different algorithm, and since round 4 no refusal either -- the reference's default has no limit on the
        // number of b-values (nnls_solver.py:37, 88-127)
        P->qr = zero && !dev_getenv("PNX_NNLS_NO_QR");
    }
    // more than 256 bins: the wide instantiations (eight bins per lane) of this file's kernel and of the QR-form kernels; no block
    // kernel (its LDS copy of the basis would leave room for two voxels per CU), no MFMA Gram step (A^T y on the VALU)
    const bool wide = n_bins > kNnlsMaxBins;
    P->bstride = wide ? kNnlsWideBins : kNnlsMaxBins;
    const size_t bs = (size_t)P->bstride;
    const size_t nb = (size_t)n_meas * n_bins, nr = (size_t)n_reg * n_bins, ng = (bs + 1) * bs;  // one row more: the block kernel gathers column 256 (its padding bin) of a row
    for (size_t i = 0; i < nb; ++i)
        if (!std::isfinite(basis[i])) return set_error(PNX_ERR_INVALID, "basis contains non-finite values");
    PNX_HIPN(hipMalloc(&P->B, nb * sizeof(double)));
    PNX_HIPN(hipMalloc(&P->Bp, (size_t)n_meas * bs * sizeof(double)));
    PNX_HIPN(hipMemset(P->Bp, 0, (size_t)n_meas * bs * sizeof(double)));
    PNX_HIPN(hipMalloc(&P->RT, (nr ? nr : 1) * sizeof(double)));
    PNX_HIPN(hipMalloc(&P->G, ng * sizeof(double)));
    PNX_HIPN(hipMemset(P->G, 0, ng * sizeof(double)));  // rows padded to bstride columns
    PNX_HIPN(hipMalloc(&P->queue, sizeof(unsigned long long)));
    PNX_HIPN(hipMemcpy(P->B, basis, nb * sizeof(double), hipMemcpyHostToDevice));
    PNX_HIPN(hipMemcpy2D(P->Bp, bs * sizeof(double), basis, (size_t)n_bins * sizeof(double), (size_t)n_bins * sizeof(double), n_meas, hipMemcpyHostToDevice));
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
                       n_reg, P->G, P->bstride);
    PNX_HIPN(hipGetLastError());
    if (!P->qr && !wide && nnls_blk_applicable(P)) {
        P->blk = true;
        const int rc_ = nnls_blk_plan_init(P);
        if (rc_ != PNX_OK) return rc_;
    }
    // persistent grid: as many single-wave workgroups as fit (LDS bound), one scratch slab each
    int occ = 0;
    if (wide) {
        PNX_HIPN(hipFuncSetAttribute((const void *)nnls_kernel<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nnls_lds_bytes()));
        PNX_HIPN(hipFuncSetAttribute((const void *)nnls_kernel<6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nnls_lds_bytes()));
        PNX_HIPN(hipFuncSetAttribute((const void *)nnls_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nnls_lds_bytes()));
        if (n_bins <= 6 * kW)
            PNX_HIPN(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nnls_kernel<6, 4>, kW, nnls_lds_bytes()));
        else
            PNX_HIPN(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nnls_kernel<8, 4>, kW, nnls_lds_bytes()));
    } else {
        PNX_HIPN(hipFuncSetAttribute((const void *)nnls_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nnls_lds_bytes()));
        PNX_HIPN(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nnls_kernel<4, 4>, kW, nnls_lds_bytes()));
    }
    if (occ < 1) return set_error(PNX_ERR_HIP, "nnls kernel does not fit on a CU");
    P->n_waves = occ * cus;
    P->mglob_stride = glob_tri<4>();  // 256 KB per resident wave: rows 48 .. 255 of M (both first-pass kernels keep 256 positions)
    // a block-kernel plan runs this kernel on the voxels handed over and, when the pilot of a call finds that too many are (strong
    // regularisers: supports beyond 128 bins), on the whole call (A^T y on the VALU either way): the full grid of slabs (1 GB),
    // but no 2 GiB chunk buffer for the Gram step unless pnx_nnls_aty asks for one later
    if (P->qr) P->n_waves = 1;  // a QR-form plan never launches this kernel
    if (P->blk) {  // the slabs of the kernels behind the first pass: one set per device (pnx_nnls.hpp)
        const int rs = nnls_shared_slabs_get(device, nnls_blk4_slab_bytes(P), (size_t)P->n_waves * P->mglob_stride * sizeof(double), &P->Mblk4, &P->Mglob);
        if (rs != PNX_OK) return rs;
        P->shared_slabs = true;
    } else
        PNX_HIPN(hipMalloc(&P->Mglob, (size_t)P->n_waves * P->mglob_stride * sizeof(double)));
    if (wide && !P->qr) {  // the hand-over pass of a wide plan (passive sets beyond 256 positions): one wave per CU, 1 MB of slab each
        P->wide_waves = cus;
        PNX_HIPN(hipMalloc(&P->Mwide, (size_t)P->wide_waves * glob_tri<8>() * sizeof(double)));
        P->blk_bail_cap = (size_t)kAtyChunk;
        PNX_HIPN(hipMalloc(&P->blk_bail, (P->blk_bail_cap + 1) * sizeof(int32_t)));
    }
    P->mfma_ok = !dev_getenv("PNX_NNLS_NO_MFMA") && n_meas <= 64 && !wide;  // LDS stage of Bp: n_meas * 2 KiB
    if (P->mfma_ok) {
        if (!P->blk) PNX_HIPN(hipMalloc(&P->aty, (size_t)kAtyChunk * kNnlsMaxBins * sizeof(double)));
        PNX_HIPN(hipFuncSetAttribute((const void *)nnls_aty_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    PNX_HIPN(hipDeviceSynchronize());
    return PNX_OK;
}

void nnls_plan_free(NnlsPlanData *P) {
    if (P->B) (void)hipFree(P->B);
    if (P->Bp) (void)hipFree(P->Bp);
    if (P->RT) (void)hipFree(P->RT);
    if (P->G) (void)hipFree(P->G);
    if (P->shared_slabs)
        nnls_shared_slabs_put(P->device);
    else {
        if (P->Mglob) (void)hipFree(P->Mglob);
        if (P->Mblk4) (void)hipFree(P->Mblk4);
    }
    if (P->Mblk) (void)hipFree(P->Mblk);
    if (P->blk4_bail) (void)hipFree(P->blk4_bail);
    if (P->Mwide) (void)hipFree(P->Mwide);
    if (P->qr_slab) (void)hipFree(P->qr_slab);
    if (P->blk_bail) (void)hipFree(P->blk_bail);
    if (P->route) (void)hipFree(P->route);
    if (P->aty) (void)hipFree(P->aty);
    if (P->queue) (void)hipFree(P->queue);
    *P = NnlsPlanData();
}

int nnls_solve_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d,
                      double *rnorm_d, int8_t *status_d, int32_t *iters_d, hipStream_t stream) {
    // Voxels go through in chunks of kAtyChunk: the MFMA Gram step fills ATY for the chunk, the persistent
    // active-set kernel consumes it (256 voxels per resident wave at full occupancy keep the drain tail small; 2 GiB of ATY scratch).
    if (P->qr) return nnls_qr_solve_device(P, n_vox, y_d, max_iter, coeff_d, rnorm_d, status_d, iters_d, stream);
    if (P->blk) return nnls_blk_solve_device(P, n_vox, y_d, max_iter, coeff_d, rnorm_d, status_d, iters_d, stream);
    const bool use_mfma = P->aty != nullptr;
    for (int64_t off = 0; off < n_vox; off += kAtyChunk) {
        const int64_t c = (n_vox - off) < kAtyChunk ? (n_vox - off) : kAtyChunk;
        NnlsArgs a;
        a.y = y_d + (size_t)off * P->n_meas;
        a.coeff = coeff_d + (size_t)off * P->n_bins;
        a.rnorm = rnorm_d + off;
        a.status = status_d ? status_d + off : nullptr;
        a.iters = iters_d ? iters_d + off : nullptr;
        a.G = P->G;
        a.Bp = P->Bp;
        a.RT = P->RT;
        a.aty = use_mfma ? P->aty : nullptr;
        a.Mglob = P->Mglob;
        a.queue = P->queue;
        a.n_vox = c;
        a.n_meas = P->n_meas;
        a.n_bins = P->n_bins;
        a.n_reg = P->n_reg;
        a.max_iter = max_iter;
        for (int k = 0; k < 5; ++k) a.rc[k] = P->rc[k];
        a.rhb = P->rhb;
        a.redo_list = a.redo_count = nullptr;
        a.bail = nullptr;
        if (use_mfma) {
            const int kpad = (P->n_meas + 3) & ~3;
            const size_t lds = (size_t)kpad * kNnlsMaxBins * sizeof(double);
            const long long strips = (c + 15) / 16;
            long long grid = (strips + kAtyWaves - 1) / kAtyWaves;
            const long long cap = (long long)P->cus * 2;
            if (grid > cap) grid = cap;
            hipLaunchKernelGGL(nnls_aty_mfma_kernel, dim3((unsigned)grid), dim3(kAtyWaves * 64), lds, stream, a.y, P->Bp, P->aty,
                               (long long)c, P->n_meas);
            PNX_HIPN(hipGetLastError());
        }
        PNX_HIPN(hipMemsetAsync(P->queue, 0, sizeof(unsigned long long), stream));
        long long grid = c < P->n_waves ? c : P->n_waves;
        if (P->bstride == kNnlsWideBins) {
            // first pass: 256 positions per voxel; the voxels that want more go to the list and through <8, 8> right behind it
            // (stream order; the count stays on the device: an empty list costs the second launch one queue pull per wave)
            a.bail = P->blk_bail;
            PNX_HIPN(hipMemsetAsync(P->blk_bail, 0, sizeof(int32_t), stream));
            if (P->n_bins <= 6 * kW)
                hipLaunchKernelGGL((nnls_kernel<6, 4>), dim3((unsigned)grid), dim3(kW), nnls_lds_bytes(), stream, a);
            else
                hipLaunchKernelGGL((nnls_kernel<8, 4>), dim3((unsigned)grid), dim3(kW), nnls_lds_bytes(), stream, a);
            PNX_HIPN(hipGetLastError());
            PNX_HIPN(hipMemsetAsync(P->queue, 0, sizeof(unsigned long long), stream));
            a.bail = nullptr;
            a.redo_list = P->blk_bail + 1;
            a.redo_count = P->blk_bail;
            a.Mglob = P->Mwide;
            long long g2 = c < P->wide_waves ? c : P->wide_waves;
            hipLaunchKernelGGL((nnls_kernel<8, 8>), dim3((unsigned)g2), dim3(kW), nnls_lds_bytes(), stream, a);
        } else
            hipLaunchKernelGGL((nnls_kernel<4, 4>), dim3((unsigned)grid), dim3(kW), nnls_lds_bytes(), stream, a);
        PNX_HIPN(hipGetLastError());
    }
    return PNX_OK;
}

int nnls_redo_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                     int8_t *status_d, int32_t *iters_d, const int32_t *list, const int32_t *count, hipStream_t stream) {
    NnlsArgs a;
    a.y = y_d;
    a.coeff = coeff_d;
    a.rnorm = rnorm_d;
    a.status = status_d;
    a.iters = iters_d;
    a.G = P->G;
    a.Bp = P->Bp;
    a.RT = P->RT;
    a.aty = nullptr;
    a.Mglob = P->Mglob;
    a.queue = P->queue;
    a.n_vox = n_vox;
    a.n_meas = P->n_meas;
    a.n_bins = P->n_bins;
    a.n_reg = P->n_reg;
    a.max_iter = max_iter;
    for (int k = 0; k < 5; ++k) a.rc[k] = P->rc[k];
    a.rhb = P->rhb;
    a.redo_list = list;
    a.redo_count = count;
    a.bail = nullptr;
    NnlsSharedUse use(P, stream);  // block-kernel plans: Mglob is the device's shared slab
    PNX_HIPN(hipMemsetAsync(P->queue, 0, sizeof(unsigned long long), stream));
    // the plan's persistent grid: with an empty list a wave costs one queue pull
    long long grid = P->n_waves;
    if (grid > n_vox) grid = n_vox;
    hipLaunchKernelGGL((nnls_kernel<4, 4>), dim3((unsigned)grid), dim3(kW), nnls_lds_bytes(), stream, a);  // block-kernel plans only: never wide
    PNX_HIPN(hipGetLastError());
    return PNX_OK;
}

int nnls_routed_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                       int8_t *status_d, int32_t *iters_d, const int32_t *route, hipStream_t stream) {
    NnlsArgs a;
    a.y = y_d;
    a.coeff = coeff_d;
    a.rnorm = rnorm_d;
    a.status = status_d;
    a.iters = iters_d;
    a.G = P->G;
    a.Bp = P->Bp;
    a.RT = P->RT;
    a.aty = nullptr;
    a.Mglob = P->Mglob;
    a.queue = P->queue;
    a.n_vox = n_vox;
    a.n_meas = P->n_meas;
    a.n_bins = P->n_bins;
    a.n_reg = P->n_reg;
    a.max_iter = max_iter;
    for (int k = 0; k < 5; ++k) a.rc[k] = P->rc[k];
    a.rhb = P->rhb;
    a.redo_list = a.redo_count = nullptr;
    a.bail = nullptr;
    a.route = route;
    NnlsSharedUse use(P, stream);
    PNX_HIPN(hipMemsetAsync(P->queue, 0, sizeof(unsigned long long), stream));
    long long grid = P->n_waves;
    if (grid > n_vox) grid = n_vox;
    hipLaunchKernelGGL((nnls_kernel<4, 4>), dim3((unsigned)grid), dim3(kW), nnls_lds_bytes(), stream, a);  // block-kernel plans only: never wide
    PNX_HIPN(hipGetLastError());
    return PNX_OK;
}

int nnls_aty_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, double *aty_d, hipStream_t stream) {
    if (!P->mfma_ok) return set_error(PNX_ERR_UNSUPPORTED, "the MFMA Gram step is disabled for this plan (n_meas=%d)", P->n_meas);
    if (!aty_d && !P->aty) PNX_HIPN(hipMalloc(&P->aty, (size_t)kAtyChunk * kNnlsMaxBins * sizeof(double)));  // first use by a block-kernel plan
    if (n_vox > kAtyChunk) return set_error(PNX_ERR_INVALID, "n_vox=%lld > %lld per call", (long long)n_vox, (long long)kAtyChunk);
    const int kpad = (P->n_meas + 3) & ~3;
    const size_t lds = (size_t)kpad * kNnlsMaxBins * sizeof(double);
    const long long strips = (n_vox + 15) / 16;
    long long grid = (strips + kAtyWaves - 1) / kAtyWaves;
    const long long cap = (long long)P->cus * 2;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(nnls_aty_mfma_kernel, dim3((unsigned)grid), dim3(kAtyWaves * 64), lds, stream, y_d, P->Bp, aty_d ? aty_d : P->aty,
                       (long long)n_vox, P->n_meas);
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
