// pnx_nnls_qr.hip -- Lawson-Hanson NNLS for the UNREGULARISED case (reg_order = 0, the reference's default:
// src/pyneapple/solvers/nnls_solver.py:37; model_functions/nnls.py:62-64 returns a zero matrix), fp64.
//
// Why a second kernel: without regulariser rows A = B is n_meas x n_bins with n_meas << n_bins (32 x 250) and numerical
// rank ~25; which columns may enter the passive set is decided by how far a column is from the span of the passive
// ones.  The Gram-form kernel (pnx_nnls.hip) measures that distance as sqrt(G_jj - |l|^2) -- cond(A)^2 -- and picks a
// different, equally feasible column than SciPy on a few per cent of such voxels (fixture g9_nnls_250_r0).  Here the
// passive set lives in an explicit orthonormal basis Q (n_meas x p, p <= n_meas) and a triangular R, like SciPy's own
// QR-based kernel: entering columns are orthogonalised twice (Gram-Schmidt with re-orthogonalisation, error ~ eps *
// cond(A)), leaving columns are removed with Givens rotations (LAPACK dlartgp convention: r >= 0).  The
// problem is tiny (a 32 x p factor), so this costs a fraction of the regularised solve.
//
// One wavefront owns one voxel.  lane = measurement for vectors in R^m (residual, candidate column, columns of Q),
// lane = passive position for x / z / Q^T b / the diagonal of R, four bins per lane for the dual (same bin ownership
// as pnx_nnls.hip).  Q is stored position major (Qt[i][m], row stride MP + 1: contiguous for lane = m, bank-conflict
// free for lane = i), R column major (Rc[k][i] = R[i][k]): the column that is appended is exactly the vector l = Q^T a.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "pnx_internal.hpp"
#include "pnx_nnls.hpp"

namespace pnx {

namespace {
constexpr int kW = 64;
constexpr int kNone = 1 << 30;

struct QrArgs {
    const double *y;
    double *coeff;
    double *rnorm;
    int8_t *status;
    int32_t *iters;
    const double *Bp;  // (n_meas, 64 KB) zero padded rows
    long long n_vox;
    int n_meas, n_bins, max_iter;
};

__device__ inline int binof(int lane, int s) { return ((s >> 1) << 7) + 2 * lane + (s & 1); }
__device__ inline double rl(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ inline int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }  // compiler-only fence (see pnx_nnls.hip)
// Every branch of this kernel is wave uniform by construction (one wave owns one voxel), but the values the decisions
// are taken on come out of cross-lane reductions and live in vector registers: unless told otherwise the compiler
// builds EXEC-masked "divergent" loops around code that contains further cross-lane operations (a first build hung in
// exactly such a loop).  uni() moves a decision to a scalar register, so that the branch is a scalar branch.
__device__ __forceinline__ bool uni(bool b) { return __builtin_amdgcn_readfirstlane(b ? 1 : 0) != 0; }
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Givens rotation with non-negative r (LAPACK dlartgp, what SciPy 1.15 links)
__device__ inline void givens(double f, double g, double &c, double &s, double &r) {
    if (g == 0) {
        c = copysign(1.0, f);
        s = 0;
        r = fabs(f);
    } else if (f == 0) {
        c = 0;
        s = copysign(1.0, g);
        r = fabs(g);
    } else {
        r = hypot(f, g);
        c = f / r;
        s = g / r;
    }
}

// KB: bins per lane -- 4 up to 256 bins, 8 for the wide plans (257 .. 512 bins; the bins only appear in the dual, the arg-max,
// the passive flags and the output)
template <int MP, int KB> __global__ void __launch_bounds__(kW) nnls_qr_kernel(const QrArgs A) {
    constexpr int ST = MP + 1;
    constexpr int kBS = kW * KB;
    extern __shared__ double lds[];
    double *Qt = lds;             // [MP][ST]  Qt[i][m]: column i of Q
    double *Rc = Qt + MP * ST;    // [MP][ST]  Rc[k][i] = R[i][k]
    double *abuf = Rc + MP * ST;  // [64] broadcast buffer (by measurement)
    double *lbuf = abuf + kW;     // [64] broadcast buffer (by position)
    const int lane = threadIdx.x;
    const int n = A.n_bins, nm = A.n_meas;

    // grid-stride over the voxels (a uniform, scalar loop bound; the solves are short and the assignment is interleaved, so
    // a work queue buys nothing here)
    for (long long vox = blockIdx.x; vox < A.n_vox; vox += gridDim.x) {
        const double yv = lane < nm ? A.y[(size_t)vox * nm + lane] : 0.0;
        const bool finite = __all(isfinite(yv) ? 1 : 0) != 0;
        const double yn2 = wave_sum(yv * yv);
        double res = yv;                      // y - Q Q^T y, by measurement
        double xpos = 0, z = 0, qtb = 0, diag = 1;  // by position
        int pidx = 0;
        bool inP[KB] = {};
        int p = 0, iteration = 0, status = finite ? 1 : -2;

        int guard = 0;  // belt and braces: every pass of the loops below is counted; a voxel that exceeds any sane count stops
        const int guard_max = 8 * A.max_iter + 4 * n + 64;
        while (status == 1 && p < n && p < nm) {
            if (++guard > guard_max) {
                status = 0;
                break;
            }
#ifdef PNX_QR_TRACE
            if (lane == 0 && guard < 40) printf("O vox=%lld p=%d it=%d guard=%d\n", vox, p, iteration, guard);
#endif
            // ---- dual w = B^T res on the zero set
            double w[KB] = {};
#pragma unroll 4
            for (int k = 0; k < nm; ++k) {
                const double rk = rl(res, k);
                const double *br = A.Bp + (size_t)k * kBS + 2 * lane;
                double2 bh[KB / 2];
#pragma unroll
                for (int h = 0; h < KB / 2; ++h) bh[h] = *reinterpret_cast<const double2 *>(br + 128 * h);
#pragma unroll
                for (int h = 0; h < KB / 2; ++h) {
                    w[2 * h] = fma(bh[h].x, rk, w[2 * h]);
                    w[2 * h + 1] = fma(bh[h].y, rk, w[2 * h + 1]);
                }
            }
#pragma unroll
            for (int s = 0; s < KB; ++s)
                if (inP[s] || binof(lane, s) >= n) w[s] = -INFINITY;

            bool accepted = false;
            int jmax = 0;
            double lam = 0, qn = 0, l = 0, v = 0;
            for (;;) {
                if (++guard > guard_max) break;
                double best = w[0];
#pragma unroll
                for (int s = 1; s < KB; ++s) best = fmax(best, w[s]);
                best = wave_max(best);
                if (uni(!(best > 0))) break;  // KKT satisfied
                int bj = kNone;
#pragma unroll
                for (int s = KB - 1; s >= 0; --s)
                    if (w[s] == best) bj = binof(lane, s);
                jmax = uni_i(wave_min_i(bj));
#ifdef PNX_QR_TRACE
                if (lane == 0 && guard < 40) printf("C vox=%lld best=%g jmax=%d\n", vox, best, jmax);
#endif
                if (jmax >= n) break;  // cannot happen (some lane owns the maximum); never index B with it
                // ---- candidate column a (by measurement), orthogonalised against Q twice
                v = lane < nm ? A.Bp[(size_t)lane * kBS + jmax] : 0.0;
                l = 0;
                for (int pass = 0; pass < 2; ++pass) {
                    lds_order();
                    abuf[lane] = v;
                    lds_order();
                    double li = 0;  // (Q^T v)_i, lane = position
                    if (lane < p) {
                        const double *qi = Qt + lane * ST;
#pragma unroll 8
                        for (int m = 0; m < nm; ++m) li = fma(qi[m], abuf[m], li);
                    }
                    lds_order();
                    lbuf[lane] = li;
                    lds_order();
                    if (lane < MP)  // rows of Qt / Rc hold MP entries: lanes beyond stay at v = 0
                        for (int i = 0; i < p; ++i) v = fma(-Qt[i * ST + lane], lbuf[i], v);  // lanes >= nm hold zeros of Q
                    l += li;
                }
                const double ll = wave_sum(lane < p ? l * l : 0.0);
                const double vv = wave_sum(v * v);
                lam = sqrt(vv);
                const double un = sqrt(ll);
                const double vr = wave_sum(v * res);  // v is orthogonal to Q: v^T b = v^T res
                bool ok = uni(((un + lam * 0.01) - un) > 0);  // Lawson-Hanson linear-independence test
                if (ok) {
                    qn = vr / lam;
                    ok = uni((qn / lam) > 0);  // ztest
                }
                if (ok) {
                    accepted = true;
                    break;
                }
#pragma unroll
                for (int s = 0; s < KB; ++s)
                    if (binof(lane, s) == jmax) w[s] = 0.0;  // reject: look for the next largest
            }
            if (!accepted) break;

            // ---- column jmax enters at position p
            {
                const double qv = v / lam;
                lds_order();
                if (lane < MP) {
                    Qt[p * ST + lane] = qv;  // lanes >= nm: v = 0
                    Rc[p * ST + lane] = lane < p ? l : (lane == p ? lam : 0.0);
                }
                lds_order();
                res = fma(-qv, qn, res);
                if (lane == p) {
                    qtb = qn;
                    diag = lam;
                    pidx = jmax;
                    xpos = 0;
                }
#pragma unroll
                for (int s = 0; s < KB; ++s)
                    if (binof(lane, s) == jmax) inP[s] = true;
                p += 1;
            }

            // ---- inner loop
            for (;;) {
                if (status != 1) break;
                // z = R^{-1} Q^T b: back substitution over the columns of R
                {
                    double t = qtb;
                    for (int k = p - 1; k >= 0; --k) {
                        const double zk = rl(t, k) / rl(diag, k);
                        if (lane == k) z = zk;
                        if (lane < k) t = fma(-Rc[k * ST + lane], zk, t);
                    }
                }
#ifdef PNX_QR_TRACE
                if (lane == 0 && guard < 40) printf("I vox=%lld p=%d it=%d z0=%g\n", vox, p, iteration, z);
#endif
                iteration += 1;
                if (iteration == A.max_iter) {
                    status = 0;
                    break;
                }
                const bool viol = lane < p && z <= 0;
                if (!__any(viol ? 1 : 0)) {
                    if (lane < p) xpos = z;
                    break;
                }
                double T = viol ? -xpos / (z - xpos) : INFINITY;
                const double alpha = -wave_max(-T);
                int jj = uni_i(wave_min_i((viol && T == alpha) ? lane : kNone));  // first position with the minimum
                if (lane < p) xpos = xpos + alpha * (z - xpos);
                for (;;) {
                    if (++guard > guard_max || jj >= p) {
                        status = 0;
                        break;
                    }
                    // ---- position jj leaves: delete column jj of R, restore the triangle with Givens rotations on the
                    // rows (i, i + 1), i = jj .. p - 2, and rotate the columns of Q and the entries of Q^T b with them
                    const int bin_out = __builtin_amdgcn_readlane(pidx, jj);
                    lds_order();
                    for (int k = jj; k < p - 1; ++k) {
                        const double cv = lane < MP ? Rc[(k + 1) * ST + lane] : 0.0;
                        lds_order();
                        if (lane < MP) Rc[k * ST + lane] = cv;
                        lds_order();
                    }
                    for (int i = jj; i < p - 1; ++i) {
                        lds_order();
                        const double f = Rc[i * ST + i], g = Rc[i * ST + i + 1];  // uniform reads
                        double c, s, r;
                        givens(f, g, c, s, r);
                        if (lane >= i && lane < p - 1) {  // lane = column k
                            const double a = Rc[lane * ST + i], b = Rc[lane * ST + i + 1];
                            Rc[lane * ST + i] = lane == i ? r : c * a + s * b;
                            Rc[lane * ST + i + 1] = lane == i ? 0.0 : c * b - s * a;
                        }
                        if (lane < MP) {  // columns i, i + 1 of Q (lane = measurement)
                            const double a = Qt[i * ST + lane], b = Qt[(i + 1) * ST + lane];
                            Qt[i * ST + lane] = c * a + s * b;
                            Qt[(i + 1) * ST + lane] = c * b - s * a;
                        }
                        const double t1 = rl(qtb, i), t2 = rl(qtb, i + 1);
                        if (lane == i) {
                            qtb = c * t1 + s * t2;
                            diag = r;
                        }
                        if (lane == i + 1) qtb = c * t2 - s * t1;
                        lds_order();
                    }
                    // the last (rotated) direction leaves the basis: its share of b goes back into the residual
                    res = fma(lane < MP ? Qt[(p - 1) * ST + lane] : 0.0, rl(qtb, p - 1), res);
                    {  // positions above jj move down by one
                        const double xs = __shfl_down(xpos, 1);
                        const int ps = __shfl_down(pidx, 1);
                        if (lane >= jj && lane < p - 1) {
                            xpos = xs;
                            pidx = ps;
                        }
                    }
#pragma unroll
                    for (int s = 0; s < KB; ++s)
                        if (binof(lane, s) == bin_out) inP[s] = false;
                    p -= 1;
                    // round-off clean-up: any remaining x <= 0 leaves too (first position first)
                    const int bad = uni_i(wave_min_i((lane < p && xpos <= 0) ? lane : kNone));
                    if (bad == kNone) break;
                    jj = bad;
                }
            }
        }

#ifdef PNX_QR_TRACE
        if (lane == 0) printf("E vox=%lld status=%d it=%d p=%d guard=%d\n", vox, status, iteration, p, guard);
#endif
        // ---- outputs
        double xb[KB] = {};
        if (status == 1) {
            for (int i = 0; i < p; ++i) {
                const int b = __builtin_amdgcn_readlane(pidx, i);
                const double xv = rl(xpos, i);
#pragma unroll
                for (int s = 0; s < KB; ++s)
                    if (binof(lane, s) == b) xb[s] = xv;
            }
        }
#ifdef PNX_QR_TRACE
        if (lane == 0) printf("1 vox=%lld coeff=%p n=%d\n", vox, (void *)A.coeff, n);
#endif
        double *cv = A.coeff + (size_t)vox * n;
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            const int j = binof(lane, s);
            if (j < n) cv[j] = xb[s];
        }
#ifdef PNX_QR_TRACE
        if (lane == 0) printf("2 vox=%lld\n", vox);
#endif
        const double r2 = wave_sum(res * res);
#ifdef PNX_QR_TRACE
        if (lane == 0) printf("3 vox=%lld r2=%g\n", vox, r2);
#endif
        const double rn = sqrt(status == 1 ? r2 : yn2);  // failure: zeros, ||y|| (nnls_solver.py:205-210)
#ifdef PNX_QR_TRACE
        if (lane == 0) printf("4 vox=%lld rn=%g rnorm=%p status=%p iters=%p\n", vox, rn, (void *)A.rnorm, (void *)A.status, (void *)A.iters);
#endif
        if (lane == 0) {
            A.rnorm[vox] = rn;
            if (A.status) A.status[vox] = (int8_t)status;
            if (A.iters) A.iters[vox] = iteration;
        }
#ifdef PNX_QR_TRACE
        if (lane == 0) printf("W vox=%lld written\n", vox);
#endif
    }
#ifdef PNX_QR_TRACE
    if (lane == 0) printf("X block=%d exit\n", blockIdx.x);
#endif
}

template <int MP> size_t qr_lds_bytes() { return sizeof(double) * (2 * MP * (MP + 1) + 2 * kW); }

template <int MP, int KB> int launch_qr(NnlsPlanData *P, const QrArgs &a, hipStream_t stream) {
    static bool attr_done[64] = {false};
    auto kern = nnls_qr_kernel<MP, KB>;
    if (!attr_done[P->device & 63]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qr_lds_bytes<MP>()) != hipSuccess)
            return set_error(PNX_ERR_HIP, "hipFuncSetAttribute(nnls_qr_kernel) failed");
        attr_done[P->device & 63] = true;
    }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kW, qr_lds_bytes<MP>()) != hipSuccess || occ < 1)
        return set_error(PNX_ERR_HIP, "nnls_qr_kernel does not fit on a CU");
    long long grid = (long long)occ * P->cus;
    if (grid > a.n_vox) grid = a.n_vox;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kW), qr_lds_bytes<MP>(), stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PNX_ERR_HIP, "nnls_qr launch: %s", hipGetErrorString(e));
    return PNX_OK;
}

// ---- 65 .. 128 measurements (round 4) ---------------------------------------------------------------------------
// The reference's default regulariser (reg_order = 0) has no limit on the number of b-values (nnls_solver.py:37, 88-127).  With
// more than 64 of them a measurement-indexed vector takes two register slots per lane, and Q and R (128 x 129 doubles each) no
// longer fit the LDS of a CU: they live in a per-wave slab in global memory (zero pages until touched; the passive set of an
// unregularised fit is small -- 5.5 bins on average on the reference workload -- so the rows in use stay in L2).  Same
// algorithm, same decisions and the same order of every floating-point sum as the kernel above wherever a sum is not a wave
// reduction; (Q^T v)_i is a wave reduction here (the rows of Q are read coalesced) where the LDS kernel walks a row per lane,
// so the two differ by rounding in l = Q^T v -- as either does from SciPy's Householder form.
constexpr int kQS = 2;             // register slots of a vector by measurement / by position
constexpr int kQM = kQS * kW;      // 128
constexpr int kQST = kQM + 1;
constexpr size_t kQrSlab = 2 * (size_t)kQM * kQST;  // doubles per wave: Qt | Rc

__device__ __forceinline__ double rlk(const double (&a)[kQS], int k) { return (k >> 6) ? rl(a[1], k & 63) : rl(a[0], k & 63); }
__device__ __forceinline__ int rlk_i(const int (&a)[kQS], int k) {
    return (k >> 6) ? __builtin_amdgcn_readlane(a[1], k & 63) : __builtin_amdgcn_readlane(a[0], k & 63);
}

struct QrBigArgs {
    QrArgs a;
    double *slab;  // kQrSlab doubles per workgroup
};

template <int KB> __global__ void __launch_bounds__(kW) nnls_qr_big_kernel(const QrBigArgs BA) {
    constexpr int kBS = kW * KB;
    const QrArgs &A = BA.a;
    __shared__ double lbuf[kQM];  // (Q^T v) by position, read back with uniform addresses
    double *Qt = BA.slab + (size_t)blockIdx.x * kQrSlab;  // [kQM][kQST]  Qt[i][m]: column i of Q
    double *Rc = Qt + (size_t)kQM * kQST;                 // [kQM][kQST]  Rc[k][i] = R[i][k]
    const int lane = threadIdx.x;
    const int n = A.n_bins, nm = A.n_meas;
    auto sync = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };

    for (long long vox = blockIdx.x; vox < A.n_vox; vox += gridDim.x) {
        double yv[kQS], res[kQS], xpos[kQS], z[kQS], qtb[kQS], diag[kQS];
        int pidx[kQS];
        double y2 = 0;
        bool fin = true;
#pragma unroll
        for (int s = 0; s < kQS; ++s) {
            const int m = lane + kW * s;
            yv[s] = m < nm ? A.y[(size_t)vox * nm + m] : 0.0;
            fin = fin && isfinite(yv[s]);
            y2 = fma(yv[s], yv[s], y2);
            res[s] = yv[s];
            xpos[s] = z[s] = qtb[s] = 0;
            diag[s] = 1;
            pidx[s] = 0;
        }
        const bool finite = __all(fin ? 1 : 0) != 0;
        const double yn2 = wave_sum(y2);
        bool inP[KB] = {};
        int p = 0, iteration = 0, status = finite ? 1 : -2;
        int guard = 0;
        const int guard_max = 8 * A.max_iter + 4 * n + 64;
        while (status == 1 && p < n && p < nm) {
            if (++guard > guard_max) {
                status = 0;
                break;
            }
            // ---- dual w = B^T res on the zero set
            double w[KB] = {};
#pragma unroll 4
            for (int k = 0; k < nm; ++k) {
                const double rk = rlk(res, k);
                const double *br = A.Bp + (size_t)k * kBS + 2 * lane;
                double2 bh[KB / 2];
#pragma unroll
                for (int h = 0; h < KB / 2; ++h) bh[h] = *reinterpret_cast<const double2 *>(br + 128 * h);
#pragma unroll
                for (int h = 0; h < KB / 2; ++h) {
                    w[2 * h] = fma(bh[h].x, rk, w[2 * h]);
                    w[2 * h + 1] = fma(bh[h].y, rk, w[2 * h + 1]);
                }
            }
#pragma unroll
            for (int s = 0; s < KB; ++s)
                if (inP[s] || binof(lane, s) >= n) w[s] = -INFINITY;

            bool accepted = false;
            int jmax = 0;
            double lam = 0, qn = 0;
            double l[kQS] = {0, 0}, v[kQS] = {0, 0};
            for (;;) {
                if (++guard > guard_max) break;
                double best = w[0];
#pragma unroll
                for (int s = 1; s < KB; ++s) best = fmax(best, w[s]);
                best = wave_max(best);
                if (uni(!(best > 0))) break;  // KKT satisfied
                int bj = kNone;
#pragma unroll
                for (int s = KB - 1; s >= 0; --s)
                    if (w[s] == best) bj = binof(lane, s);
                jmax = uni_i(wave_min_i(bj));
                if (jmax >= n) break;
                // ---- candidate column a (by measurement), orthogonalised against Q twice
#pragma unroll
                for (int s = 0; s < kQS; ++s) {
                    const int m = lane + kW * s;
                    v[s] = m < nm ? A.Bp[(size_t)m * kBS + jmax] : 0.0;
                    l[s] = 0;
                }
                for (int pass = 0; pass < 2; ++pass) {
                    sync();
                    for (int i = 0; i < p; ++i) {  // (Q^T v)_i: a coalesced row of Q against v, summed over the wave
                        double part = 0;
#pragma unroll
                        for (int s = 0; s < kQS; ++s) part = fma(Qt[(size_t)i * kQST + lane + kW * s], v[s], part);  // measurements >= nm hold zeros
                        const double li = wave_sum(part);
                        if (lane == 0) lbuf[i] = li;
                    }
                    sync();
#pragma unroll
                    for (int s = 0; s < kQS; ++s) {
                        const int i = lane + kW * s;
                        if (i < p) l[s] += lbuf[i];
                    }
                    for (int i = 0; i < p; ++i) {
                        const double li = lbuf[i];
#pragma unroll
                        for (int s = 0; s < kQS; ++s) v[s] = fma(-Qt[(size_t)i * kQST + lane + kW * s], li, v[s]);
                    }
                }
                double l2 = 0, v2 = 0, vres = 0;
#pragma unroll
                for (int s = 0; s < kQS; ++s) {
                    if (lane + kW * s < p) l2 = fma(l[s], l[s], l2);
                    v2 = fma(v[s], v[s], v2);
                    vres = fma(v[s], res[s], vres);
                }
                const double ll = wave_sum(l2);
                const double vv = wave_sum(v2);
                lam = sqrt(vv);
                const double un = sqrt(ll);
                const double vr = wave_sum(vres);  // v is orthogonal to Q: v^T b = v^T res
                bool ok = uni(((un + lam * 0.01) - un) > 0);  // Lawson-Hanson linear-independence test
                if (ok) {
                    qn = vr / lam;
                    ok = uni((qn / lam) > 0);  // ztest
                }
                if (ok) {
                    accepted = true;
                    break;
                }
#pragma unroll
                for (int s = 0; s < KB; ++s)
                    if (binof(lane, s) == jmax) w[s] = 0.0;  // reject: look for the next largest
            }
            if (!accepted) break;

            // ---- column jmax enters at position p
            {
                sync();
#pragma unroll
                for (int s = 0; s < kQS; ++s) {
                    const int k = lane + kW * s;
                    const double qv = v[s] / lam;
                    Qt[(size_t)p * kQST + k] = qv;  // measurements >= nm: v = 0
                    Rc[(size_t)p * kQST + k] = k < p ? l[s] : (k == p ? lam : 0.0);
                    res[s] = fma(-qv, qn, res[s]);
                    if (k == p) {
                        qtb[s] = qn;
                        diag[s] = lam;
                        pidx[s] = jmax;
                        xpos[s] = 0;
                    }
                }
                sync();
#pragma unroll
                for (int s = 0; s < KB; ++s)
                    if (binof(lane, s) == jmax) inP[s] = true;
                p += 1;
            }

            // ---- inner loop
            for (;;) {
                if (status != 1) break;
                // z = R^{-1} Q^T b: back substitution over the columns of R
                {
                    double t[kQS] = {qtb[0], qtb[1]};
                    for (int k = p - 1; k >= 0; --k) {
                        const double zk = rlk(t, k) / rlk(diag, k);
#pragma unroll
                        for (int s = 0; s < kQS; ++s) {
                            const int i = lane + kW * s;
                            if (i == k) z[s] = zk;
                            if (i < k) t[s] = fma(-Rc[(size_t)k * kQST + i], zk, t[s]);
                        }
                    }
                }
                iteration += 1;
                if (iteration == A.max_iter) {
                    status = 0;
                    break;
                }
                bool viol[kQS];
                bool anyv = false;
#pragma unroll
                for (int s = 0; s < kQS; ++s) {
                    viol[s] = lane + kW * s < p && z[s] <= 0;
                    anyv = anyv || viol[s];
                }
                if (!__any(anyv ? 1 : 0)) {
#pragma unroll
                    for (int s = 0; s < kQS; ++s)
                        if (lane + kW * s < p) xpos[s] = z[s];
                    break;
                }
                double T[kQS];
                double tmin = INFINITY;
#pragma unroll
                for (int s = 0; s < kQS; ++s) {
                    T[s] = viol[s] ? -xpos[s] / (z[s] - xpos[s]) : INFINITY;
                    tmin = fmin(tmin, T[s]);
                }
                const double alpha = -wave_max(-tmin);
                int cand = kNone;
#pragma unroll
                for (int s = kQS - 1; s >= 0; --s)
                    if (viol[s] && T[s] == alpha) cand = lane + kW * s;
                int jj = uni_i(wave_min_i(cand));  // first position with the minimum
#pragma unroll
                for (int s = 0; s < kQS; ++s)
                    if (lane + kW * s < p) xpos[s] = xpos[s] + alpha * (z[s] - xpos[s]);
                for (;;) {
                    if (++guard > guard_max || jj >= p) {
                        status = 0;
                        break;
                    }
                    // ---- position jj leaves: delete column jj of R, restore the triangle with Givens rotations on the
                    // rows (i, i + 1), i = jj .. p - 2, and rotate the columns of Q and the entries of Q^T b with them
                    const int bin_out = rlk_i(pidx, jj);
                    sync();
                    for (int k = jj; k < p - 1; ++k) {
                        double cv[kQS];
#pragma unroll
                        for (int s = 0; s < kQS; ++s) cv[s] = Rc[(size_t)(k + 1) * kQST + lane + kW * s];
                        sync();
#pragma unroll
                        for (int s = 0; s < kQS; ++s) Rc[(size_t)k * kQST + lane + kW * s] = cv[s];
                        sync();
                    }
                    for (int i = jj; i < p - 1; ++i) {
                        sync();
                        const double f = Rc[(size_t)i * kQST + i], g = Rc[(size_t)i * kQST + i + 1];  // uniform reads
                        double c, sn, r;
                        givens(f, g, c, sn, r);
                        sync();
#pragma unroll
                        for (int s = 0; s < kQS; ++s) {
                            const int k = lane + kW * s;
                            if (k >= i && k < p - 1) {  // k = column of R
                                const double a = Rc[(size_t)k * kQST + i], b = Rc[(size_t)k * kQST + i + 1];
                                Rc[(size_t)k * kQST + i] = k == i ? r : c * a + sn * b;
                                Rc[(size_t)k * kQST + i + 1] = k == i ? 0.0 : c * b - sn * a;
                            }
                            {  // columns i, i + 1 of Q (k = measurement)
                                const double a = Qt[(size_t)i * kQST + k], b = Qt[(size_t)(i + 1) * kQST + k];
                                Qt[(size_t)i * kQST + k] = c * a + sn * b;
                                Qt[(size_t)(i + 1) * kQST + k] = c * b - sn * a;
                            }
                        }
                        const double t1 = rlk(qtb, i), t2 = rlk(qtb, i + 1);
#pragma unroll
                        for (int s = 0; s < kQS; ++s) {
                            const int k = lane + kW * s;
                            if (k == i) {
                                qtb[s] = c * t1 + sn * t2;
                                diag[s] = r;
                            }
                            if (k == i + 1) qtb[s] = c * t2 - sn * t1;
                        }
                    }
                    sync();
                    // the last (rotated) direction leaves the basis: its share of b goes back into the residual
                    {
                        const double ql = rlk(qtb, p - 1);
#pragma unroll
                        for (int s = 0; s < kQS; ++s) res[s] = fma(Qt[(size_t)(p - 1) * kQST + lane + kW * s], ql, res[s]);
                    }
                    {  // positions above jj move down by one
                        double xs[kQS];
                        int ps[kQS];
#pragma unroll
                        for (int s = 0; s < kQS; ++s) {
                            xs[s] = __shfl_down(xpos[s], 1);
                            ps[s] = __shfl_down(pidx[s], 1);
                            if (lane == kW - 1 && s + 1 < kQS) {
                                xs[s] = rl(xpos[s + 1 < kQS ? s + 1 : s], 0);
                                ps[s] = __builtin_amdgcn_readlane(pidx[s + 1 < kQS ? s + 1 : s], 0);
                            }
                        }
#pragma unroll
                        for (int s = 0; s < kQS; ++s) {
                            const int k = lane + kW * s;
                            if (k >= jj && k < p - 1) {
                                xpos[s] = xs[s];
                                pidx[s] = ps[s];
                            }
                        }
                    }
#pragma unroll
                    for (int s = 0; s < KB; ++s)
                        if (binof(lane, s) == bin_out) inP[s] = false;
                    p -= 1;
                    // round-off clean-up: any remaining x <= 0 leaves too (first position first)
                    int bc = kNone;
#pragma unroll
                    for (int s = kQS - 1; s >= 0; --s)
                        if (lane + kW * s < p && xpos[s] <= 0) bc = lane + kW * s;
                    const int bad = uni_i(wave_min_i(bc));
                    if (bad == kNone) break;
                    jj = bad;
                }
            }
        }
        // ---- outputs
        double xb[KB] = {};
        if (status == 1) {
            for (int i = 0; i < p; ++i) {
                const int b = rlk_i(pidx, i);
                const double xv = rlk(xpos, i);
#pragma unroll
                for (int s = 0; s < KB; ++s)
                    if (binof(lane, s) == b) xb[s] = xv;
            }
        }
        double *cv = A.coeff + (size_t)vox * n;
#pragma unroll
        for (int s = 0; s < KB; ++s) {
            const int j = binof(lane, s);
            if (j < n) cv[j] = xb[s];
        }
        double r2p = 0;
#pragma unroll
        for (int s = 0; s < kQS; ++s) r2p = fma(res[s], res[s], r2p);
        const double r2 = wave_sum(r2p);
        const double rn = sqrt(status == 1 ? r2 : yn2);  // failure: zeros, ||y|| (nnls_solver.py:205-210)
        if (lane == 0) {
            A.rnorm[vox] = rn;
            if (A.status) A.status[vox] = (int8_t)status;
            if (A.iters) A.iters[vox] = iteration;
        }
        sync();  // the next voxel's first append overwrites row 0 of the slab
    }
}

template <int KB> int launch_qr_big(NnlsPlanData *P, const QrArgs &a, hipStream_t stream) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, nnls_qr_big_kernel<KB>, kW, 0) != hipSuccess || occ < 1)
        return set_error(PNX_ERR_HIP, "nnls_qr_big_kernel does not fit on a CU");
    if (occ > 8) occ = 8;  // 264 KB of Q / R per wave: 8 waves per CU are 540 MB of slab, of which a fit touches the rows of its passive set
    long long grid = (long long)occ * P->cus;
    if (!P->qr_slab || P->qr_slab_groups < grid) {
        if (P->qr_slab) (void)hipFree(P->qr_slab);
        P->qr_slab = nullptr;
        if (hipMalloc(&P->qr_slab, (size_t)grid * kQrSlab * sizeof(double)) != hipSuccess)
            return set_error(PNX_ERR_NOMEM, "hipMalloc of the Q / R slab (%zu bytes) failed", (size_t)grid * kQrSlab * sizeof(double));
        // rows / columns >= p are never read before they are written, except the measurements >= n_meas of a row of Q, which the
        // append writes as zeros: no initialisation needed
        P->qr_slab_groups = (int)grid;
    }
    if (grid > a.n_vox) grid = a.n_vox;
    QrBigArgs ba;
    ba.a = a;
    ba.slab = P->qr_slab;
    hipLaunchKernelGGL(nnls_qr_big_kernel<KB>, dim3((unsigned)grid), dim3(kW), 0, stream, ba);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PNX_ERR_HIP, "nnls_qr_big launch: %s", hipGetErrorString(e));
    return PNX_OK;
}
}  // namespace

int nnls_qr_solve_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                         int8_t *status_d, int32_t *iters_d, hipStream_t stream) {
    if (n_vox <= 0) return PNX_OK;
    QrArgs a;
    a.y = y_d;
    a.coeff = coeff_d;
    a.rnorm = rnorm_d;
    a.status = status_d;
    a.iters = iters_d;
    a.Bp = P->Bp;
    a.n_vox = n_vox;
    a.n_meas = P->n_meas;
    a.n_bins = P->n_bins;
    a.max_iter = max_iter;
    {
        // 33 .. 64 measurements fit the LDS kernel (MP = 64: 67 KB, two waves per CU), but the slab kernel with its eight waves
        // per CU is the faster one there too: 4.16 against 2.82 M voxels/s at 33 b-values, 2.85 against 1.72 M at 64
        // (profiles/nnls_cliff_probe.py); PNX_NNLS_QR_SLAB_FROM=65 brings the LDS kernel back for comparison
        static const int slab_from = dev_getenv("PNX_NNLS_QR_SLAB_FROM") ? atoi(dev_getenv("PNX_NNLS_QR_SLAB_FROM")) : 33;
        const bool wide = P->bstride == kNnlsWideBins;
        if (P->n_meas >= slab_from || P->n_meas > 64) return wide ? launch_qr_big<8>(P, a, stream) : launch_qr_big<4>(P, a, stream);
        if (wide) return P->n_meas <= 32 ? launch_qr<32, 8>(P, a, stream) : launch_qr<64, 8>(P, a, stream);
    }
    return P->n_meas <= 32 ? launch_qr<32, 4>(P, a, stream) : launch_qr<64, 4>(P, a, stream);
}

}  // namespace pnx
