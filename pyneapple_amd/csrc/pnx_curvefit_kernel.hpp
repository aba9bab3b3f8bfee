// pnx_curvefit_kernel.hpp -- batched bounded NLLS (Trust-Region-Reflective) for gfx950, fp64.
//
// Hot path replaced: CurveFitSolver._fit_single_pixel (reference src/pyneapple/solvers/curvefit.py:246-317)
// -> scipy curve_fit(method="trf") -> least_squares -> trf_bounds (scipy/optimize/_lsq/trf.py:205-394),
// for every voxel of a volume at once.
//
// Mapping to the machine (see DESIGN.md for the measurements behind these choices):
//   * ONE LANE OWNS ONE VOXEL.  All of TRF's per-voxel algebra (n x n SVD, More' root find, reflective
//     step selection) is scalar work per problem; giving a voxel a whole wavefront would run that part
//     at 1/64 efficiency.  The wavefront is the unit that shares the b-value table (LDS), the signal
//     tile (LDS, [row][lane] so every ds_read_b64 is conflict-free) and the work queue.
//   * PERSISTENT LANES + WORK QUEUE.  Voxels need 5..60 function evaluations; a lane that finishes pulls the
//     next voxel from a global atomic counter instead of idling until the slowest voxel of its wave is done.
//     Every lane leaves the loop as soon as the queue is empty, so the grid always drains.
//   * ONE FUSED "ROW PASS" per trial point: residual, cost, Jacobian row (2-point FD exactly as
//     scipy/_numdiff.py:584-625, or analytic), J^T f and a Householder QR of J accumulated row_blk<N>() rows at a time.
//     The Jacobian never exists in memory; TRF's SVD of the (m+n) x n augmented matrix is taken from the
//     n x n triangular factor (QR then one-sided Jacobi), which is as accurate as LAPACK's SVD of J_aug.
//   * The Jacobian at the trial point is computed speculatively together with f(x_new): an accepted step
//     (the common case) then needs no second pass.
//   * ASYNCHRONOUS REFILL: a lane that has written its results claims the next voxel and issues global->LDS DMA loads
//     (global_load_lds) of its signal row; they land while the other lanes run the heavy phases and are collected by
//     one s_waitcnt vmcnt(0) in front of the next row pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnx {

constexpr int kWave = 64;
constexpr int kMaxB = 128;
constexpr int kMaxP = 8;
// rows merged into the QR factor per Householder block step: the block is RB x (N + 1) doubles of registers.  Measured
// (1 Mi voxels, profiles/model_probe.py): N = 3: 321 (RB 8) vs 288 (RB 4) M voxels/s; N = 5: 72.6 vs 75.8; N = 6: 46-49 vs 58
// (RB 8 spills ~500 B per lane there, RB 4 none).  PNX_ROW_BLK overrides for experiments.
#ifdef PNX_ROW_BLK
template <int N> constexpr int row_blk() { return PNX_ROW_BLK; }
#else
template <int N> constexpr int row_blk() { return N >= 7 ? 2 : (N >= 5 ? 4 : 8); }  // N = 7: 450 -> 200 B of spills with 2
#endif
constexpr double kEps = 2.220446049250313e-16;
constexpr double kSqrtEps = 1.4901161193847656e-08;
typedef __attribute__((address_space(3))) void lds_void;

// Control block of a streamed launch.  `ready` is written by 8-byte H2D copies enqueued behind each granule's upload (stream
// order = data before watermark); `done[g]` counts the waves that will never touch granule g again.
struct StreamCtl {
    unsigned long long ready;  // voxels [0, ready) are resident
    unsigned int abort;        // (unused since round 4: the abort word is host_flags[n_granules], see CurvefitArgs)
    unsigned int timed_out;    // kernel -> host: a lane gave up waiting for the watermark
    unsigned int done[1];      // [n_granules]
};

struct CurvefitArgs {
    const double *y;      // (n_vox, n_b)
    const double *p0;     // (N, n_vox) when per_voxel
    const double *lo;
    const double *hi;
    const double *fixed;  // (n_fixed, n_vox) when fixed_per_voxel
    double *popt;         // (N, n_vox)
    double *pcov;         // (n_vox, N, N) or null
    int8_t *status;
    int32_t *nfev;
    double *cost;
    unsigned long long *queue;  // work-queue head, zeroed before launch
    const int32_t *order;       // null, or (n_vox) a permutation: the queue's k-th pull fits voxel order[k] (longest fits first, see
                                // pnx_curvefit_opts::queue_order); never with a streamed launch (granules complete in index order)
    // Streamed launch (host-pointer calls, pnx_api.hip `curvefit_streamed`): ONE persistent kernel runs while the volume is
    // still being uploaded and its results are already being downloaded.  ctl != null selects the STREAM instantiation.
    StreamCtl *ctl;             // device memory: upload watermark, abort word, per-granule wave counts
    unsigned int *host_flags;   // pinned host memory: host_flags[g] = 1 once every wave has left granule g behind; host_flags[n_granules]
                                // is the ABORT word (host -> kernel: stop waiting for the watermark and leave).  It lives in host
                                // memory so that the host can raise it with a plain store -- a copy into device memory would queue up
                                // behind whatever stalls the upload stream, which is exactly when it is needed
    int granule_shift;          // granule = 1 << granule_shift voxels
    unsigned int stream_spins;  // watermark polls before a lane gives up (ctl->timed_out = 1)
    int phase;                  // 0: fit + covariance epilogue, 1: fit only, 2: covariance epilogue only
    long long n_vox;
    int n_b;
    int per_voxel;
    int fixed_per_voxel;
    int max_nfev;
    int n_fixed;
    int t1_mode;          // 0 none, 1 T1: S*(1-exp(-TR/T1)), 2 STEAM: additionally *exp(-TM/T1) (multiexp.py:210-241)
    double tr, tm;
    double ftol, xtol, gtol;
    double p0s[kMaxP], los[kMaxP], his[kMaxP], fixeds[kMaxP];
    int free_idx[kMaxP];
    int fixed_idx[kMaxP];
    int use_sigma;       // curve_fit(sigma = 1-D): residual and Jacobian rows are scaled by w[i] = 1 / sigma_i (scipy:_minpack_py.py:958-960, 545-562)
    int absolute_sigma;  // curve_fit(absolute_sigma = True): the covariance is not scaled by 2 cost / (n_b - n_free) (:1057-1063)
    double b[kMaxB];
    double w[kMaxB];     // 1 / sigma per measurement (use_sigma only)
};

// ---------------------------------------------------------------------------------------------
// Forward models.  Parameter order and operation order follow model_functions/multiexp.py:35-202;
// analytic Jacobian columns follow models/{monoexp.py:143-148, biexp.py:172-204, triexp.py:190-230}.
// E[c] = exp(-b * D_c).
// ---------------------------------------------------------------------------------------------
template <int MODEL> struct Model;

template <> struct Model<0> {  // mono [S0, D]
    static constexpr int NALL = 2, NC = 1;
    __device__ static constexpr int dpos(int c) { return 1; }
    __device__ static double signal(const double *p, const double *E) { return p[0] * E[0]; }
    __device__ static void jac(const double *p, const double *E, double x, double *r) {
        r[0] = E[0];
        r[1] = -x * p[0] * E[0];
    }
};
template <> struct Model<1> {  // bi reduced [f1, D1, D2]
    static constexpr int NALL = 3, NC = 2;
    __device__ static constexpr int dpos(int c) { return c == 0 ? 1 : 2; }
    __device__ static double signal(const double *p, const double *E) { return p[0] * E[0] + (1 - p[0]) * E[1]; }
    __device__ static void jac(const double *p, const double *E, double x, double *r) {
        r[0] = E[0] - E[1];
        r[1] = -x * p[0] * E[0];
        r[2] = -x * (1 - p[0]) * E[1];
    }
};
template <> struct Model<2> {  // bi S0 [f1, D1, D2, S0]
    static constexpr int NALL = 4, NC = 2;
    __device__ static constexpr int dpos(int c) { return c == 0 ? 1 : 2; }
    __device__ static double signal(const double *p, const double *E) {
        return p[3] * (p[0] * E[0] + (1 - p[0]) * E[1]);
    }
    __device__ static void jac(const double *p, const double *E, double x, double *r) {
        r[0] = p[3] * (E[0] - E[1]);
        r[1] = -x * p[3] * p[0] * E[0];
        r[2] = -x * p[3] * (1 - p[0]) * E[1];
        r[3] = p[0] * E[0] + (1 - p[0]) * E[1];
    }
};
template <> struct Model<3> {  // bi full [f1, D1, f2, D2]
    static constexpr int NALL = 4, NC = 2;
    __device__ static constexpr int dpos(int c) { return c == 0 ? 1 : 3; }
    __device__ static double signal(const double *p, const double *E) { return p[0] * E[0] + p[2] * E[1]; }
    __device__ static void jac(const double *p, const double *E, double x, double *r) {
        r[0] = E[0];
        r[1] = -x * p[0] * E[0];
        r[2] = E[1];
        r[3] = -x * p[2] * E[1];
    }
};
template <> struct Model<4> {  // tri reduced [f1, D1, f2, D2, D3]
    static constexpr int NALL = 5, NC = 3;
    __device__ static constexpr int dpos(int c) { return c == 0 ? 1 : (c == 1 ? 3 : 4); }
    __device__ static double signal(const double *p, const double *E) {
        return p[0] * E[0] + p[2] * E[1] + (1 - p[0] - p[2]) * E[2];
    }
    __device__ static void jac(const double *p, const double *E, double x, double *r) {
        const double f3 = 1 - p[0] - p[2];
        r[0] = E[0] - E[2];
        r[1] = -x * p[0] * E[0];
        r[2] = E[1] - E[2];
        r[3] = -x * p[2] * E[1];
        r[4] = -x * f3 * E[2];
    }
};
template <> struct Model<5> {  // tri S0 [f1, D1, f2, D2, D3, S0]
    static constexpr int NALL = 6, NC = 3;
    __device__ static constexpr int dpos(int c) { return c == 0 ? 1 : (c == 1 ? 3 : 4); }
    __device__ static double signal(const double *p, const double *E) {
        return p[5] * (p[0] * E[0] + p[2] * E[1] + (1 - p[0] - p[2]) * E[2]);
    }
    __device__ static void jac(const double *p, const double *E, double x, double *r) {
        const double f3 = 1 - p[0] - p[2], S0 = p[5];
        r[0] = S0 * (E[0] - E[2]);
        r[1] = -x * S0 * p[0] * E[0];
        r[2] = S0 * (E[1] - E[2]);
        r[3] = -x * S0 * p[2] * E[1];
        r[4] = -x * S0 * f3 * E[2];
        r[5] = p[0] * E[0] + p[2] * E[1] + f3 * E[2];
    }
};
template <> struct Model<6> {  // tri full [f1, D1, f2, D2, f3, D3]
    static constexpr int NALL = 6, NC = 3;
    __device__ static constexpr int dpos(int c) { return c == 0 ? 1 : (c == 1 ? 3 : 5); }
    __device__ static double signal(const double *p, const double *E) {
        return p[0] * E[0] + p[2] * E[1] + p[4] * E[2];
    }
    __device__ static void jac(const double *p, const double *E, double x, double *r) {
        r[0] = E[0];
        r[1] = -x * p[0] * E[0];
        r[2] = E[1];
        r[3] = -x * p[2] * E[1];
        r[4] = E[2];
        r[5] = -x * p[4] * E[2];
    }
};

// Column order used inside the QR / SVD (speed only, any order gives the same step): columns sorted by typical
// norm, largest first.  One-sided Jacobi then needs 2.6-3.6 sweeps instead of 3-6 (measured offline on synthetic
// Jacobians of every model).  Index k of the factor <-> parameter colperm<MODEL>(k).
template <int MODEL> __host__ __device__ constexpr int colperm(int k) {
    constexpr int P[7][6] = {{0, 1, 2, 3, 4, 5}, {2, 0, 1, 3, 4, 5}, {2, 0, 1, 3, 4, 5}, {3, 1, 2, 0, 4, 5},
                             {4, 3, 0, 2, 1, 5}, {4, 3, 0, 2, 1, 5}, {5, 3, 1, 4, 2, 0}};
    return P[MODEL][k];
}

template <int MODEL> __device__ constexpr int comp_of_param(int j) {
    for (int c = 0; c < Model<MODEL>::NC; ++c)
        if (Model<MODEL>::dpos(c) == j) return c;
    return -1;
}

// ---------------------------------------------------------------------------------------------
// Small dense helpers, everything statically indexed (register resident).
// ---------------------------------------------------------------------------------------------
template <int N> __device__ inline double dotn(const double *a, const double *b) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) s += a[i] * b[i];
    return s;
}
template <int N> __device__ inline double normn(const double *a) { return sqrt(dotn<N>(a, a)); }

// Reciprocal / reciprocal square root to ~1 ulp from the hardware seeds (v_rcp_f64 / v_rsq_f64) and two
// Newton steps: 5-6 VALU ops instead of the 11 (fdiv) / 18 (sqrt) of the IEEE-exact expansions.  Used only
// inside orthogonal transformations (Householder, Givens/Jacobi), where a last-bit error perturbs
// orthogonality at the 1e-16 level and nothing else.  Arguments are finite and > 0.
__device__ inline double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// returns sqrt(x), *rs = 1/sqrt(x)
__device__ inline double fast_sqrt_rsqrt(double x, double *rs) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    *rs = h + h;
    return g;
}

// exp(x), fp64, <= 1 ulp class: the reduction and the degree-11 polynomial of the device library's exp (k = rint(x / ln 2),
// r = x - k ln2_hi - k ln2_lo, |r| <= 0.347) without its overflow / underflow selects -- v_ldexp_f64 saturates to inf / 0
// by itself, v_cvt_i32_f64 saturates k, NaN propagates through the polynomial -- and with the polynomial split into
// even / odd halves (two dependent chains of 6 instead of one of 11: the row pass evaluates 3 of these per row at one
// wave per SIMD, where the FMA latency of a single chain is exposed).
#ifndef PNX_CF_FAST_EXP
#define PNX_CF_FAST_EXP 1
#endif
// branch-weight hint for the rare fallbacks inside the row loop (expm1(z)/z with its division): the compiler lays them out
// behind the loop, which halves the instruction footprint of the row pass (1 991 -> 993 lines of assembly)
#ifndef PNX_CF_HINTS
#define PNX_CF_HINTS 1
#endif
#if PNX_CF_HINTS
#define PNX_LIKELY(x) __builtin_expect(!!(x), 1)
#else
#define PNX_LIKELY(x) (x)
#endif
__device__ inline double exp_fast(double x) {
#if PNX_CF_FAST_EXP
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = fma(k, -0.6931471805599453, x);
    r = fma(k, -2.3190468138462996e-17, r);
    const double r2 = r * r;
    // exp(r) = 1 + r (1 + r q(r)), q = c2 + c3 r + ... + c11 r^9 with the library's minimax coefficients
    double qe = 2.7630903490112654e-07;         // c10
    qe = fma(qe, r2, 2.480149103909504e-05);    // c8
    qe = fma(qe, r2, 0.0013888888945916382);    // c6
    qe = fma(qe, r2, 0.041666666666519754);     // c4
    qe = fma(qe, r2, 0.5000000000000012);       // c2
    double qo = 2.502232256764614e-08;          // c11
    qo = fma(qo, r2, 2.755751454582531e-06);    // c9
    qo = fma(qo, r2, 0.00019841269589115522);   // c7
    qo = fma(qo, r2, 0.008333333333455043);     // c5
    qo = fma(qo, r2, 0.16666666666666477);      // c3
    const double q = fma(r, qo, qe);
    return ldexp(fma(r, fma(r, q, 1.0), 1.0), __double2int_rz(k));  // v_cvt_i32_f64 saturates, v_ldexp_f64 over / underflows by itself
#else
    return exp(x);
#endif
}

// Merge RB rows (J part in blk[r][0..N), rhs in blk[r][N]) into the upper-triangular factor R | q
// by Householder reflections acting on [R[k][k]; blk[:,k]].  Afterwards blk is garbage.
template <int N, int RB> __device__ inline void qr_merge(double (&R)[N][N], double (&q)[N], double (&blk)[RB][N + 1]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double sig = 0;
#pragma unroll
        for (int r = 0; r < RB; ++r) sig += blk[r][k] * blk[r][k];
        if (sig > 0) {
            const double alpha = R[k][k];
            double rs;
            const double nrm = fast_sqrt_rsqrt(alpha * alpha + sig, &rs);
            const double beta = alpha <= 0 ? nrm : -nrm;
            const double v0 = alpha - beta;            // |v0| >= nrm > 0
            const double inv_v0 = copysign(fast_rcp(fabs(v0)), v0);
            const double tau = -v0 * (alpha <= 0 ? rs : -rs);  // -v0 / beta
#pragma unroll
            for (int r = 0; r < RB; ++r) blk[r][k] *= inv_v0;
            R[k][k] = beta;
#pragma unroll
            for (int j = k + 1; j <= N; ++j) {
                double w = (j < N) ? R[k][j] : q[k];
#pragma unroll
                for (int r = 0; r < RB; ++r) w += blk[r][k] * blk[r][j];
                w *= tau;
                if (j < N)
                    R[k][j] -= w;
                else
                    q[k] -= w;
#pragma unroll
                for (int r = 0; r < RB; ++r) blk[r][j] -= w * blk[r][k];
            }
        }
    }
}

// One-sided Jacobi SVD of the N x N matrix W (in place: W <- U*diag(s)), V accumulates the right
// singular vectors (columns).  Callers pass the TRANSPOSE of a triangular factor (lower triangular W):
// row-cyclic Jacobi converges in ~1 sweep less on R^T than on R (Drmac-Veselic), see DESIGN.md.
// A sweep whose largest |cos(angle)| was below 1e-8 is the last one: Jacobi converges quadratically, so the
// rotations of that sweep already leave the off-diagonal at the 1e-16 level.
template <int N> __device__ inline void jacobi_svd(double (&W)[N][N], double (&V)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    if (N == 1) return;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;  // some pair of this sweep had g^2 / (a b) >= 1e-16 (tested as a product: no division)
#pragma unroll
        for (int p = 0; p < N - 1; ++p) {
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                double a = 0, b = 0, g = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    a += W[i][p] * W[i][p];
                    b += W[i][q] * W[i][q];
                    g += W[i][p] * W[i][q];
                }
                const double g2 = g * g, ab = a * b;
                if (g2 > 1.0e-30 * ab) {  // |cos| > 1e-15
                    rotated = rotated || (g2 >= 1.0e-16 * ab);
                    // t = tan(theta) = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)),  zeta = (b - a) / (2 g)
                    const double num = b - a, den = 2.0 * g;
                    double rs;
                    const double hyp = fast_sqrt_rsqrt(num * num + den * den, &rs);
                    double t = fabs(den) * fast_rcp(fabs(num) + hyp);
                    t = ((num < 0) != (den < 0)) ? -t : t;
                    double c;
                    (void)fast_sqrt_rsqrt(1.0 + t * t, &c);
                    const double s = c * t;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const double wp = W[i][p], wq = W[i][q];
                        W[i][p] = c * wp - s * wq;
                        W[i][q] = s * wp + c * wq;
                        const double vp = V[i][p], vq = V[i][q];
                        V[i][p] = c * vp - s * vq;
                        V[i][q] = s * vp + c * vq;
                    }
                }
            }
        }
        if (!rotated) break;
    }
}

// scipy/optimize/_lsq/common.py:400-464 with rstep = 0 (np.nextafter form)
__device__ inline double strictly_feasible0(double x, double lb, double ub) {
    if (x <= lb) x = nextafter(lb, ub);
    if (x >= ub) x = nextafter(ub, lb);  // numpy applies the lower mask first, then the upper one
    if (x < lb || x > ub) x = 0.5 * (lb + ub);
    return x;
}
// same with rstep = 1e-10 (least_squares.py:827-828)
__device__ inline double strictly_feasible_r(double x, double lb, double ub) {
    const double rstep = 1e-10;
    const double lower_dist = x - lb, upper_dist = ub - x;
    const double lt = rstep * fmax(1.0, fabs(lb)), ut = rstep * fmax(1.0, fabs(ub));
    int active = 0;
    if (isfinite(lb) && lower_dist <= fmin(upper_dist, lt)) active = -1;
    if (isfinite(ub) && upper_dist <= fmin(lower_dist, ut)) active = 1;
    if (active == -1) x = lb + lt;
    if (active == 1) x = ub - ut;
    if (x < lb || x > ub) x = 0.5 * (lb + ub);
    return x;
}

// scipy/optimize/_lsq/common.py:367-397
template <int N>
__device__ inline double step_size_to_bound(const double *x, const double *s, const double *lb, const double *ub,
                                            int *hits) {
    double steps[N];
    double mn = INFINITY;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // max((lb - x) / s, (ub - x) / s) is the quotient with the larger numerator for s > 0 and with the smaller one for
        // s < 0 (lb <= ub): one IEEE division instead of two, the same value bit for bit
        const double num = (s[i] > 0) ? (ub[i] - x[i]) : (lb[i] - x[i]);
        steps[i] = (s[i] != 0) ? num / s[i] : INFINITY;
        mn = fmin(mn, steps[i]);
    }
    if (hits) {
#pragma unroll
        for (int i = 0; i < N; ++i) hits[i] = (steps[i] == mn) ? ((s[i] > 0) - (s[i] < 0)) : 0;
    }
    return mn;
}

// 0.5*||R2 s||^2 + g_h.s  ==  evaluate_quadratic(J_h, g_h, s, diag=diag_h) (common.py:325-362), because
// R2^T R2 = J_h^T J_h + diag(diag_h).
template <int N> __device__ inline void rmul(const double (&R)[N][N], const double *s, double *out) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double a = 0;
#pragma unroll
        for (int j = i; j < N; ++j) a += R[i][j] * s[j];
        out[i] = a;
    }
}

// common.py:303-322
__device__ inline double minimize_quadratic_1d(double a, double b, double lb, double ub, double c, double &y) {
    double tb = lb, yb = lb * (a * lb + b) + c;
    const double yu = ub * (a * ub + b) + c;
    if (yu < yb) {
        yb = yu;
        tb = ub;
    }
    if (a != 0) {
        const double ext = -0.5 * b / a;
        if (lb < ext && ext < ub) {
            const double ye = ext * (a * ext + b) + c;
            if (ye < yb) {
                yb = ye;
                tb = ext;
            }
        }
    }
    y = yb;
    return tb;
}

// common.py:57-168
template <int N>
__device__ inline double solve_lsq_trust_region(int m, const double *uf, const double *s, const double (&V)[N][N],
                                                double smax, double smin, double Delta, double initial_alpha,
                                                double *p) {
    double suf[N], t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) suf[i] = s[i] * uf[i];
    const bool full_rank = (m >= N) && (smin > kEps * m * smax);
    if (full_rank) {
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = uf[i] / s[i];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double a = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) a += V[i][k] * t[k];
            p[i] = -a;
        }
        if (normn<N>(p) <= Delta) return 0.0;
    }
    double alpha_upper = normn<N>(suf) / Delta;
    double alpha_lower = 0.0;
    auto phi_and_derivative = [&](double alpha, double &phi, double &phi_prime) {
        double pn2 = 0, sp = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double denom = s[i] * s[i] + alpha;
            const double q = suf[i] / denom;
            pn2 += q * q;
            sp += suf[i] * suf[i] / (denom * denom * denom);
        }
        const double p_norm = sqrt(pn2);
        phi = p_norm - Delta;
        phi_prime = -sp / p_norm;
    };
    double phi, phi_prime;
    if (full_rank) {
        phi_and_derivative(0.0, phi, phi_prime);
        alpha_lower = -phi / phi_prime;
    }
    double alpha;
    if (!full_rank && initial_alpha == 0)
        alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    else
        alpha = initial_alpha;
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper)
            alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        phi_and_derivative(alpha, phi, phi_prime);
        if (phi < 0) alpha_upper = alpha;
        const double ratio = phi / phi_prime;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = suf[i] / (s[i] * s[i] + alpha);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double a = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) a += V[i][k] * t[k];
        p[i] = -a;
    }
    const double sc = Delta / normn<N>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] *= sc;
    return alpha;
}

// trf.py:128-202.  Returns predicted reduction; writes step / step_h.
template <int N>
__device__ inline double select_step(const double *x, const double (&R2)[N][N], const double *g_h, double *p,
                                     double *p_h, const double *d, double Delta, const double *lb, const double *ub,
                                     double theta, double *step, double *step_h) {
    bool inb = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double xp = x[i] + p[i];
        inb = inb && (xp >= lb[i]) && (xp <= ub[i]);
    }
    double t1[N], t2[N];
    if (inb) {
        rmul<N>(R2, p_h, t1);
        const double p_value = 0.5 * dotn<N>(t1, t1) + dotn<N>(p_h, g_h);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            step[i] = p[i];
            step_h[i] = p_h[i];
        }
        return -p_value;
    }
    int hits[N];
    const double p_stride = step_size_to_bound<N>(x, p, lb, ub, hits);
    double r_h[N], r[N], xb[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r_h[i] = hits[i] ? -p_h[i] : p_h[i];
        r[i] = d[i] * r_h[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        p[i] *= p_stride;
        p_h[i] *= p_stride;
        xb[i] = x[i] + p[i];
    }
    // intersect_trust_region(p_h, r_h, Delta) (common.py:18-54), positive root
    double to_tr;
    {
        const double a = dotn<N>(r_h, r_h), b = dotn<N>(p_h, r_h), c = dotn<N>(p_h, p_h) - Delta * Delta;
        const double dd = sqrt(b * b - a * c);
        const double q = -(b + copysign(dd, b));
        const double ta = q / a, tb = c / q;
        to_tr = fmax(ta, tb);
    }
    double to_bound = step_size_to_bound<N>(xb, r, lb, ub, nullptr);
    double r_stride = fmin(to_bound, to_tr);
    double r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = (1 - theta) * p_stride / r_stride;
        r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        // build_quadratic_1d(J_h, g_h, r_h, s0=p_h, diag=diag_h) (common.py:250-300)
        rmul<N>(R2, r_h, t1);   // v
        rmul<N>(R2, p_h, t2);   // u
        const double a = 0.5 * dotn<N>(t1, t1);
        const double b = dotn<N>(g_h, r_h) + dotn<N>(t2, t1);
        const double c = 0.5 * dotn<N>(t2, t2) + dotn<N>(g_h, p_h);
        r_stride = minimize_quadratic_1d(a, b, r_stride_l, r_stride_u, c, r_value);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            r_h[i] = r_h[i] * r_stride + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else
        r_value = INFINITY;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        p[i] *= theta;
        p_h[i] *= theta;
    }
    rmul<N>(R2, p_h, t1);
    const double p_value = 0.5 * dotn<N>(t1, t1) + dotn<N>(p_h, g_h);

    double ag_h[N], ag[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ag_h[i] = -g_h[i];
        ag[i] = d[i] * ag_h[i];
    }
    to_tr = Delta / normn<N>(ag_h);
    to_bound = step_size_to_bound<N>(x, ag, lb, ub, nullptr);
    double ag_stride = (to_bound < to_tr) ? theta * to_bound : to_tr;
    double ag_value;
    {
        rmul<N>(R2, ag_h, t1);
        const double a = 0.5 * dotn<N>(t1, t1);
        const double b = dotn<N>(g_h, ag_h);
        ag_stride = minimize_quadratic_1d(a, b, 0.0, ag_stride, 0.0, ag_value);
    }
    if (p_value < r_value && p_value < ag_value) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            step[i] = p[i];
            step_h[i] = p_h[i];
        }
        return -p_value;
    } else if (r_value < p_value && r_value < ag_value) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            step[i] = r[i];
            step_h[i] = r_h[i];
        }
        return -r_value;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            step[i] = ag[i] * ag_stride;
            step_h[i] = ag_h[i] * ag_stride;
        }
        return -ag_value;
    }
}

// ---------------------------------------------------------------------------------------------
// The kernel.  Block = WAVES wavefronts; LDS: b-values + per-wave signal tile y[row][lane].
// ---------------------------------------------------------------------------------------------
enum LaneState : int { ST_IDLE = 0, ST_INIT = 1, ST_RUN = 2, ST_FINAL = 3 };

template <int N> struct Park {
    static constexpr int NR = N * (N + 1) / 2;          // packed R factor of J at the current iterate (for pcov)
    static constexpr bool kParkV = (N <= 5);            // right singular vectors live in LDS between B and C
    static constexpr int NV = kParkV ? N * N : 0;
    // doubles of LDS per wave besides the b-value table
    // the signal tile is [ceil(n_b / 2)][64 lanes][2] (the layout a 16-byte LDS-DMA load per lane produces)
    __host__ __device__ static constexpr int per_wave(int n_b) { return kWave * (((n_b + 1) & ~1) + NR + NV); }
};

// Loop order -- one iteration = one row pass per lane:
//   refill   idle lanes pull the next voxel (x_new = strictly feasible p0) or leave for good;
//   pass     residual / cost / Jacobian / J^T f / QR(J) at x_new  (every lane);
//   D        accept or reject, trust radius, ftol / xtol tests;
//   B-light  Coleman-Li scaling + gtol test for lanes whose iterate changed or that are about to stop;
//   final    lanes that stop write their outputs and become idle (refilled at the top of the next iteration,
//            so a finished voxel never costs a wasted pass);
//   B-heavy  lanes that accepted: QR of the augmented factor + Jacobi SVD (consumes the pass's R directly);
//   C        trust-region step -> next x_new.
// Register diet: bounds stay in SGPRs unless they are per voxel (PV); R (for the covariance) and the
// singular vectors V are parked in LDS; nothing produced by the pass stays live across the next pass.
#ifndef PNX_CF_WAVES_PER_SIMD
#define PNX_CF_WAVES_PER_SIMD 1  // 2 was measured at 47-60 M voxels/s (spills, and the LDS park of 8 waves does not fit for n_b = 32)
#endif
#ifndef PNX_CF_BLOCK_WAVES
#define PNX_CF_BLOCK_WAVES 4
#endif
template <int MODEL, int N, bool FD, bool PV, bool T1, bool STREAM = false>
__global__ void __launch_bounds__(64 * PNX_CF_BLOCK_WAVES, PNX_CF_WAVES_PER_SIMD) curvefit_kernel(const CurvefitArgs A) {
    using M = Model<MODEL>;
    using PK = Park<N>;
    constexpr int NALL = M::NALL;            // parameters of the diffusion model
    constexpr int NP = NALL + (T1 ? 1 : 0);  // + T1 as the last parameter (models/*.py: `names.append("T1")`)
    constexpr int NC = M::NC;
    constexpr bool HASFIXED = (N != NP);
    static_assert(!(FD && HASFIXED), "finite-difference mode is only built without fixed parameters");
    // factor column order: static per model when all parameters are free, natural order with fixed parameters
    auto CP = [](int k) constexpr { return (HASFIXED || k >= NALL) ? k : colperm<MODEL>(k); };

    extern __shared__ double smem[];
    const int n_b = A.n_b;
    double *bsh = smem;                                   // [kMaxB]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const bool use_w = A.use_sigma != 0;                  // wave uniform: a scalar branch per row, nothing else changes without sigma
    const double *wsh = smem + kMaxB;                     // [kMaxB] 1 / sigma per measurement -- only allocated with sigma
    double *ytile = smem + (use_w ? 2 * kMaxB : kMaxB) + (size_t)wave * PK::per_wave(n_b);  // [ceil(n_b/2)][64][2], wave-uniform base
    double *ysh = ytile + 2 * lane;                                         // this lane's pair column
    double *rpark = ytile + (size_t)((n_b + 1) & ~1) * kWave + lane;        // [NR][64]
    double *vpark = rpark + (size_t)PK::NR * kWave;                           // [NV][64]
    for (int i = threadIdx.x; i < n_b; i += blockDim.x) bsh[i] = A.b[i];
    if (use_w)
        for (int i = threadIdx.x; i < n_b; i += blockDim.x) smem[kMaxB + i] = A.w[i];
    __syncthreads();
    // the refill uses asynchronous 16-byte global->LDS loads, one per pair of b-values.  The global side needs no more than the
    // 8-byte alignment every row of doubles has (rows of an odd number of b-values start on odd multiples of 8), and the last
    // value of such a row travels with the first value of the next row as its unused partner (round 4: 31 b-values had cost
    // 18 % against 32 on the synchronous row copy, 23 against 24 12 %: profiles/odd_nb_probe.py)
    const bool dma_ok = ((reinterpret_cast<uintptr_t>(A.y) & 7) == 0);

    // ---- per-lane persistent state
    int state = ST_IDLE;
    long long vox = -1;
    unsigned long long ready_seen = 0;  // STREAM: last upload watermark this lane saw
    int pub = 0;                        // STREAM: granules [0, pub) published as left behind by this wave (wave-uniform)
    // STREAM: the queue hands out ascending indices, so once no lane of the wave holds a voxel of granule <= g the wave
    // never writes to granule g again.  It then makes its stores visible (system-scope release) and counts itself out of
    // g; the wave that completes the count raises the host's flag, and the host starts that granule's download while the
    // kernel keeps running.
    auto publish = [&](bool all) {
        const int n_gran = (int)((A.n_vox + (1ll << A.granule_shift) - 1) >> A.granule_shift);
        int upto = pub;
        while (upto < n_gran && (all || !__ballot(state != ST_IDLE && (vox >> A.granule_shift) <= upto))) ++upto;
        if (upto == pub) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        const unsigned int n_waves = gridDim.x * (blockDim.x / kWave);
        if ((threadIdx.x & (kWave - 1)) == __ffsll((unsigned long long)__ballot(1)) - 1) {
            for (int g = pub; g < upto; ++g) {
                const unsigned int old = __hip_atomic_fetch_add(&A.ctl->done[g], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
                if (old + 1 == n_waves) __hip_atomic_store(&A.host_flags[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        pub = upto;
    };
    double x[N], lb[N], ub[N];
    double pfull[NP];
    double g[N];
    double s[N], uf[N], R2[N][N], d[N], g_h[N];
    double Vreg[PK::kParkV ? 1 : N][PK::kParkV ? 1 : N];
    double smax = 0, smin = 0, theta = 0;
    double cost = 0, Delta = 0, alpha = 0;
    int nfev = 0, term = -99;
    bool first = false;
    double xn[N], step_h_norm = 0, step_norm = 0, predicted = 0;
    if (!PV) {
#pragma unroll
        for (int k = 0; k < N; ++k) {  // wave-uniform, never written again: stays in SGPRs
            lb[k] = A.los[k];
            ub[k] = A.his[k];
        }
    }

#ifdef PNX_CF_STAMP
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, act[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl = __builtin_amdgcn_s_memtime();
#define CFSTAMP(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[k] += t_ - tl; tl = t_; } while (0)
#define CFACT(k, cond) do { act[k] += __popcll(__ballot(cond)); cnt[k] += 1; } while (0)
#else
#define CFSTAMP(k) do {} while (0)
#define CFACT(k, cond) do {} while (0)
#endif
    auto refill = [&]() {
    while (state == ST_IDLE) {
        const unsigned long long idx = atomicAdd(A.queue, 1ULL);
        if (idx >= (unsigned long long)A.n_vox) break;  // queue empty: this lane is done for good
        if constexpr (STREAM) {
            // the voxel's signal may still be on its way: wait for the upload watermark (acquire: the rows behind it are
            // visible).  Bounded: on abort or after stream_spins polls the lane leaves, so the grid always drains.
            if (idx >= ready_seen) {
                bool got = false;
                const unsigned int *abort_word = A.host_flags + ((A.n_vox + (1ll << A.granule_shift) - 1) >> A.granule_shift);
                for (unsigned int spins = 0; spins < A.stream_spins; ++spins) {
                    ready_seen = __hip_atomic_load(&A.ctl->ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (idx < ready_seen) {
                        got = true;
                        break;
                    }
                    // the abort word lives in pinned HOST memory: every read crosses PCIe, beside the upload the lanes are waiting
                    // for -- so it is looked at on every 16th poll only (an abort is seen within ~100 us instead of ~6)
                    if ((spins & 15u) == 15u && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
                    __builtin_amdgcn_s_sleep(127);
                }
                if (!got) {
                    if (!__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                        __hip_atomic_store(&A.ctl->timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;  // leaves with state IDLE
                }
            }
        }
        vox = (long long)idx;
        if constexpr (!STREAM) {
            if (A.order) vox = A.order[idx];
        }
        const double *yv = A.y + (size_t)vox * n_b;
        if (dma_ok) {
            // asynchronous refill: 16-byte global->LDS loads (no VGPR round trip, nothing waits here); they land
            // while the other lanes run phases B / C and are waited for (vmcnt) right before the next row pass
            const int pairs = n_b >> 1;
            for (int c = 0; c < pairs; ++c)
                __builtin_amdgcn_global_load_lds((const void *)(yv + 2 * c), (lds_void *)(ytile + c * 2 * kWave), 16, 0, 0);
            if (n_b & 1) {
                if (vox + 1 < A.n_vox)  // the pair (last value, first value of the next row): the second half is never read
                    __builtin_amdgcn_global_load_lds((const void *)(yv + n_b - 1), (lds_void *)(ytile + pairs * 2 * kWave), 16, 0, 0);
                else  // the last row of the array has no row behind it: nothing is read beyond the caller's buffer
                    ysh[pairs * 2 * kWave] = yv[n_b - 1];
            }
        } else {
            for (int i = 0; i < n_b; ++i) ysh[(i >> 1) * 2 * kWave + (i & 1)] = yv[i];
        }
        bool okb = true, okp = true;
        double p0v[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (PV) {
                p0v[k] = A.p0[(size_t)k * A.n_vox + vox];
                lb[k] = A.lo[(size_t)k * A.n_vox + vox];
                ub[k] = A.hi[(size_t)k * A.n_vox + vox];
            } else
                p0v[k] = A.p0s[k];
            okb = okb && (lb[k] < ub[k]);                          // least_squares.py:814-816
            okp = okp && (p0v[k] >= lb[k]) && (p0v[k] <= ub[k]);  // least_squares.py:818-819
        }
        if (HASFIXED) {
#pragma unroll
            for (int j = 0; j < NP; ++j) pfull[j] = 0;
            for (int f = 0; f < A.n_fixed; ++f) {
                const double fv = A.fixed_per_voxel ? A.fixed[(size_t)f * A.n_vox + vox] : A.fixeds[f];
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if (A.fixed_idx[f] == j) pfull[j] = fv;
            }
        }
        if (!okb || !okp) {
            // reference: ValueError inside curve_fit -> params = p0, cov = NaN, success = False
            // (a non-finite signal is detected in the voxel's first row pass)
            const int st = !okb ? -1 : -3;
#pragma unroll
            for (int k = 0; k < N; ++k) A.popt[(size_t)k * A.n_vox + vox] = p0v[k];
            if (A.status) A.status[vox] = (int8_t)st;  // pcov_kernel writes the NaN covariance
            if (A.nfev) A.nfev[vox] = 0;
            if (A.cost) A.cost[vox] = NAN;
            continue;  // stays IDLE -> next voxel
        }
#pragma unroll
        for (int k = 0; k < N; ++k) xn[k] = strictly_feasible_r(p0v[k], lb[k], ub[k]);
        state = ST_INIT;
        nfev = 0;
        term = -99;
        alpha = 0.0;
    }
    };
    refill();
    if constexpr (STREAM) {
        // `pub` is only uniform among the lanes that are still in the loop below (a lane that left keeps the value it left
        // with), so nothing is published behind the loop: the last lane of the wave to run dry publishes the rest from inside
        // it, and a wave that never got a voxel does so here.
        if (!__ballot(state != ST_IDLE)) publish(true);
    }
    for (;;) {
        CFSTAMP(7);
        if (state == ST_IDLE) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA signal tiles have landed
        CFSTAMP(0);
        CFACT(1, true);

        // ------------------------------------------------------------------ row pass at xn
        // residual f = model(xn) - y, cost, Jacobian (closed-form 2-point FD or analytic), g = J^T f, QR of J
        double Rn[N][N], qn[N], gn[N], cost_new = 0;
        bool finite_f = true, yfinite = true;
        {
            double pe[NP];  // full parameter vector at the evaluation point
            if (HASFIXED) {
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    pe[j] = pfull[j];
#pragma unroll
                    for (int k = 0; k < N; ++k)
                        if (A.free_idx[k] == j) pe[j] = xn[k];
                }
            } else {
#pragma unroll
                for (int j = 0; j < NP; ++j) pe[j] = xn[j];
            }
            double dxv[N];
            if (FD) {
                // _numdiff.py:146-163 (_compute_absolute_step) and :13-90 ('1-sided', num_steps = 1)
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double sign = (xn[k] >= 0) ? 1.0 : -1.0;
                    double h = kSqrtEps * sign * fmax(1.0, fabs(xn[k]));
                    const double lower_dist = xn[k] - lb[k], upper_dist = ub[k] - xn[k];
                    const double xx = xn[k] + h;
                    const bool violated = (xx < lb[k]) || (xx > ub[k]);
                    const bool fitting = fabs(h) <= fmax(lower_dist, upper_dist);
                    if (violated && fitting) h = -h;
                    if (!fitting) h = (upper_dist >= lower_dist) ? upper_dist : -lower_dist;
                    dxv[k] = (xn[k] + h) - xn[k];  // _numdiff.py:596 "recompute dx as exactly representable number"
                }
            }
            // T1 / STEAM relaxation factor (row independent): S = base * A1 [* eTM], A1 = 1 - exp(-TR/T1)
            // (model_functions/multiexp.py:210-241); its T1 derivative analytically (multiexp.py:244-302) or as
            // SciPy's 2-point quotient (fac(T1 + dx) - fac(T1)) / dx -- two extra exp per PASS, not per row.
            double A1 = 1.0, eTM = 1.0, fac = 1.0, dfac = 0.0;
            if (T1) {
                const bool steam = A.t1_mode == 2;
                const double T1v = pe[NALL];
                const double eTR = exp(-A.tr / T1v);
                A1 = 1 - eTR;
                eTM = steam ? exp(-A.tm / T1v) : 1.0;
                fac = A1 * eTM;
                dfac = steam ? eTM / (T1v * T1v) * (-A.tr * eTR + A.tm * A1) : (-eTR * A.tr / (T1v * T1v));
                if (FD) {  // T1 is the last free parameter in FD mode (no fixed parameters there)
                    const double dx = dxv[N - 1];
                    const double T2 = T1v + dx;
                    const double fac2 = (1 - exp(-A.tr / T2)) * (steam ? exp(-A.tm / T2) : 1.0);
                    dfac = (fac2 - fac) / dx;
                }
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                qn[i] = 0;
                gn[i] = 0;
#pragma unroll
                for (int j = 0; j < N; ++j) Rn[i][j] = 0;
            }
            constexpr int kRowBlk = row_blk<N>();
            for (int i0 = 0; i0 < n_b; i0 += kRowBlk) {
                double blk[kRowBlk][N + 1];
#pragma unroll
                for (int r = 0; r < kRowBlk; ++r) {
                    const int i = i0 + r;
                    const bool live = i < n_b;
                    const int ii = live ? i : 0;
                    const double bb = bsh[ii];
                    const double nb = -bb;
                    const double yi = ysh[(ii >> 1) * 2 * kWave + (ii & 1)];
                    yfinite = yfinite && isfinite(yi);
                    double E[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) E[c] = exp_fast(nb * pe[M::dpos(c)]);
                    const double base = M::signal(pe, E);
                    double r0 = (T1 ? base * A1 * eTM : base) - yi;
                    double jr[N];
                    {
                        double ja[NP];
                        M::jac(pe, E, bb, ja);
                        if (FD) {
                            // SciPy's 2-point quotient (f(x + dx e_k) - f(x)) / dx in closed form.  The models are
                            // linear in every parameter except the D_c, and for a D_c
                            //   f(D_c + dx) - f(D_c) = w_c exp(-b D_c) (exp(-b dx) - 1),
                            // so the quotient is the analytic column times g(z) = expm1(z)/z, z = -b dx: no second
                            // exp per column, no cancellation, and SciPy's deterministic O(dx) bias (up to 1e-5
                            // relative at b = 1200) -- which is what separates its iterates from an
                            // analytic-Jacobian run -- is kept exactly.
#pragma unroll
                            for (int k = 0; k < NALL; ++k) {
                                if (comp_of_param<MODEL>(k) >= 0) {
                                    const double z = nb * dxv[k];
                                    double gz;
#if PNX_CF_FAST_EXP
                                    if (PNX_LIKELY(fabs(z) < 2e-4))  // dx ~ 1.5e-8 max(1, |D|): the normal case; z^4 / 120 < 2e-17
                                        gz = fma(z, fma(z, fma(z, 1.0 / 24, 1.0 / 6), 0.5), 1.0);
                                    else
#endif
                                    if (fabs(z) < 1e-3)
                                        gz = 1.0 + z * (0.5 + z * (1.0 / 6 + z * (1.0 / 24 + z * (1.0 / 120 + z * (1.0 / 720)))));
                                    else
                                        gz = expm1(z) / z;
                                    ja[k] *= gz;
                                }
                            }
                        }
                        if (T1) {
#pragma unroll
                            for (int k = 0; k < NALL; ++k) ja[k] *= fac;
                            ja[NP - 1] = base * dfac;
                        }
                        if (use_w) {  // transform * (f - y), transform[:, None] * jac (the 2-point quotient of scaled residuals)
                            const double t = wsh[ii];
                            r0 *= t;
#pragma unroll
                            for (int k = 0; k < NP; ++k) ja[k] *= t;
                        }
                        if (HASFIXED) {
#pragma unroll
                            for (int k = 0; k < N; ++k) {
                                jr[k] = 0;
#pragma unroll
                                for (int j = 0; j < NP; ++j)
                                    if (A.free_idx[k] == j) jr[k] = ja[j];
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < N; ++k) jr[k] = ja[k];
                        }
                    }
                    const double rr = live ? r0 : 0.0;
                    finite_f = finite_f && isfinite(rr);
                    cost_new += rr * rr;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        const double jk = live ? jr[k] : 0.0;
                        gn[k] += jk * rr;
                    }
#pragma unroll
                    for (int k = 0; k < N; ++k) blk[r][k] = live ? jr[CP(k)] : 0.0;  // factor column k = parameter CP(k)
                    blk[r][N] = rr;
                }
                CFSTAMP(1);  // rows of the block: exp, model, Jacobian row, J^T f (diagnostic builds: -DPNX_CF_STAMP)
                qr_merge<N, kRowBlk>(Rn, qn, blk);
                CFSTAMP(6);  // the sequential Householder merge of the block into R
            }
            cost_new *= 0.5;
        }

        CFSTAMP(1);
        // ------------------------------------------------------------------ phase D: accept / reject
        bool accepted = false;
        int final_status = 0;
        if (state == ST_INIT) {
            if (!finite_f) {
                // non-finite signal: curve_fit's asarray_chkfinite raises (-2); otherwise least_squares.py:857-858
                // "Residuals are not finite in the initial point" (-4) -> failure sentinel either way
                final_status = yfinite ? -4 : -2;
                cost = NAN;
                nfev = 0;
                state = ST_FINAL;
            } else {
                accepted = true;
                nfev = 1;
                first = true;
                state = ST_RUN;
            }
        } else {  // ST_RUN
            nfev += 1;
            if (!finite_f) {
                Delta = 0.25 * step_h_norm;  // trf.py:337-339
            } else {
                const double actual = cost - cost_new;
                double ratio;  // update_tr_radius (common.py:222-245)
                if (predicted > 0)
                    ratio = actual / predicted;
                else if (predicted == 0 && actual == 0)
                    ratio = 1;
                else
                    ratio = 0;
                double Delta_new = Delta;
                if (ratio < 0.25)
                    Delta_new = 0.25 * step_h_norm;
                else if (ratio > 0.75 && step_h_norm > 0.95 * Delta)
                    Delta_new = Delta * 2.0;
                // check_termination (common.py:705-717)
                const bool ftol_ok = (actual < A.ftol * cost) && (ratio > 0.25);
                const bool xtol_ok = step_norm < A.xtol * (A.xtol + normn<N>(x));
                if (ftol_ok && xtol_ok)
                    term = 4;
                else if (ftol_ok)
                    term = 2;
                else if (xtol_ok)
                    term = 3;
                if (term == -99) {
                    alpha *= Delta / Delta_new;
                    Delta = Delta_new;
                }
                accepted = actual > 0;
            }
        }
        if (accepted) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                x[i] = xn[i];
                g[i] = gn[i];
            }
            cost = cost_new;
            int t = 0;
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int j = i; j < N; ++j) rpark[(t++) * kWave] = Rn[i][j];
        }

        // ------------------------------------------------------------------ light head of phase B
        // (trf.py:260-272: scaling vector, first trust radius, gtol test, leave when done or out of evaluations)
        double v[N], dv[N], g_norm = 0;
        if (state == ST_RUN && (accepted || term != -99 || nfev >= A.max_nfev)) {
#pragma unroll
            for (int i = 0; i < N; ++i) {  // CL_scaling_vector (common.py:467-508)
                v[i] = 1.0;
                dv[i] = 0.0;
                if (g[i] < 0 && isfinite(ub[i])) {
                    v[i] = ub[i] - x[i];
                    dv[i] = -1;
                }
                if (g[i] > 0 && isfinite(lb[i])) {
                    v[i] = x[i] - lb[i];
                    dv[i] = 1;
                }
                g_norm = fmax(g_norm, fabs(g[i] * v[i]));
            }
            if (first) {  // trf.py:232-236
                double t = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double q = x[i] / sqrt(v[i]);
                    t += q * q;
                }
                Delta = sqrt(t);
                if (Delta == 0) Delta = 1.0;
                first = false;
            }
            if (g_norm < A.gtol) term = 1;
            if (term != -99 || nfev >= A.max_nfev) {
                final_status = (term == -99) ? 0 : term;
                state = ST_FINAL;
            }
        }

        CFSTAMP(2);
        CFACT(3, state == ST_FINAL);
        // ------------------------------------------------------------------ outputs of finished voxels
        if (state == ST_FINAL) {
            const bool ok = final_status > 0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                // failure: the reference returns the p0 it was given (curvefit.py:308-317)
                const double pk = ok ? x[k] : (PV ? A.p0[(size_t)k * A.n_vox + vox] : A.p0s[k]);
                A.popt[(size_t)k * A.n_vox + vox] = pk;
            }
            if (A.status) A.status[vox] = (int8_t)final_status;
            if (A.nfev) A.nfev[vox] = nfev;
            if (A.cost) A.cost[vox] = cost;
            if (A.pcov && ok) {
                // The covariance needs another n x n SVD; done here it would run with a handful of lanes active
                // in (almost) every loop iteration.  The packed R factor of the final Jacobian goes to the voxel's
                // pcov slot instead and pcov_kernel turns it into pcov with all lanes busy.
                double *pc = A.pcov + (size_t)vox * N * N;
#pragma unroll
                for (int t = 0; t < PK::NR; ++t) pc[t] = rpark[t * kWave];
            }
            state = ST_IDLE;
        }
        CFACT(0, state == ST_IDLE);
        if constexpr (STREAM) {
            const bool any_idle = __ballot(state == ST_IDLE) != 0;
            if (state == ST_IDLE) refill();
            if (any_idle) publish(false);
        } else {
            if (state == ST_IDLE) refill();  // claim the next voxel now: its signal streams in behind phases B / C
        }

        CFSTAMP(3);
        CFACT(4, state == ST_RUN && accepted);
        CFACT(5, state == ST_RUN);
        if (state == ST_RUN) {
            // -------------------------------------------------------------- heavy phase B (iterate changed)
            if (accepted) {
                double diag_h[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    d[i] = sqrt(v[i]);
                    diag_h[i] = g[i] * dv[i];
                    g_h[i] = d[i] * g[i];
                }
                // QR of [R*D; diag(sqrt(diag_h))] -> R2, q2 ;  J_aug = Q R2 (trf.py:300-306)
                double q2[N];
                double blk[N][N + 1];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    q2[i] = qn[i];
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        R2[i][j] = (j >= i) ? Rn[i][j] * d[CP(j)] : 0.0;
                        blk[i][j] = (i == j) ? sqrt(diag_h[CP(i)]) : 0.0;
                    }
                    blk[i][N] = 0.0;
                }
                qr_merge<N, N>(R2, q2, blk);
                // SVD of R2 through one-sided Jacobi on W = R2^T (lower triangular):  W Vw = Uw S  =>
                // R2 = Vw S Uw^T: right singular vectors of J_aug = normalised columns of W, uf = Vw^T q2.
                double W[N][N], Vw[N][N];
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = 0; j < N; ++j) W[i][j] = (i >= j) ? R2[j][i] : 0.0;
                jacobi_svd<N>(W, Vw);
                smax = 0;
                smin = INFINITY;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    double nn = 0, dq = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        nn += W[i][k] * W[i][k];
                        dq += Vw[i][k] * q2[i];
                    }
                    double inv = 0.0;  // singular value and its reciprocal from one v_rsq_f64 seed (normalisation only: ~1 ulp is fine)
                    if (nn > 0) nn = fast_sqrt_rsqrt(nn, &inv);
                    s[k] = nn;
                    uf[k] = dq;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        if (PK::kParkV)
                            vpark[(i * N + k) * kWave] = W[i][k] * inv;
                        else
                            Vreg[PK::kParkV ? 0 : i][PK::kParkV ? 0 : k] = W[i][k] * inv;
                    }
                    smax = fmax(smax, nn);
                    smin = fmin(smin, nn);
                }
                theta = fmax(0.995, 1 - g_norm);
            }
            CFSTAMP(4);
            // -------------------------------------------------------------- phase C: trial step
            double V[N][N];
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int k = 0; k < N; ++k)
                    V[i][k] = PK::kParkV ? vpark[(i * N + k) * kWave] : Vreg[PK::kParkV ? 0 : i][PK::kParkV ? 0 : k];
            // everything below works in the factor's column order (a relabelling of the parameters)
            double xP[N], lbP[N], ubP[N], dP[N], ghP[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                xP[i] = x[CP(i)];
                lbP[i] = lb[CP(i)];
                ubP[i] = ub[CP(i)];
                dP[i] = d[CP(i)];
                ghP[i] = g_h[CP(i)];
            }
            double p_h[N], p[N], step[N], step_h[N];
            alpha = solve_lsq_trust_region<N>(n_b, uf, s, V, smax, smin, Delta, alpha, p_h);
#pragma unroll
            for (int i = 0; i < N; ++i) p[i] = dP[i] * p_h[i];
            predicted = select_step<N>(xP, R2, ghP, p, p_h, dP, Delta, lbP, ubP, theta, step, step_h);
#pragma unroll
            for (int i = 0; i < N; ++i) xn[CP(i)] = strictly_feasible0(xP[i] + step[i], lbP[i], ubP[i]);
            step_h_norm = normn<N>(step_h);
            step_norm = normn<N>(step);
            CFSTAMP(5);
        }
    }
#ifdef PNX_CF_STAMP
    if (threadIdx.x == 0 && blockIdx.x == 3)
        printf("CFSTAMP refill=%llu pass_rows=%llu pass_qr_merge=%llu D+Blight=%llu final=%llu Bheavy=%llu C=%llu loop=%llu | iters=%llu idle_at_top=%.2f final=%.2f Bheavy=%.2f run=%.2f\n",
               seg[0], seg[1], seg[6], seg[2], seg[3], seg[4], seg[5], seg[7], cnt[1], (double)act[0] / cnt[0], (double)act[3] / cnt[3],
               (double)act[4] / cnt[4], (double)act[5] / cnt[5]);
#endif
}

// ---------------------------------------------------------------------------------------------
// Covariance epilogue (one lane per voxel, fully convergent): turns the packed R factor of the final
// Jacobian that curvefit_kernel parked in pcov[vox] into SciPy's pcov
//   pinv(J^T J) * 2 cost / (m - n)   with singular values <= eps * max(m, n) * s_max dropped  (absolute_sigma: unscaled)
// (scipy/optimize/_minpack_py.py:1036-1066), NaN for failed voxels (curvefit.py:236-243, 312-317).
// ---------------------------------------------------------------------------------------------
struct ColPerm {
    int p[kMaxP];
};

template <int N>
__global__ void __launch_bounds__(256) pcov_kernel(double *pcov, const int8_t *status, const double *cost,
                                                   long long n_vox, int n_b, const ColPerm cp, int absolute_sigma) {
    const long long vox = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vox >= n_vox) return;
    double *pc = pcov + (size_t)vox * N * N;
    if (status[vox] <= 0) {
#pragma unroll
        for (int k = 0; k < N * N; ++k) pc[k] = NAN;
        return;
    }
    double W[N][N], Vw[N][N];
    {
        int t = 0;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) W[i][j] = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = i; j < N; ++j) W[j][i] = pc[t++];  // W = R^T
    }
    jacobi_svd<N>(W, Vw);
    double s2[N], sm = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double nn = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) nn += W[i][k] * W[i][k];
        s2[k] = nn;
        sm = fmax(sm, nn);
    }
    const double thr = kEps * (n_b > N ? n_b : N) * sqrt(sm);
    const bool dof = absolute_sigma || n_b > N;  // absolute_sigma: no scaling, hence no degrees-of-freedom condition (_minpack_py.py:1057-1063)
    const double s_sq = absolute_sigma ? 1.0 : (dof ? 2.0 * cost[vox] / (double)(n_b - N) : 0.0);
    double wgt[N];
#pragma unroll
    for (int k = 0; k < N; ++k) wgt[k] = (sqrt(s2[k]) > thr) ? 1.0 / (s2[k] * s2[k]) : 0.0;  // V_k V_k^T / s_k^2 = w_k w_k^T / s_k^4
    bool bad = false;
    double out[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double a = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) a += W[i][k] * W[j][k] * wgt[k];
            out[i][j] = a;
            bad = bad || isnan(a);
        }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) pc[cp.p[i] * N + cp.p[j]] = (bad || !dof) ? INFINITY : out[i][j] * s_sq;  // factor order -> parameter order
}

}  // namespace pnx
