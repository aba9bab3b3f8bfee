/*
 * pnx_oracle_nnls.c -- CPU restatement (plain C, fp64, scalar) of the reference's
 * per-voxel regularised NNLS path.
 *
 * TEST INFRASTRUCTURE ONLY (see pnx_oracle_trf.c header): never linked, loaded or
 * called by the product; used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.
 *
 * What it restates:
 *   - src/pyneapple/solvers/nnls_solver.py:61-127   A = [basis; reg], y_ext = [y | 0]
 *   - src/pyneapple/solvers/nnls_solver.py:182-210  per-voxel solve; failure => zeros, ||y_ext||
 *   - scipy.optimize.nnls (SciPy 1.15.3, scipy/optimize/_nnls.py:8-97).  Its kernel is the
 *     compiled module scipy/optimize/_cython_nnls (no source on disk; links dlarfgp / dlarf /
 *     dlartgp / dnrm2), i.e. the Lawson & Hanson (1995, ch. 23) active-set algorithm with
 *     Householder column elimination when an index enters the passive set and Givens
 *     rotations when one leaves it.  The published algorithm is restated here step by step
 *     (same selection rule, same 0.01 linear-independence test, same ztest rejection, same
 *     alpha interpolation, same round-off clean-up loop, `iteration == maxiter` => failure).
 *
 * Pinning: tests/golden/g4_nnls_*.npz (outputs of the reference's NNLSSolver run in the
 * survey container, incl. a max_iter failure case) and direct comparison with
 * scipy.optimize.nnls in tests/test_oracle_nnls.py (coefficients, rnorm and the smallest
 * maxiter that succeeds).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* A is column-major (m x n), lda = m. */
#define AT(i, j) A[(size_t)(j) * m + (i)]

/* Householder vector for x[0..len): returns beta (>= 0, LAPACK dlarfgp convention), tau,
 * and overwrites x[1..] with v[1..] (v[0] = 1 implied). */
static double house_gen(double *x, int len, double *tau)
{
    double alpha = x[0];
    double xnorm = 0;
    for (int i = 1; i < len; ++i) xnorm += x[i] * x[i];
    xnorm = sqrt(xnorm);
    if (xnorm == 0) {
        if (alpha >= 0) {
            *tau = 0;
            return alpha;
        }
        *tau = 2;
        for (int i = 1; i < len; ++i) x[i] = 0;
        return -alpha;
    }
    const double nrm = hypot(alpha, xnorm);
    double a2;
    if (alpha < 0) {
        a2 = alpha - nrm;
        *tau = -a2 / nrm;
    } else {
        a2 = xnorm * (xnorm / (alpha + nrm));
        *tau = a2 / nrm;
        a2 = -a2;
    }
    const double beta = nrm;
    const double sc = 1.0 / a2;
    for (int i = 1; i < len; ++i) x[i] *= sc;
    return beta;
}

/* apply H = I - tau v v^T (v[0]=1) to vector c[0..len) */
static void house_apply(const double *v, double tau, int len, double *c)
{
    if (tau == 0) return;
    double s = c[0];
    for (int i = 1; i < len; ++i) s += v[i] * c[i];
    s *= tau;
    c[0] -= s;
    for (int i = 1; i < len; ++i) c[i] -= s * v[i];
}

/* Givens with non-negative r (dlartgp convention) */
static void givens_gen(double f, double g, double *cs, double *sn, double *r)
{
    if (g == 0) {
        *cs = copysign(1.0, f);
        *sn = 0;
        *r = fabs(f);
    } else if (f == 0) {
        *cs = 0;
        *sn = copysign(1.0, g);
        *r = fabs(g);
    } else {
        double rr = hypot(f, g);
        *cs = f / rr;
        *sn = g / rr;
        *r = rr;
    }
}

static void tri_solve(const double *A, int m, const int *inds, int nsetp, double *zz)
{
    for (int l = 0; l < nsetp; ++l) {
        int ip = nsetp - 1 - l;
        if (l != 0) {
            int jj = inds[ip + 1];
            double z = zz[ip + 1];
            for (int ii = 0; ii <= ip; ++ii) zz[ii] -= AT(ii, jj) * z;
        }
        zz[ip] /= AT(ip, inds[ip]);
    }
}

/* Test hook shared with the HIP block kernel (pnx_nnls_blk.hip, PNX_NNLS_TEST_REJECT=k,n): with nsetp % k == k - 1 the first n
 * candidates of an outer iteration are rejected unseen, so that the kernel's bookkeeping of rejected columns -- which the
 * reference workload never exercises -- can be checked against this restatement.  The kernel's list holds eight columns; a
 * ninth rejection hands the voxel to the kernel's hand-over target (since the last part of round 4 its four-slot instantiation,
 * before that the Gram-form kernel), which is run without the hook: mode -2 asks the caller to solve the voxel again without it. */
static int g_rej_k = 0, g_rej_n = 0;
#define PNX_ORACLE_MAX_REJ 8

/* returns mode: 1 ok, -1 iteration limit, -2 (hook only) solve again without the hook */
static int nnls_one(double *A, int m, int n, double *b, int maxiter, double *x, double *rnorm, int *iters,
                    double *w, double *zz, double *work, int *inds, int rej_k, int rej_n)
{
    int iz1 = 0, nsetp = 0, iteration = 0, skip = 0, izmax = 0, nrej = 0;
    for (int j = 0; j < n; ++j) {
        x[j] = 0;
        w[j] = 0;
        inds[j] = j;
    }
    int mode = 1;
    while (iz1 < n && nsetp < m) {
        if (skip)
            skip = 0;
        else {
            for (int iz = iz1; iz < n; ++iz) {
                int col = inds[iz];
                double sm = 0;
                for (int l = nsetp; l < m; ++l) sm += AT(l, col) * b[l];
                w[col] = sm;
            }
        }
        double wmax = 0;
        for (int iz = iz1; iz < n; ++iz) {
            int col = inds[iz];
            if (w[col] > wmax) {
                wmax = w[col];
                izmax = iz;
            }
        }
        if (wmax <= 0) break;
        if (rej_k > 0 && nsetp >= 128) return -2; /* hook: the block kernel hands voxels with more than 128 passive bins over, too */
        int iz = izmax, j = inds[iz];

        int len = m - nsetp;
        for (int l = 0; l < len; ++l) work[l] = AT(nsetp + l, j);
        double tau;
        double beta = house_gen(work, len, &tau);
        double unorm = 0;
        for (int l = 0; l < nsetp; ++l) unorm += AT(l, j) * AT(l, j);
        unorm = sqrt(unorm);
        double ztest = 0;
        int accept = 0;
        const int forced = rej_k > 0 && (nsetp % rej_k) == rej_k - 1 && nrej < rej_n;
        if (!forced && ((unorm + fabs(beta) * 0.01) - unorm) > 0) {
            memcpy(zz, b, sizeof(double) * m);
            house_apply(work, tau, len, zz + nsetp);
            ztest = zz[nsetp] / beta;
            if (ztest > 0) accept = 1;
        }
        if (!accept) {
            if (rej_k > 0 && nrej >= PNX_ORACLE_MAX_REJ) return -2;
            nrej += 1;
            w[j] = 0;
            skip = 1;
            continue;
        }
        nrej = 0;
        /* column j enters the passive set */
        memcpy(b, zz, sizeof(double) * m);
        inds[iz] = inds[iz1];
        inds[iz1] = j;
        iz1 += 1;
        for (int jz = iz1; jz < n; ++jz) house_apply(work, tau, len, &AT(nsetp, inds[jz]));
        AT(nsetp, j) = beta;
        for (int l = nsetp + 1; l < m; ++l) AT(l, j) = 0;
        nsetp += 1;
        w[j] = 0;
        tri_solve(A, m, inds, nsetp, zz);

        for (;;) {
            iteration += 1;
            if (iteration == maxiter) {
                mode = -1;
                goto done;
            }
            double alpha = 2.0;
            int jj = 0;
            for (int ip = 0; ip < nsetp; ++ip) {
                int k = inds[ip];
                if (zz[ip] <= 0) {
                    double T = -x[k] / (zz[ip] - x[k]);
                    if (alpha > T) {
                        alpha = T;
                        jj = ip;
                    }
                }
            }
            if (alpha == 2.0) break;
            for (int ip = 0; ip < nsetp; ++ip) {
                int k = inds[ip];
                x[k] = x[k] + alpha * (zz[ip] - x[k]);
            }
            int i = inds[jj];
            for (;;) {
                x[i] = 0;
                if (jj != nsetp - 1) {
                    jj += 1;
                    for (int jc = jj; jc < nsetp; ++jc) {
                        int ii = inds[jc];
                        inds[jc - 1] = ii;
                        double cc, ss, r;
                        givens_gen(AT(jc - 1, ii), AT(jc, ii), &cc, &ss, &r);
                        AT(jc - 1, ii) = r;
                        AT(jc, ii) = 0;
                        for (int col = 0; col < n; ++col) {
                            if (col == ii) continue;
                            double t1 = AT(jc - 1, col), t2 = AT(jc, col);
                            AT(jc - 1, col) = cc * t1 + ss * t2;
                            AT(jc, col) = -ss * t1 + cc * t2;
                        }
                        double t1 = b[jc - 1], t2 = b[jc];
                        b[jc - 1] = cc * t1 + ss * t2;
                        b[jc] = -ss * t1 + cc * t2;
                    }
                }
                nsetp -= 1;
                iz1 -= 1;
                inds[iz1] = i;
                int again = 0;
                for (jj = 0; jj < nsetp; ++jj) {
                    i = inds[jj];
                    if (x[i] <= 0) {
                        again = 1;
                        break;
                    }
                }
                if (!again) break;
            }
            memcpy(zz, b, sizeof(double) * m);
            tri_solve(A, m, inds, nsetp, zz);
        }
        for (int ip = 0; ip < nsetp; ++ip) x[inds[ip]] = zz[ip];
    }
done:;
    double sm = 0;
    for (int l = nsetp; l < m; ++l) sm += b[l] * b[l];
    *rnorm = sqrt(sm);
    *iters = iteration;
    return mode;
}

/*
 * Batched driver; layouts mirror include/pnx.h:
 *   basis (n_meas, n_bins) row-major; reg (n_reg, n_bins) row-major or NULL; y (n_vox, n_meas);
 *   coeff (n_vox, n_bins); rnorm (n_vox); status 1 ok / 0 iteration limit / -2 non-finite input.
 * Failure: coeff = 0, rnorm = ||y_ext|| = ||y||  (nnls_solver.py:201-210).
 */
int pnxo_nnls_batch(long n_vox, int n_meas, int n_bins, const double *basis, const double *reg, int n_reg,
                    const double *y, int max_iter, double *coeff, double *rnorm, int8_t *status, int32_t *iters,
                    int n_threads)
{
    if (n_meas < 1 || n_bins < 1 || n_reg < 0) return -1;
    const int m = n_meas + n_reg, n = n_bins;
    if (!max_iter) max_iter = 3 * n;
    g_rej_k = g_rej_n = 0;
    {
        const char *t = getenv("PNX_NNLS_TEST_REJECT");
        if (t && (sscanf(t, "%d,%d", &g_rej_k, &g_rej_n) != 2 || g_rej_k < 1 || g_rej_n < 1)) g_rej_k = g_rej_n = 0;
    }
    /* column-major master copy of A */
    double *A0 = (double *)malloc(sizeof(double) * (size_t)m * n);
    if (!A0) return -2;
    int finite_A = 1;
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n_meas; ++i) A0[(size_t)j * m + i] = basis[(size_t)i * n + j];
        for (int i = 0; i < n_reg; ++i) A0[(size_t)j * m + n_meas + i] = reg[(size_t)i * n + j];
    }
    for (size_t k = 0; k < (size_t)m * n; ++k)
        if (!isfinite(A0[k])) finite_A = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#endif
    {
        double *A = (double *)malloc(sizeof(double) * (size_t)m * n);
        double *buf = (double *)malloc(sizeof(double) * ((size_t)3 * m + 2 * n));
        int *inds = (int *)malloc(sizeof(int) * n);
        double *b = buf, *zz = buf + m, *work = buf + 2 * m, *w = buf + 3 * m, *x = buf + 3 * m + n;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (long vx = 0; vx < n_vox; ++vx) {
            const double *yv = y + (size_t)vx * n_meas;
            int finite = finite_A;
            double yn = 0;
            for (int i = 0; i < n_meas; ++i) {
                if (!isfinite(yv[i])) finite = 0;
                b[i] = yv[i];
                yn += yv[i] * yv[i];
            }
            for (int i = n_meas; i < m; ++i) b[i] = 0;
            int st, it = 0;
            double rn = 0;
            if (!finite) {
                st = -2; /* asarray_chkfinite -> ValueError -> failure path */
            } else {
                memcpy(A, A0, sizeof(double) * (size_t)m * n);
                int mode = nnls_one(A, m, n, b, max_iter, x, &rn, &it, w, zz, work, inds, g_rej_k, g_rej_n);
                if (mode == -2) { /* hook: the kernel's hand-over to its general kernel */
                    memcpy(A, A0, sizeof(double) * (size_t)m * n);
                    for (int i = 0; i < n_meas; ++i) b[i] = yv[i];
                    for (int i = n_meas; i < m; ++i) b[i] = 0;
                    mode = nnls_one(A, m, n, b, max_iter, x, &rn, &it, w, zz, work, inds, 0, 0);
                }
                st = (mode == 1) ? 1 : 0;
            }
            double *cv = coeff + (size_t)vx * n;
            if (st == 1) {
                memcpy(cv, x, sizeof(double) * n);
                rnorm[vx] = rn;
            } else {
                for (int j = 0; j < n; ++j) cv[j] = 0;
                rnorm[vx] = sqrt(yn); /* NaN propagates like np.linalg.norm */
            }
            if (status) status[vx] = (int8_t)st;
            if (iters) iters[vx] = it;
        }
        free(A);
        free(buf);
        free(inds);
    }
    free(A0);
    return 0;
}
