/* oracle/selftest.c -- sanitizer harness for the CPU restatement (test infrastructure, like everything in oracle/).
 * Built by `make -C oracle sanitize` with -fsanitize=address,undefined and run once: a few hundred curve fits of every
 * model (shared and per-voxel start values, one fixed parameter, T1 factor, NaN / out-of-bounds failure cases) and
 * NNLS solves with every regularisation order.  Any out-of-bounds access, use of uninitialised stack through UBSan's
 * checks, signed overflow or misaligned access aborts with a report; the exit status is what tests/ looks at.
 * (sanitizers run on the CPU build only -- GPU ASan is not available on the pool.) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int pnxo_model_n_all(int model, int t1_mode);
int pnxo_curvefit_batch(int model, int t1_mode, double tr, double tm, long n_vox, int n_b, const double *b,
                        const double *y, int n_free, const int *free_idx, int n_fixed, const int *fixed_idx,
                        const double *fixed_vals, int fixed_per_voxel, const double *p0, const double *lo,
                        const double *hi, int per_voxel, int max_nfev, double ftol, double xtol, double gtol,
                        int jac_mode, double *popt, double *pcov, int8_t *status, int32_t *nfev, double *cost,
                        int n_threads);
int pnxo_nnls_batch(long n_vox, int n_meas, int n_bins, const double *basis, const double *reg, int n_reg,
                    const double *y, int max_iter, double *coeff, double *rnorm, int8_t *status, int32_t *iters,
                    int n_threads);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

/* model parameter layouts (reference order); D positions flagged */
static const int NALL[7] = {2, 3, 4, 4, 5, 6, 6};
static void truth(int model, double *p) {
    const double f1 = 0.1 + 0.2 * urand(), f2 = 0.2 + 0.2 * urand();
    const double D1 = 0.03 + 0.05 * urand(), D2 = 0.003 + 0.004 * urand(), D3 = 0.0005 + 0.001 * urand();
    const double S0 = 500 + 1000 * urand();
    switch (model) {
    case 0: p[0] = S0; p[1] = D3; break;
    case 1: p[0] = f1; p[1] = D1; p[2] = D3; break;
    case 2: p[0] = f1; p[1] = D1; p[2] = D3; p[3] = S0; break;
    case 3: p[0] = f1 * S0; p[1] = D1; p[2] = (1 - f1) * S0; p[3] = D3; break;
    case 4: p[0] = f1; p[1] = D1; p[2] = f2; p[3] = D2; p[4] = D3; break;
    case 5: p[0] = f1; p[1] = D1; p[2] = f2; p[3] = D2; p[4] = D3; p[5] = S0; break;
    default: p[0] = f1 * S0; p[1] = D1; p[2] = f2 * S0; p[3] = D2; p[4] = (1 - f1 - f2) * S0; p[5] = D3; break;
    }
}
static double signal(int model, const double *p, double b) {
    switch (model) {
    case 0: return p[0] * exp(-b * p[1]);
    case 1: return p[0] * exp(-b * p[1]) + (1 - p[0]) * exp(-b * p[2]);
    case 2: return p[3] * (p[0] * exp(-b * p[1]) + (1 - p[0]) * exp(-b * p[2]));
    case 3: return p[0] * exp(-b * p[1]) + p[2] * exp(-b * p[3]);
    case 4: return p[0] * exp(-b * p[1]) + p[2] * exp(-b * p[3]) + (1 - p[0] - p[2]) * exp(-b * p[4]);
    case 5: return p[5] * (p[0] * exp(-b * p[1]) + p[2] * exp(-b * p[3]) + (1 - p[0] - p[2]) * exp(-b * p[4]));
    default: return p[0] * exp(-b * p[1]) + p[2] * exp(-b * p[3]) + p[4] * exp(-b * p[5]);
    }
}
static void bounds(int model, int k, double *lo, double *hi, const double *pt) {
    /* generous box around the truth, wide enough that several bounds become active on noisy data */
    (void)model;
    *lo = pt[k] * 0.2;
    *hi = pt[k] * 5.0;
}

int main(void) {
    enum { NV = 96, NB = 24 };
    double b[NB];
    for (int i = 0; i < NB; ++i) b[i] = 1200.0 * i / (NB - 1);
    long checked = 0;
    for (int model = 0; model < 7; ++model)
        for (int variant = 0; variant < 4; ++variant) { /* 0 shared FD, 1 per-voxel analytic, 2 one fixed, 3 T1 free */
            const int t1 = variant == 3 ? 1 : 0;
            const int n_all = NALL[model] + t1;
            if (pnxo_model_n_all(model, t1) != n_all) return 2;
            const int n_fixed = variant == 2 ? 1 : 0, n_free = n_all - n_fixed;
            int free_idx[8], fixed_idx[8];
            for (int k = 0, f = 0; k < n_all; ++k) {
                if (n_fixed && k == 1) fixed_idx[0] = k;
                else free_idx[f++] = k;
            }
            const int pv = variant == 1;
            double *y = malloc(sizeof(double) * NV * NB), *p0 = malloc(sizeof(double) * n_free * NV);
            double *lo = malloc(sizeof(double) * n_free * NV), *hi = malloc(sizeof(double) * n_free * NV);
            double *fx = malloc(sizeof(double) * NV), *popt = malloc(sizeof(double) * n_free * NV);
            double *pcov = malloc(sizeof(double) * NV * n_free * n_free), *cost = malloc(sizeof(double) * NV);
            int8_t *status = malloc(NV);
            int32_t *nfev = malloc(sizeof(int32_t) * NV);
            for (int v = 0; v < NV; ++v) {
                double pt[8];
                truth(model, pt);
                if (t1) pt[n_all - 1] = 800 + 800 * urand();
                const double fac = t1 ? 1 - exp(-3000.0 / pt[n_all - 1]) : 1.0;
                for (int i = 0; i < NB; ++i) y[v * NB + i] = signal(model, pt, b[i]) * fac * (1 + 0.02 * (urand() - 0.5));
                fx[v] = pt[1];
                for (int k = 0; k < n_free; ++k) {
                    const int j = free_idx[k];
                    double l, h;
                    bounds(model, j, &l, &h, pt);
                    const double s = pt[j] * (0.7 + 0.6 * urand());
                    if (pv) {
                        p0[k * NV + v] = s; lo[k * NV + v] = l; hi[k * NV + v] = h;
                    } else if (v == 0) {
                        p0[k] = s; lo[k] = pt[j] * 0.05; hi[k] = pt[j] * 20.0;
                    }
                }
            }
            /* failure sentinels: non-finite signal, p0 outside the box, lb == ub */
            y[3 * NB + 5] = NAN;
            if (pv) {
                p0[0 * NV + 7] = hi[0 * NV + 7] * 2;
                lo[0 * NV + 9] = hi[0 * NV + 9];
            }
            const int rc = pnxo_curvefit_batch(model, t1, 3000.0, 0.0, NV, NB, b, y, n_free, free_idx, n_fixed, fixed_idx, fx,
                                               1, p0, lo, hi, pv, variant == 0 ? 6 : 250, 1e-8, 1e-8, 1e-8,
                                               (variant == 0 || variant == 3) ? 0 : 1, popt, pcov, status, nfev, cost, 2);
            if (rc) { fprintf(stderr, "curvefit rc=%d model=%d variant=%d\n", rc, model, variant); return 3; }
            if (status[3] != -2) { fprintf(stderr, "NaN voxel not flagged (model %d variant %d)\n", model, variant); return 4; }
            if (pv && (status[7] != -3 || status[9] != -1)) { fprintf(stderr, "bound sentinels wrong\n"); return 5; }
            for (int v = 0; v < NV; ++v) checked += status[v] > 0;
            free(y); free(p0); free(lo); free(hi); free(fx); free(popt); free(pcov); free(cost); free(status); free(nfev);
        }
    /* NNLS: every regularisation order incl. none (rank-deficient Gram), tiny iteration limit, NaN signal */
    enum { NM = 16, NBINS = 50, NVN = 48 };
    double bn[NM], bins[NBINS], *basis = malloc(sizeof(double) * NM * NBINS), *reg = calloc(NBINS * NBINS, sizeof(double));
    for (int i = 0; i < NM; ++i) bn[i] = 1000.0 * i / (NM - 1);
    for (int j = 0; j < NBINS; ++j) bins[j] = pow(10.0, -4.0 + 3.0 * j / (NBINS - 1));
    for (int i = 0; i < NM; ++i)
        for (int j = 0; j < NBINS; ++j) basis[i * NBINS + j] = exp(-bn[i] * bins[j]);
    for (int order = -1; order < 4; ++order) {
        memset(reg, 0, sizeof(double) * NBINS * NBINS);
        for (int j = 0; j < NBINS; ++j) {
            if (order == 1) { reg[j * NBINS + j] = -0.02; if (j + 1 < NBINS) reg[j * NBINS + j + 1] = 0.02; }
            if (order == 2) { reg[j * NBINS + j] = -0.04; if (j + 1 < NBINS) reg[j * NBINS + j + 1] = 0.02; if (j) reg[j * NBINS + j - 1] = 0.02; }
            if (order == 3) {
                reg[j * NBINS + j] = -0.12;
                if (j + 1 < NBINS) reg[j * NBINS + j + 1] = 0.04;
                if (j) reg[j * NBINS + j - 1] = 0.04;
                if (j + 2 < NBINS) reg[j * NBINS + j + 2] = 0.02;
                if (j > 1) reg[j * NBINS + j - 2] = 0.02;
            }
        }
        double *y = malloc(sizeof(double) * NVN * NM), *coeff = malloc(sizeof(double) * NVN * NBINS), *rn = malloc(sizeof(double) * NVN);
        int8_t *status = malloc(NVN);
        int32_t *iters = malloc(sizeof(int32_t) * NVN);
        for (int v = 0; v < NVN; ++v) {
            const double f = 0.2 + 0.3 * urand(), Da = 0.02 + 0.03 * urand(), Db = 0.001 + 0.002 * urand();
            for (int i = 0; i < NM; ++i) y[v * NM + i] = 1000 * (f * exp(-bn[i] * Da) + (1 - f) * exp(-bn[i] * Db)) * (1 + 0.01 * (urand() - 0.5));
        }
        y[5 * NM + 2] = NAN;
        const int rc = pnxo_nnls_batch(NVN, NM, NBINS, basis, order < 0 ? NULL : reg, order < 0 ? 0 : NBINS, y, order == 0 ? 3 : 250,
                                       coeff, rn, status, iters, 2);
        if (rc) { fprintf(stderr, "nnls rc=%d order=%d\n", rc, order); return 6; }
        if (status[5] != -2) { fprintf(stderr, "NaN NNLS voxel not flagged\n"); return 7; }
        for (int v = 0; v < NVN; ++v) {
            checked += status[v] == 1;
            for (int j = 0; j < NBINS; ++j)
                if (!(coeff[v * NBINS + j] >= 0)) { fprintf(stderr, "negative / NaN coefficient\n"); return 8; }
        }
        free(y); free(coeff); free(rn); free(status); free(iters);
    }
    free(basis); free(reg);
    printf("oracle selftest ok: %ld converged solves under the sanitizers\n", checked);
    return checked > 1500 ? 0 : 9;
}
