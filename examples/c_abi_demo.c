/* examples/c_abi_demo.c -- the C ABI of libpnx_hip.so used from plain C (no Python, no torch).
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Lpyneapple_amd -lpnx_hip -Wl,-rpath,$PWD/pyneapple_amd -lm -o c_abi_demo
 *
 * Fits 10 000 synthetic bi-exponential voxels (host arrays) and solves 2 000 NNLS spectra, checks the recovered
 * parameters against the truth and prints one line per call.  Exit status 0 = all checks passed, 77 = no HIP device.
 * tests/test_c_abi_demo.py compiles it (CPU) and runs it (GPU). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pnx.h"

static uint64_t rs = 88172645463325252ull;
static double urand(void) {
    rs ^= rs << 13;
    rs ^= rs >> 7;
    rs ^= rs << 17;
    return (double)(rs >> 11) / 9007199254740992.0;
}

static int fail(const char *what) {
    char msg[512];
    pnx_last_error(msg, (int)sizeof(msg));
    fprintf(stderr, "%s failed: %s\n", what, msg);
    return 1;
}

int main(void) {
    if (pnx_device_count() < 1) {
        printf("no HIP device visible: nothing to run\n");
        return 77;
    }
    printf("libpnx version %d, %d device(s)\n", pnx_version(), pnx_device_count());
    enum { NV = 10000, NB = 16 };
    double b[NB];
    for (int i = 0; i < NB; ++i) b[i] = 1000.0 * i / (NB - 1);
    double *y = malloc(sizeof(double) * NV * NB), *truth = malloc(sizeof(double) * 3 * NV);
    for (int v = 0; v < NV; ++v) {
        const double f1 = 0.1 + 0.3 * urand(), D1 = 0.005 + 0.045 * urand(), D2 = 0.0005 + 0.0015 * urand();
        truth[v] = f1; truth[NV + v] = D1; truth[2 * NV + v] = D2;
        for (int i = 0; i < NB; ++i) y[v * NB + i] = f1 * exp(-b[i] * D1) + (1 - f1) * exp(-b[i] * D2);
    }
    pnx_curvefit_opts o;
    memset(&o, 0, sizeof(o));
    o.model = PNX_MODEL_BI_REDUCED;
    o.n_b = NB;
    o.n_free = pnx_model_n_params(PNX_MODEL_BI_REDUCED);
    for (int k = 0; k < o.n_free; ++k) o.free_idx[k] = k;
    o.max_nfev = 250;
    o.jac_mode = PNX_JAC_FD;
    o.ftol = o.xtol = o.gtol = 1e-8;
    const double p0[3] = {0.2, 0.01, 0.001}, lo[3] = {0.0, 1e-3, 1e-5}, hi[3] = {1.0, 0.1, 5e-3};
    double *popt = malloc(sizeof(double) * 3 * NV), *pcov = malloc(sizeof(double) * 9 * NV), *cost = malloc(sizeof(double) * NV);
    int8_t *status = malloc(NV);
    int32_t *nfev = malloc(sizeof(int32_t) * NV);
    if (pnx_curvefit_batch_f64(&o, NV, b, y, p0, lo, hi, NULL, popt, pcov, status, nfev, cost, PNX_MEM_HOST, 0, NULL))
        return fail("pnx_curvefit_batch_f64");
    int ok = 0;
    double worst = 0;
    for (int v = 0; v < NV; ++v) {
        if (status[v] <= 0) continue;
        ++ok;
        for (int k = 0; k < 3; ++k) {
            const double e = fabs(popt[k * NV + v] - truth[k * NV + v]) / truth[k * NV + v];
            if (e > worst) worst = e;
        }
    }
    printf("curve fit: %d of %d voxels converged, worst relative parameter error %.2e (noise-free data)\n", ok, NV, worst);
    if (ok != NV || worst > 1e-4) return 2;
    /* invalid arguments come back as error codes with a message, not as a crash */
    o.n_b = PNX_MAX_BVALUES + 1;
    if (pnx_curvefit_batch_f64(&o, NV, b, y, p0, lo, hi, NULL, popt, NULL, status, NULL, NULL, PNX_MEM_HOST, 0, NULL) != PNX_ERR_INVALID) return 3;

    enum { NVN = 2000, NBINS = 100 };
    double bins[NBINS], *basis = malloc(sizeof(double) * NB * NBINS), *reg = malloc(sizeof(double) * NBINS * NBINS);
    if (pnx_nnls_bins(1e-4, 0.2, NBINS, bins) || pnx_nnls_basis(NB, b, NBINS, bins, basis, 0) ||
        pnx_nnls_regularization_matrix(NBINS, 2, 0.02, reg))
        return fail("NNLS builders");
    pnx_nnls_plan *plan = NULL;
    if (pnx_nnls_plan_create(&plan, NB, NBINS, basis, reg, NBINS, 0)) return fail("pnx_nnls_plan_create");
    double *coeff = malloc(sizeof(double) * NVN * NBINS), *rnorm = malloc(sizeof(double) * NVN);
    int32_t *iters = malloc(sizeof(int32_t) * NVN);
    for (int v = 0; v < NVN * NB; ++v) y[v] *= 1000.0;
    if (pnx_nnls_solve_f64(plan, NVN, y, 250, coeff, rnorm, status, iters, PNX_MEM_HOST, NULL)) return fail("pnx_nnls_solve_f64");
    int okn = 0, neg = 0;
    double worst_fit = 0;
    for (int v = 0; v < NVN; ++v) {
        okn += status[v] == 1;
        double ny = 0;
        for (int i = 0; i < NB; ++i) ny += y[v * NB + i] * y[v * NB + i];
        for (int j = 0; j < NBINS; ++j) neg += coeff[v * NBINS + j] < 0;
        const double rel = rnorm[v] / sqrt(ny);
        if (rel > worst_fit) worst_fit = rel;
    }
    printf("NNLS: %d of %d spectra converged, %d negative coefficients, worst ||A x - y_ext|| / ||y|| = %.2e\n", okn, NVN, neg, worst_fit);
    pnx_nnls_plan_destroy(plan);
    if (okn != NVN || neg || worst_fit > 0.02) return 4;
    printf("C ABI demo ok\n");
    return 0;
}
