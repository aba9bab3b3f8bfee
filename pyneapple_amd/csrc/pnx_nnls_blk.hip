// pnx_nnls_blk.hip -- Lawson-Hanson NNLS for the reference's banded Tikhonov regularisers, second formulation (fp64).
//
// Hot path replaced: NNLSSolver._fit_single_pixel -> scipy.optimize.nnls(A, y_ext, maxiter) per voxel
// (reference src/pyneapple/solvers/nnls_solver.py:129-210) with A = [basis; mu * R] (nnls_solver.py:61-73) and R one of
// the banded Toeplitz matrices of model_functions/nnls.py:46-85.  Same active-set path as pnx_nnls.hip (selection rule,
// 0.01 independence test, alpha step, clean-up loop, `iteration == maxiter`), so iteration counts stay those of SciPy.
//
// Why a second kernel.  pnx_nnls.hip keeps every unit of a CU 55-70 % busy at once: the vector L1 (the dual
// w = A^T y - G[:,P] x streams p rows of G = 2 KB each per outer iteration: 87 k L1 accesses per voxel), the VALU and the
// scalar unit (one dependent LDS round trip and ~8 instructions per row of the inverse factor M, four sweeps over M per
// iteration).  This kernel takes both loads away:
//   * THE BASIS LIVES IN LDS, ONCE PER CU (32 x 258 doubles = 66 KB, shared by the 12 waves of the only workgroup a CU
//     holds).  The dual is evaluated in residual form, w = B^T (y - B_P x_P) - R^T (R x): p column gathers and 32 row
//     reads out of LDS plus two stencils -- no row of G is read at all (G is only gathered: p numbers per candidate
//     column), and the cancellation-free residual form is the more accurate one.  A^T y is never formed, so the MFMA
//     Gram step and its 2 KB per voxel round trip through HBM are not part of this path.
//   * M = L^-1 IS DISTRIBUTED IN 8 x 8 BLOCKS OVER THE WAVE.  Lane (a, b) = (lane >> 3, lane & 7) owns element
//     (8 I + a, 8 K + b) of block (I, K).  l = M g is one FMA per block and lane (21 for 48 rows) and a three-step DPP
//     butterfly per block row; the blocks are consumed row by row, each row's l going straight into the column sums of
//     l^T M, which a reduce-scatter (v_permlane32 / 16_swap, two block columns per swap) delivers in position order.
//     All block loads of a sweep are in flight together -- instead of p dependent row steps.  Rows < 32 of M live in
//     LDS, the others in a per-wave global slab (rows padded to multiples of 8: a block read is whole 64-byte lines,
//     a row access is contiguous); rows >= 48 and the Givens sweep of a removal go row by row with the loads of 4 - 12
//     rows in flight.  Position-indexed vectors hold 128 positions (two register slots): a voxel whose passive set wants
//     to grow beyond that (about one in 10^4 on the reference workload) is handed to pnx_nnls.hip through a list.
// Round 4 took a third of the vector instructions out (64 k -> 43 k per voxel, 6.65 -> 8.6 M voxels/s on the reference workload):
//   * a removal's rotation coefficients travel through LDS (one broadcast read per row instead of four v_readlane), q is rotated
//     in closed form (a prefix sum), row masks are scalar, shifts by one position are DPP moves;
//   * the dual gathers x by bin through LDS addresses kept per passive position, and its stencil passes ride on the round trips
//     of the gathers and of the B^T r product; passive flags are scalar wave masks;
//   * "LDS or slab" accesses are typed by address space (the compiler folds plain pointers into flat accesses that wait for
//     both counters);
//   * a rejected candidate column sets its passive flag and the dual is evaluated again -- no loop around the append, which had
//     cost every outer iteration 128 bytes of scratch per lane: the kernel has none left.
// The kernel arguments are read from the kernarg segment where they are used, the LDS scratch sits below 64 KB (DS offsets as
// immediates), phases take a fresh copy of the lane id: all three keep hoisted values out of registers the hot loops need.
// Round 5 (profiles/r05_nnls_experiments.md): the dual of a passive set of 1 .. 16 columns in Gram form (rows of G out of L2, A^T y
// = B^T y once per voxel), R^T R x as one five-point stencil for the tridiagonal regularisers -- and then the measurement that a
// SCALAR instruction costs what a vector instruction costs, both their share of all the instructions a wave issues (100 dummy
// s_add_u32 per outer iteration: + 6.6 %, 100 v_add_u32: + 6.4 %).  What bounds the kernel is the length of a wave's instruction
// stream, so the second half of the round took 16 % of ALL instructions out (86.2 k -> 72.1 k per voxel, 9.4 -> 10.7 M voxels/s):
//   * the candidate step is append_prepare<NI> (reads the voxel state: arg-max bin by s_ff1 / s_min_u32, l = M g, independence test,
//     l^T M) and ONE append_commit (lane writes, no EXEC masks, nothing to clear: x is zero behind the passive set);
//   * the outer iteration is TWO loops: its rare ways out leave the inner one with the state untouched, so the state is loop carried
//     along one path and updated in place (the structuriser had paid ~30 register moves per iteration for merging "unchanged" with
//     "appended"); the test hook that forces rejections is a kernel of its own (nnls_blk_hook_kernel);
//   * row offsets of a batch of rotation rows are worked out by the lanes (moff_batch), row masks are the previous mask shifted,
//     passive flags change by scalar conditional moves, wave-uniform reductions end in row broadcasts, stage_bins has no branch,
//     scalar loads are issued before LDS traffic is in flight (they share its counter).
// One wavefront owns one voxel; waves pull voxels from an atomic queue and never meet after the basis is staged.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pnx_internal.hpp"
#include "pnx_nnls.hpp"
#include "pnx_nnls_dev.hpp"

// ---- the two instantiations of the kernel body ---------------------------------------------------------------
// blk2: two position slots (passive sets up to 128 columns), twelve waves per CU at 168 registers -- every voxel of the reference
// workload but one in 10^4.  blk4 (last part of round 4): four slots (256 positions: every passive set of a 256-bin plan fits),
// eight waves per CU at 256 registers, (c, s) pairs of a removal for 256 positions in its LDS scratch.  It takes the voxels blk2
// hands over -- and whole calls when the pilot finds that blk2 would hand over most of them (strong regularisers) -- so that
// those run on the residual-form dual too (32 rows of B out of LDS per outer iteration) instead of the Gram form's p rows of G.
#define PNX_BLK_NS blk2
#define PNX_BLK_KERNEL nnls_blk_kernel
#define PNX_BLK_KERNEL_HOOK nnls_blk_hook_kernel
#define PNX_BLK_PS 2
#define PNX_BLK_ROWS2D 48
#ifndef PNX_BLK_WAVES
#define PNX_BLK_WAVES 12
#endif
#ifndef PNX_BLK_LDS_ROWS
#define PNX_BLK_LDS_ROWS 32
#endif
#include "pnx_nnls_blk_kernel.hpp"
#undef PNX_BLK_NS
#undef PNX_BLK_KERNEL
#undef PNX_BLK_PS
#undef PNX_BLK_ROWS2D
#undef PNX_BLK_WAVES
#undef PNX_BLK_LDS_ROWS
#define PNX_BLK_NS blk4
#define PNX_BLK_KERNEL nnls_blk4_kernel
#define PNX_BLK_PS 4
#define PNX_BLK_ROWS2D 64
#define PNX_BLK_WAVES 8
#define PNX_BLK_LDS_ROWS 32
#include "pnx_nnls_blk_kernel.hpp"
#undef PNX_BLK_NS
#undef PNX_BLK_KERNEL
#undef PNX_BLK_PS
#undef PNX_BLK_ROWS2D
#undef PNX_BLK_WAVES
#undef PNX_BLK_LDS_ROWS

namespace pnx {

#define PNX_HIPB(call)                                                                             \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) return set_error(PNX_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

bool nnls_blk_applicable(const NnlsPlanData *P) {
    return P->rhb != 0 && P->n_meas <= blk2::kBMeas && P->n_reg == P->n_bins && !dev_getenv("PNX_NNLS_NO_BLK");
}

// scratch of the block kernels: one workgroup per CU, Variant::mslab doubles of M per wave (zero initialised: the block sweeps
// read whole blocks, also rows no voxel of this wave has written yet).  slab == nullptr: the grid only (the four-slot kernel's
// slabs are the device's shared set, pnx_nnls.hpp)
template <class V> static int blk_variant_init(const NnlsPlanData *P, int *groups, double **slab) {
    PNX_HIPB(hipFuncSetAttribute(V::kernel(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)V::lds_bytes()));
    if (V::kernel_hook()) PNX_HIPB(hipFuncSetAttribute(V::kernel_hook(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)V::lds_bytes()));
    int occ = 0;
    PNX_HIPB(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, V::kernel(), V::waves * kW, V::lds_bytes()));
    if (occ < 1) return set_error(PNX_ERR_HIP, "nnls block kernel does not fit on a CU");
    *groups = occ * P->cus;
    if (!slab) return PNX_OK;
    const size_t bytes = (size_t)*groups * V::waves * V::mslab * sizeof(double);
    PNX_HIPB(hipMalloc(slab, bytes));
    PNX_HIPB(hipMemset(*slab, 0, bytes));
    return PNX_OK;
}
size_t nnls_blk4_slab_bytes(const NnlsPlanData *P) { return (size_t)P->blk4_groups * blk4::Variant::waves * blk4::Variant::mslab * sizeof(double); }
int nnls_blk_plan_init(NnlsPlanData *P) {
    int r;
    if ((r = blk_variant_init<blk2::Variant>(P, &P->blk_groups, &P->Mblk))) return r;
    if ((r = blk_variant_init<blk4::Variant>(P, &P->blk4_groups, nullptr))) return r;
    P->blk_bail_cap = (size_t)kAtyChunk;
    PNX_HIPB(hipMalloc(&P->blk_bail, (1 + P->blk_bail_cap) * sizeof(int32_t)));   // [0]: count, [1 ..]: voxel indices
    PNX_HIPB(hipMalloc(&P->blk4_bail, (1 + P->blk_bail_cap) * sizeof(int32_t)));  // the same for what blk4 hands to the Gram-form kernel
    PNX_HIPB(hipMalloc(&P->route, sizeof(int32_t)));
    PNX_HIPB(hipMemset(P->route, 0, sizeof(int32_t)));
    return PNX_OK;
}

// Copies the signal rows of the voxels handed over since the last call into the deferred list's dense side buffer (one
// block; a chunk hands over a few dozen voxels).  counters[0] = voxels handed over so far, counters[1] = rows copied so far.
__global__ void __launch_bounds__(256) bail_gather_kernel(int32_t *counters, const int32_t *bail, const double *y, long long base,
                                                          int n_meas, double *y_side, int cap) {
    const int n = counters[0] < cap ? counters[0] : cap, p = counters[1];
    for (long long e = (long long)p * n_meas + threadIdx.x; e < (long long)n * n_meas; e += blockDim.x) {
        const int i = (int)(e / n_meas), j = (int)(e % n_meas);
        y_side[e] = y[((long long)bail[i] - base) * n_meas + j];
    }
    __syncthreads();
    if (threadIdx.x == 0 && n > p) counters[1] = n;
}

// The pilot of a call.  blk2 pays for every voxel it hands over twice: ~130 outer iterations there, then the whole solve again
// in blk4.  At the reference's regularisation strength one voxel in 10^4 is handed over; with a stronger regulariser (order 1
// with mu = 0.5: half of them) the two-pass plan loses to one pass of the wider kernel (profiles/nnls_mu_probe.py).  So a call
// of at least 4 kBlkPilot voxels solves its first kBlkPilot voxels in blk2, and what that pilot hands over decides the route of
// the rest ON THE DEVICE: both kernels are launched over the remaining voxels, and the one that was not chosen leaves at once
// (one scalar load per wave).  The choice depends on the pilot's voxels only, never on timing: the same call takes the same
// route every time.
constexpr int64_t kBlkPilot = 12288;  // four voxels per resident wave of a full grid
__global__ void blk_route_kernel(const int32_t *n_bail, int pilot, int permille, int32_t *route) {
    route[0] = ((long long)n_bail[0] * 1000 > (long long)permille * pilot) ? 1 : 0;
}

struct BlkCall {  // what every launch of a call shares
    NnlsPlanData *P;
    const double *y;
    double *coeff, *rnorm;
    int8_t *status;
    int32_t *iters;
    int max_iter;
    hipStream_t stream;
};
// one launch of instantiation V over the voxels [off, off + c) of the call -- or, with a list, over list[0 .. min(*count, c))
template <class V>
static int blk_launch(const BlkCall &C, int64_t off, int64_t c, long long vox_base, int32_t *n_bail, int32_t *bail, const int32_t *route,
                      int route_want, const int32_t *list, const int32_t *count) {
    NnlsPlanData *P = C.P;
    typename V::Args a;
    a.y = C.y + (size_t)off * P->n_meas;
    a.coeff = C.coeff + (size_t)off * P->n_bins;
    a.rnorm = C.rnorm + off;
    a.status = C.status ? C.status + off : nullptr;
    a.iters = C.iters ? C.iters + off : nullptr;
    a.G = P->G;
    a.Bp = P->Bp;
    a.Mglob = V::max_pos == 128 ? P->Mblk : P->Mblk4;
    a.n_bail = n_bail;
    a.bail = bail;
    a.vox_base = vox_base;
    a.queue = P->queue;
    a.n_vox = c;
    a.n_meas = P->n_meas;
    a.n_bins = P->n_bins;
    a.n_reg = P->n_reg;
    a.max_iter = C.max_iter;
    for (int k = 0; k < 5; ++k) a.rc[k] = P->rc[k];
    {  // R^T R of a tridiagonal Toeplitz R (rows a_-1, a_0, a_1 = rc[1 .. 3]) as one stencil, and the two end corrections (see dual_residual_form)
        const double am = P->rc[1], a0 = P->rc[2], ap = P->rc[3];
        a.rg[0] = am * am + a0 * a0 + ap * ap;
        a.rg[1] = am * a0 + a0 * ap;
        a.rg[2] = am * ap;
        a.rg[3] = ap * ap;  // row -1 would have added a_1^2 x_0 to bin 0
        a.rg[4] = am * am;  // row n would have added a_-1^2 x_{n-1} to bin n - 1
    }
    a.rhb = P->rhb;
    a.test_rej_k = a.test_rej_n = 0;
    if (const char *t = V::max_pos == 128 ? dev_getenv("PNX_NNLS_TEST_REJECT") : nullptr) {  // the hand-over target runs without the hook (as the oracle's restatement of the hand-over does)
        if (sscanf(t, "%d,%d", &a.test_rej_k, &a.test_rej_n) != 2 || a.test_rej_k < 1 || a.test_rej_n < 1) a.test_rej_k = a.test_rej_n = 0;
    }
    a.route = route;
    a.route_want = route_want;
    a.redo_list = list;
    a.redo_count = count;
    NnlsSharedUse use(V::max_pos == 128 ? nullptr : P, C.stream);  // the four-slot kernel's slabs are the device's shared set
    PNX_HIPB(hipMemsetAsync(P->queue, 0, sizeof(unsigned long long), C.stream));
    const int groups = V::max_pos == 128 ? P->blk_groups : P->blk4_groups;
    long long grid = (c + V::waves - 1) / V::waves;
    if (grid > groups) grid = groups;
    V::launch(dim3((unsigned)grid), V::lds_bytes(), C.stream, a);
    PNX_HIPB(hipGetLastError());
    return PNX_OK;
}

// The voxels list[0 .. min(*count, n_vox)) of a call through blk4 (every passive set of a 256-bin plan fits its 256 positions);
// what blk4 itself gives up (a ninth rejected candidate in one outer iteration: test hook only) goes to the Gram-form kernel.
int nnls_blk_redo_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                         int8_t *status_d, int32_t *iters_d, const int32_t *list, const int32_t *count, hipStream_t stream) {
    if (dev_getenv("PNX_BLK_NO_WIDE"))  // (A/B) the round-4 hand-over target
        return nnls_redo_device(P, n_vox, y_d, max_iter, coeff_d, rnorm_d, status_d, iters_d, list, count, stream);
    const BlkCall C{P, y_d, coeff_d, rnorm_d, status_d, iters_d, max_iter, stream};
    PNX_HIPB(hipMemsetAsync(P->blk4_bail, 0, sizeof(int32_t), stream));
    int r = blk_launch<blk4::Variant>(C, 0, n_vox, 0, P->blk4_bail, P->blk4_bail + 1, nullptr, 0, list, count);
    if (r) return r;
    return nnls_redo_device(P, n_vox, y_d, max_iter, coeff_d, rnorm_d, status_d, iters_d, P->blk4_bail + 1, P->blk4_bail, stream);
}

int nnls_blk_solve_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                          int8_t *status_d, int32_t *iters_d, hipStream_t stream, const NnlsDefer *defer) {
    if (n_vox <= 0) return PNX_OK;
    if (n_vox >= (int64_t)1 << 31) return set_error(PNX_ERR_INVALID, "n_vox=%lld: at most 2^31 - 1 voxels per call", (long long)n_vox);
    if ((size_t)n_vox > P->blk_bail_cap) {  // the hand-over lists hold every voxel of a call in the worst case
        (void)hipFree(P->blk_bail);  // synchronises: no earlier solve of this plan is still using them
        (void)hipFree(P->blk4_bail);
        P->blk_bail = P->blk4_bail = nullptr;
        P->blk_bail_cap = (size_t)n_vox;
        PNX_HIPB(hipMalloc(&P->blk_bail, (1 + P->blk_bail_cap) * sizeof(int32_t)));
        PNX_HIPB(hipMalloc(&P->blk4_bail, (1 + P->blk_bail_cap) * sizeof(int32_t)));
    }
    if (!defer) PNX_HIPB(hipMemsetAsync(P->blk_bail, 0, sizeof(int32_t), stream));
    const BlkCall C{P, y_d, coeff_d, rnorm_d, status_d, iters_d, max_iter, stream};
    int32_t *n_bail = defer ? defer->counters : P->blk_bail, *bail = defer ? defer->bail : P->blk_bail + 1;
    const long long base = defer ? defer->base : 0;
    // ONE launch for the rest of the call: the kernel needs no per-chunk buffer (A^T y is never formed), and every launch ends in
    // a drain tail of ~1.4 ms (C4 volume: 488.2 ms in four launches of 2^20 voxels, 484.1 ms in one; PNX_BLK_CHUNK_LOG2 = 20
    // brings the launches of kAtyChunk voxels back)
    int64_t chunk = n_vox;
    if (const char *t = dev_getenv("PNX_BLK_CHUNK_LOG2")) {
        const int l2 = atoi(t);
        if (l2 >= 10 && l2 <= 30) chunk = (int64_t)1 << l2;
    }
    // the pilot (see blk_route_kernel): in the first chunk of a call; the later chunks of a host-array call follow its route
    // (they run on the same stream, behind it)
    int permille = 150;  // the two-pass plan (blk2, then blk4 for what it hands over) against one pass of blk4, by share handed over: equal at ~14 % (profiles/nnls_mu_probe.py)
    if (const char *t = dev_getenv("PNX_BLK_ROUTE_PERMILLE")) permille = atoi(t);  // <= 0: no pilot, blk2 first for everything
    const bool wide_route = !dev_getenv("PNX_BLK_NO_WIDE");
    const bool first = !defer || defer->base == 0;
    const int64_t pilot = (permille > 0 && first && n_vox >= 4 * kBlkPilot) ? kBlkPilot : 0;
    const int32_t *route = nullptr;
    if (permille > 0 && defer && first && !pilot) PNX_HIPB(hipMemsetAsync(P->route, 0, sizeof(int32_t), stream));  // a short first chunk: blk2
    if (permille > 0 && (pilot || (defer && !first))) route = P->route;
    int r;
    if (pilot) {
        if ((r = blk_launch<blk2::Variant>(C, 0, pilot, base, n_bail, bail, nullptr, 0, nullptr, nullptr))) return r;
        hipLaunchKernelGGL(blk_route_kernel, dim3(1), dim3(1), 0, stream, n_bail, (int)pilot, permille, P->route);
        PNX_HIPB(hipGetLastError());
        if (dev_getenv("PNX_BLK_ROUTE_DEBUG")) {  // diagnostic (synchronises): what the pilot saw
            int32_t cnt = 0, rt = 0;
            PNX_HIPB(hipStreamSynchronize(stream));
            PNX_HIPB(hipMemcpy(&cnt, n_bail, sizeof(cnt), hipMemcpyDeviceToHost));
            PNX_HIPB(hipMemcpy(&rt, P->route, sizeof(rt), hipMemcpyDeviceToHost));
            fprintf(stderr, "pnx nnls pilot: %d of %d voxels handed over (%.1f %%), threshold %.1f %% -> %s\n", cnt, (int)pilot, 100.0 * cnt / (double)pilot,
                    permille / 10.0, rt ? (wide_route ? "four-slot block kernel" : "Gram-form kernel") : "block kernel");
        }
    }
    for (int64_t off = pilot; off < n_vox; off += chunk) {
        const int64_t c = (n_vox - off) < chunk ? (n_vox - off) : chunk;
        if ((r = blk_launch<blk2::Variant>(C, off, c, base + off, n_bail, bail, route, 0, nullptr, nullptr))) return r;
    }
    if (route) {  // the same voxels through the other kernel -- which leaves at once unless the pilot chose it
        if (wide_route) {
            PNX_HIPB(hipMemsetAsync(P->blk4_bail, 0, sizeof(int32_t), stream));
            if ((r = blk_launch<blk4::Variant>(C, pilot, n_vox - pilot, pilot, P->blk4_bail, P->blk4_bail + 1, route, 1, nullptr, nullptr))) return r;
            // what blk4 gives up (test hook only): the Gram-form kernel, indices relative to this call's arrays
            if ((r = nnls_redo_device(P, n_vox, y_d, max_iter, coeff_d, rnorm_d, status_d, iters_d, P->blk4_bail + 1, P->blk4_bail, stream))) return r;
        } else if ((r = nnls_routed_device(P, n_vox - pilot, y_d + (size_t)pilot * P->n_meas, max_iter, coeff_d + (size_t)pilot * P->n_bins, rnorm_d + pilot,
                                           status_d ? status_d + pilot : nullptr, iters_d ? iters_d + pilot : nullptr, route, stream)))
            return r;
    }
    // voxels whose passive set outgrew blk2 (about one in 10^4 on the reference workload, and the slowest ones: a single
    // launch for the whole call, so that their long solves overlap): blk4, from scratch
    if (defer) {
        // a host-array call made of several chunks: this chunk's handed-over voxels keep their signal rows in the side buffer,
        // and ONE pass at the end of the call solves them all (each such pass costs ~8 ms whatever the number of voxels:
        // they are the longest solves there are)
        hipLaunchKernelGGL(bail_gather_kernel, dim3(1), dim3(256), 0, stream, defer->counters, defer->bail, y_d, (long long)defer->base,
                           P->n_meas, defer->y_side, defer->cap);
        PNX_HIPB(hipGetLastError());
        return PNX_OK;
    }
    return nnls_blk_redo_device(P, n_vox, y_d, max_iter, coeff_d, rnorm_d, status_d, iters_d, P->blk_bail + 1, P->blk_bail, stream);
}

}  // namespace pnx
