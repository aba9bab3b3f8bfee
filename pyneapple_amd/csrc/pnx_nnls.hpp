// pnx_nnls.hpp -- internal interface of the batched NNLS solver (see pnx_nnls.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pnx {
constexpr int64_t kAtyChunk = 1 << 20;  // voxels per MFMA Gram step / active-set launch pair


constexpr int kNnlsMaxBins = 256;   // 4 bins per lane: every fast path (block kernel, MFMA Gram step, 16 waves per CU)
constexpr int kNnlsWideBins = 512;  // 8 bins per lane: the wide instantiations of the Gram- and QR-form kernels (257 .. 512 bins)
constexpr int kNnlsMaxMeas = 128;

struct NnlsPlanData {
    int device = 0;
    int cus = 0;
    int n_meas = 0, n_bins = 0, n_reg = 0;
    double *B = nullptr;      // (n_meas, n_bins) row-major: basis
    double *Bp = nullptr;     // (n_meas, bstride) zero padded copy (16-byte aligned rows for the kernel)
    int bstride = kNnlsMaxBins;  // row stride of Bp and G: 256, or 512 for a wide plan (n_bins > 256)
    double *RT = nullptr;     // (n_bins, n_reg)  row-major: reg transposed (column j of reg contiguous)
    double *G = nullptr;      // (n_bins + 1, bstride): A^T A = B^T B + reg^T reg, fp64, zero padded
    double *aty = nullptr;    // (chunk, 256) A^T y of the current chunk (MFMA Gram step), null = VALU path
    double *Mglob = nullptr;  // per-wave overflow rows (>= 64) of the inverse Cholesky factor
    size_t mglob_stride = 0;  // doubles per wave
    int n_waves = 0;          // persistent waves the scratch was sized for
    double rc[5] = {0, 0, 0, 0, 0};  // banded Toeplitz regulariser (orders 1-3 of the reference): reg[i][j] = rc[j - i + 2]
    int rhb = 0;              // its half bandwidth, 0 = general regulariser
    bool mfma_ok = false;     // the MFMA Gram step can run for this plan (n_meas <= 64)
    bool qr = false;          // no (or an all-zero) regulariser: QR-based kernels (pnx_nnls_qr.hip; Q and R in LDS up to 64 measurements, in a global slab beyond)
    double *qr_slab = nullptr;  // Q / R of the 65 .. 128 measurement kernel, per resident wave
    int qr_slab_groups = 0;
    bool blk = false;         // banded Toeplitz regulariser and <= 32 measurements: LDS-resident basis, block-distributed factor (pnx_nnls_blk.hip)
    double *Mblk = nullptr;   // its per-wave slabs of the inverse Cholesky factor
    double *Mblk4 = nullptr;  // the same for the four-slot instantiation (256 positions, eight waves per CU: 270 KB per wave)
    int blk4_groups = 0;
    int32_t *blk4_bail = nullptr;  // what the four-slot instantiation hands to the Gram-form kernel (a ninth rejected candidate: test hook only)
    double *Mwide = nullptr;  // wide plans: slabs of the hand-over pass nnls_kernel<8, 8> (rows up to 511), one wave per CU
    int wide_waves = 0;
    int blk_groups = 0;       // its persistent workgroups (16 waves each)
    int32_t *blk_bail = nullptr;  // [0]: number of voxels the block kernel handed to the general one (more than 128 passive bins), [1 ..]: their indices; a wide plan's nnls_kernel<8, 4> -> <8, 8> list (more than 256)
    size_t blk_bail_cap = 0;      // voxels the list can hold
    int32_t *route = nullptr;     // block-kernel plans: [0] the route the pilot of the current call chose (0: block kernel, 1: Gram-form kernel), device
    unsigned long long *queue = nullptr;
    bool shared_slabs = false;    // block-kernel plans: Mblk4 and Mglob belong to the device's shared slab set (below), not to the plan
};

// The slabs of the two kernels BEHIND a block-kernel plan's first pass -- the four-slot block kernel (0.55 GB) and the Gram-form
// kernel that remains its last resort (1 GB) -- exist once per device, for every such plan on it: they are only used by the
// hand-over pass (one voxel in 10^4 on the reference workload) and by calls the pilot routes to the four-slot kernel, so N plans
// cost 1.55 GB + N x 0.2 GB (the two-slot kernel's own slabs), not N x 1.8 GB.  Uses are ordered on the device: a launch that
// touches them waits for the event recorded behind the previous one (another plan, another stream), enqueue order = host order
// under the set's mutex; everything stays asynchronous.  The set stays allocated while idle (a plan per call -- api.nnls,
// pnx_nnls_batch_f64 -- would otherwise pay 1.55 GB of hipMalloc + memset each time); pnx_release_staging(device) frees it.
int nnls_shared_slabs_get(int device, size_t blk4_bytes, size_t gram_bytes, double **blk4, double **gram);
void nnls_shared_slabs_put(int device);
int nnls_shared_slabs_trim(int device);  // frees the set unless a plan holds it (returns 0 then, 1 otherwise)
struct NnlsSharedUse {  // brackets the enqueue of launches that use the shared slabs on `stream`
    NnlsSharedUse(const NnlsPlanData *P, hipStream_t stream);
    ~NnlsSharedUse();
    const NnlsPlanData *P;
    hipStream_t stream;
};

int nnls_plan_init(NnlsPlanData *P, int n_meas, int n_bins, const double *basis, const double *reg, int n_reg,
                   int device, int cus);
void nnls_plan_free(NnlsPlanData *P);
int nnls_solve_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d,
                      double *rnorm_d, int8_t *status_d, int32_t *iters_d, hipStream_t stream);
int nnls_qr_solve_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                         int8_t *status_d, int32_t *iters_d, hipStream_t stream);
// the general kernel on the voxels list[0 .. *count) of a chunk (A^T y on the VALU)
int nnls_redo_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                     int8_t *status_d, int32_t *iters_d, const int32_t *list, const int32_t *count, hipStream_t stream);
// the general kernel on ALL n_vox voxels (A^T y on the VALU) -- unless *route != 1 when it starts: then every wave leaves at once
int nnls_routed_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                       int8_t *status_d, int32_t *iters_d, const int32_t *route, hipStream_t stream);
bool nnls_blk_applicable(const NnlsPlanData *P);
// block-kernel plans: the voxels list[0 .. min(*count, n_vox)) through the four-slot block kernel (hand-over target of the two-slot one)
int nnls_blk_redo_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                         int8_t *status_d, int32_t *iters_d, const int32_t *list, const int32_t *count, hipStream_t stream);
int nnls_blk_plan_init(NnlsPlanData *P);
size_t nnls_blk4_slab_bytes(const NnlsPlanData *P);  // of the four-slot kernel's grid on this plan's device (after nnls_blk_plan_init)
// Deferred hand-over (host-array calls made of several chunks, pnx_api.hip): the block kernel appends the voxels it hands over
// to `bail` with their index within the whole call (`base` + index within this chunk), a gather keeps their signal rows in
// `y_side`, and the caller solves them in one pass at the end instead of one pass per chunk.
struct NnlsDefer {
    int32_t *counters;  // device: [0] voxels handed over so far, [1] signal rows gathered so far
    int32_t *bail;      // device: indices within the call, room for every voxel of the call
    double *y_side;     // device: (cap, n_meas) signal rows of the first `cap` handed-over voxels
    int cap;
    int64_t base;       // first voxel of this chunk within the call
};
int nnls_blk_solve_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, int max_iter, double *coeff_d, double *rnorm_d,
                          int8_t *status_d, int32_t *iters_d, hipStream_t stream, const NnlsDefer *defer = nullptr);
int nnls_aty_device(NnlsPlanData *P, int64_t n_vox, const double *y_d, double *aty_d, hipStream_t stream);
int nnls_build_basis(int n_meas, const double *b, int n_bins, const double *bins, double *basis);

}  // namespace pnx
