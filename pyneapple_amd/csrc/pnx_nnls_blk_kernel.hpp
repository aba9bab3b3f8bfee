// pnx_nnls_blk_kernel.hpp -- body of the NNLS block kernel (see pnx_nnls_blk.hip, which includes this file once per instantiation).
// The includer defines: PNX_BLK_NS (namespace of the instantiation), PNX_BLK_KERNEL (kernel name), PNX_BLK_PS (register slots of a
// position-indexed vector: 2 or 4), PNX_BLK_WAVES (waves per workgroup = per CU), PNX_BLK_LDS_ROWS (rows of M kept in LDS).
namespace pnx {
namespace PNX_BLK_NS {

constexpr int kBMeas = 32;                  // measurements the LDS copy of the basis holds
constexpr int kBStride = kNnlsMaxBins + 2;  // even: rows stay 16-byte aligned for ds_read_b128; a column gather (lane = measurement) is 2-way bank conflicted
constexpr int kBlkWaves = PNX_BLK_WAVES;    // waves per workgroup = voxels in flight per CU (12: 168 registers per wave, no scratch; 16 waves at 128
                                            // registers measured 7.2 against 8.2 M voxels/s; the four-slot instantiation: 8 waves, 256 registers)
constexpr int kRows2D = PNX_BLK_ROWS2D;     // rows / columns of M handled block-wise: 48 (6 x 6 blocks of 8 x 8; two-slot instantiation: eight block rows
                                            // cost it 160 bytes of scratch, measured) or 64 (four-slot instantiation: 256 registers per lane)
constexpr int kNIMax = kRows2D / 8;
static_assert(kRows2D == 48 || kRows2D == 64, "block-wise part of M: six or eight block rows");
constexpr int kPS = PNX_BLK_PS;             // register slots of a position-indexed vector: 2 (positions < 128) or 4 (< 256: every passive set fits)
constexpr int kMaxPos = kPS * kW;           // a voxel whose passive set wants to grow beyond that is handed to pnx_nnls.hip
constexpr int kXbuf = (2 + kNnlsMaxBins + 2 + 4) > 2 * kMaxPos + 8 ? (2 + kNnlsMaxBins + 2 + 4) : 2 * kMaxPos + 8;  // x by bin with its halo and four spare doubles -- or the (c, s) pairs of a removal, one per position
// per-wave LDS scratch, in doubles: ps[128] (ints): per passive position the LDS byte address of its bin's entry in xbuf |
// xbuf[264]: x by bin with a halo of two (then t = R x by bin; the (c, s) pairs of a removal; the staging vector of M^T q; its
// last four doubles: the list of rejected columns) | rb[32]: residual of the measurements.  Every LDS byte not spent here keeps
// a row of M on the chip
constexpr int kPadBin = kNnlsMaxBins;  // bin of the padding positions: column 256 of the LDS basis and xbuf[2 + 256] are zero
constexpr int kScr = kMaxPos / 2 + kXbuf + 32;
constexpr int kLdsM = PNX_BLK_LDS_ROWS;     // rows of M that live in LDS (whole block rows): what every voxel uses all the time
constexpr int kLdsMDoubles = (kLdsM / 8 + 1) * (32 * (kLdsM / 8));  // moff(kLdsM)
constexpr int kMSlab = 32 * (kMaxPos / 8) * (kMaxPos / 8 + 1) + 64;   // doubles of M per wave: moff(kMaxPos) = 32 I (I + 1) at I = kMaxPos / 8, plus the overrun of a 64-lane row read
typedef int __attribute__((may_alias)) lds_int;

struct BlkArgs {
    const double *y;
    double *coeff;
    double *rnorm;
    int8_t *status;
    int32_t *iters;
    const double *G;   // (n_bins, 256) zero padded
    const double *Bp;  // (n_meas, 256) zero padded
    double *Mglob;     // kMSlab doubles per wave, zero initialised (so that every block a sweep touches is finite)
    int32_t *n_bail;   // number of voxels handed over to the general kernel ...
    int32_t *bail;     // ... and their indices (within the call: vox_base + index within the chunk)
    long long vox_base;  // first voxel of this launch within the call
    unsigned long long *queue;
    long long n_vox;
    int n_meas, n_bins, n_reg, max_iter;
    double rc[5];  // banded Toeplitz regulariser: R[i][j] = rc[j - i + 2] (mu included)
    double rg[5];  // half bandwidth 1 only: R^T R as ONE stencil -- (R^T R)[j][j +- e] = rg[e] (e = 0, 1, 2) away from the ends; rg[3] = rc[3]^2 and
                   // rg[4] = rc[1]^2 are what the rows R does not have (-1 and n) would have added to bins 0 and n - 1
    int rhb;       // half bandwidth, 1 or 2
    int test_rej_k, test_rej_n;  // test hook (PNX_NNLS_TEST_REJECT=k,n): with p % k == k - 1 the first n candidates of an outer iteration are rejected unseen
    const int32_t *route;  // non-null: the launch only runs while *route == route_want (the pilot of the call chose this kernel)
    int route_want;
    const int32_t *redo_list, *redo_count;  // non-null: only the voxels redo_list[0 .. min(*redo_count, n_vox)) (handed over by the two-slot instantiation)
};

// -DPNX_NNLS_BLK_CHECK: every index into the slab of M / a row of G is range checked; the first violation is reported with
// printf and the access is redirected to element 0 (diagnostic builds only)
#ifdef PNX_NNLS_BLK_CHECK
__device__ int g_blk_err = 0;
__device__ __noinline__ int ck_(int idx, int lim, int code, int aux) {
    if (idx < 0 || idx >= lim) {
        if (atomicAdd(&g_blk_err, 1) == 0) printf("BLK_CHECK code=%d idx=%d lim=%d aux=%d block=%d thread=%d\n", code, idx, lim, aux, (int)blockIdx.x, (int)threadIdx.x);
        return 0;
    }
    return idx;
}
#define CK(idx, lim, code, aux) ck_((idx), (lim), (code), (aux))
#else
#define CK(idx, lim, code, aux) (idx)
#endif

// Row i of M starts at moff(i): rows are padded to a multiple of 8 entries, so block row I (rows 8 I .. 8 I + 7) has
// I + 1 blocks of 8 x 8 and each lane row of a block is one 64-byte line.
__host__ __device__ constexpr int moff(int i) {
    const int I = i >> 3, a = i & 7;
    return (I + 1) * (32 * I + 8 * a);
}

static_assert(kMSlab >= moff(kMaxPos) + kW, "slab of M too small for kMaxPos rows");
static_assert(kLdsMDoubles == moff(kLdsM) && kLdsM % 8 == 0, "LDS part of M: whole block rows");

// M of one wave: rows < kLdsM in LDS, the others in the wave's global slab (same offsets moff(i) + k in both).  A row index
// that is wave uniform picks its memory with a scalar branch.
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(1))) double glb_double;
struct MRef {
    // typed by address space: a branch "LDS or slab" then cannot be folded into one flat access through a selected base
    // pointer (a flat access waits for both the LDS and the vector-memory counter)
    lds_double *l;  // LDS, kLdsMDoubles
    glb_double *g;  // global slab, kMSlab
    // The empty asm keeps the two branches apart: without it the compiler folds them into ONE flat access through a selected
    // base pointer, and a flat access waits for both the LDS and the vector-memory counter.
    __device__ __forceinline__ double ld(int row, int idx) const {
        double v;
        if (row < kLdsM) {
            asm volatile("" ::: "memory");
            v = l[idx];
        } else
            v = g[CK(idx, kMSlab, 20, row)];
        return v;
    }
    __device__ __forceinline__ void st(int row, int idx, double v) const {
        if (row < kLdsM) {
            asm volatile("" ::: "memory");
            l[idx] = v;
        } else
            g[CK(idx, kMSlab, 21, row)] = v;
    }
};

// ---- reductions over one axis of the 8 x 8 lane grid (every lane of the group gets the sum) ----------------
// a DPP lane permutation (every lane has a source): no `old` operand, so no register copy in front of the move
template <int CTRL> __device__ __forceinline__ double dppx(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap_add16(double v) {  // + the lane 16 away (rows of 16 lanes swapped pairwise)
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double swap_add32(double v) {  // + the lane 32 away
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// a value every lane holds, moved to scalar registers: the branches that depend on it become scalar branches (a per-lane
// condition around DPP / readlane / permlane code makes the compiler mask a loop that no lane ever leaves alone)
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// v_max_f64 as it is: fmax() puts a canonicalising v_max_f64 x, x, x in front of every operand (IEEE mode), three
// instructions where one does -- the dual values are ordinary numbers or -inf, never signalling NaNs
__device__ __forceinline__ double max1(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// maximum over the wave, in every lane
__device__ __forceinline__ double allreduce_max(double v) {
    v = max1(v, dppx<0xB1>(v));
    v = max1(v, dppx<0x4E>(v));
    v = max1(v, dppx<0x141>(v));
    v = max1(v, dppx<0x128>(v));
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = max1(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    }
    {
        const int lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = max1(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
    }
    return v;
}
// Largest POSITIVE value of the wave as a wave-uniform number (0 if there is none): where only that is wanted, the last two steps are
// row broadcasts (row_bcast:15 / row_bcast:31, three instructions each) instead of the 16- and 32-lane swaps (five each), and the total is
// read from lane 63.  A broadcast has no source for row 0 (rows 0 and 1): those lanes receive zero, which a maximum that only counts
// when it is positive does not see.
__device__ __forceinline__ double wave_max_pos_uni(double v) {
    v = max1(v, dppx<0xB1>(v));
    v = max1(v, dppx<0x4E>(v));
    v = max1(v, dppx<0x141>(v));
    v = max1(v, dppx<0x140>(v));  // row_mirror: every lane of a row of 16 holds the row's maximum
    v = max1(v, dppx<0x142>(v));  // row_bcast:15: row r receives lane 15 of row r - 1
    v = max1(v, dppx<0x143>(v));  // row_bcast:31: rows 2 and 3 receive lane 31
    return rl(v, 63);
}
// sum over a = lane >> 3 of a value that does not depend on b = lane & 7, as a wave-uniform value: row_ror 8, then the two row
// broadcasts (zero where there is no source); lane 63 ends with the same tree of additions as allreduce_a -- the same bits
__device__ __forceinline__ double sum_a_uni(double v) {
    v += dppx<0x128>(v);
    v += dppx<0x142>(v);
    v += dppx<0x143>(v);
    return rl(v, 63);
}
// over b = lane & 7: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
__device__ __forceinline__ double allreduce_b(double v) {
    v += dppx<0xB1>(v);
    v += dppx<0x4E>(v);
    v += dppx<0x141>(v);
    return v;
}
// over a = lane >> 3: row_ror 8, then the two swaps
__device__ __forceinline__ double allreduce_a(double v) {
    v += dppx<0x128>(v);
    v = swap_add16(v);
    v = swap_add32(v);
    return v;
}

// Reduce-scatter over a: v[K] is lane (a, b)'s partial sum of column block K; returned is, in the lanes of lane row a, the
// total of block K = a -- which is position order (position 8 a + b lives in lane 8 a + b).  A swap of two registers
// moves two blocks per step where an all-reduce would move one: 25 instructions for eight blocks instead of 15 per block.
__device__ __forceinline__ double rs32(double x, double y) {  // lanes < 32: x + x(lane + 32); lanes >= 32: y + y(lane - 32)
    const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double rs16(double x, double y) {  // even rows of 16 lanes: x + x(next row); odd rows: y + y(previous row)
    const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
template <int NI> __device__ __forceinline__ double reduce_scatter_a(const double (&v)[NI], int la) {
    static_assert(NI <= 8, "eight lane rows");
    double R[4], Q[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) R[j] = j < NI ? rs32(v[j], j + 4 < NI ? v[j + 4 < NI ? j + 4 : 0] : 0.0) : 0.0;
#pragma unroll
    for (int j = 0; j < 2; ++j) Q[j] = rs16(R[j], R[j + 2]);
    const bool odd = la & 1;
    const double keep = odd ? Q[1] : Q[0], send = odd ? Q[0] : Q[1];
    return keep + dppx<0x128>(send);
}

// moff(i) .. moff(i + N - 1) of N <= 64 consecutive rows: lane r works out moff(i + r) -- six vector instructions and a v_readlane per row
// instead of seven scalar instructions per row
template <int N> __device__ __forceinline__ void moff_batch(int i, int lane, int (&mo)[N]) {
    const int k = i + lane, I = k >> 3;
    const int v = (int)(__umul24((unsigned)(I + 1), (unsigned)(k - 4 * I)) << 3);  // (I + 1) (32 I + 8 a), a = k - 8 I
#pragma unroll
    for (int r = 0; r < N; ++r) mo[r] = __builtin_amdgcn_readlane(v, r);
}
// lanes <= k as a wave mask in scalar registers (k wave uniform): row masks cost no VALU compare
__device__ __forceinline__ unsigned long long lanes_le(int k) {
    return k >= 63 ? ~0ull : (k < 0 ? 0ull : ((2ull << k) - 1ull));
}
// the same where k >= 0 is known (a row or a passive set that exists): three scalar instructions instead of six
__device__ __forceinline__ unsigned long long lanes_le_nn(int k) { return ~0ull >> (63 - (k < 63 ? k : 63)); }
// position i receives the value of position i + 1, on DPP (wave_shl:1: lane l reads lane l + 1; lane 63 keeps `old` = lane 0 of
// the next slot): two v_mov_dpp per double instead of two ds_bpermute round trips
__device__ __forceinline__ double wshl1(double v, double last) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(last), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(last), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int NS> __device__ __forceinline__ void shift_down_dpp(const double (&a)[NS], double (&out)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) out[s] = wshl1(a[s], s + 1 < NS ? rl(a[s + 1 < NS ? s + 1 : s], 0) : 0.0);
}
template <int NS> __device__ __forceinline__ void shift_down_dpp_i(const int (&a)[NS], int (&out)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
        out[s] = __builtin_amdgcn_update_dpp(s + 1 < NS ? __builtin_amdgcn_readlane(a[s + 1 < NS ? s + 1 : s], 0) : 0, a[s], 0x130, 0xf, 0xf, false);
}

// ---- products with the LDS-resident basis ------------------------------------------------------------------
#ifdef PNX_NNLS_STAMP
__device__ unsigned long long g_blk_rejects = 0;  // rejected candidate columns since the library was loaded (diagnostic builds)
#define STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[k] += t_ - tlast; tlast = t_; } while (0)
#define COUNT(k, v) do { cnt[k] += (v); } while (0)
#else
#define STAMP(k) do {} while (0)
#define COUNT(k, v) do {} while (0)
#endif

// The queue pull sits in a function of its own: inlined, its one-lane branch is merged by the compiler into the control flow
// of the voxel loop, which then runs under EXEC masks it derives per lane -- around DPP / readlane / permlane code.
__device__ __noinline__ unsigned long long next_voxel(unsigned long long *queue, int lane) {
    unsigned long long vq = 0;
    if (lane == 0) vq = atomicAdd(queue, 1ULL);
    return vq;
}

// A copy of the lane id the optimiser cannot see through: what a phase derives from it (LDS addresses, block offsets, lane
// masks) is computed where the phase starts instead of once per kernel -- hoisted out of the voxel loop those ~40 values
// do not fit into 128 registers and come back from scratch memory in every inner loop.
__device__ __forceinline__ int fresh(int lane) {
    asm volatile("" : "+v"(lane));
    return lane;
}

// The kernel arguments are read from the kernarg segment where they are needed (scalar loads), through a pointer the
// optimiser cannot see through: kept in scalar registers for the whole kernel, the 9 pointers and 5 regulariser
// coefficients push ~100 SGPRs into VGPR lanes, and every use costs a v_readlane -- a VALU instruction in a VALU-bound kernel.
typedef const BlkArgs __attribute__((address_space(4))) KArgs;
__device__ __forceinline__ KArgs *kargs() {
    KArgs *p = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// owner of a bin: binof(lane, slot) = 128 (slot >> 1) + 2 lane + (slot & 1)
__device__ __forceinline__ int slot_of_bin(int j) { return ((j >> 7) << 1) | (j & 1); }
__device__ __forceinline__ int lane_of_bin(int j) { return (j >> 1) & 63; }
// per-voxel state that the phases below share
struct VoxState {
    double q[kPS], x[kPS], z[kPS];  // by position
    int pidx[kPS];                  // bin of a position
    unsigned long long inP[kSlots];  // by bin, as wave masks in scalar registers (bit l of inP[s]: bin binof(l, s)); bins >= n_bins count as taken
    int nrej;                        // columns rejected in this outer iteration (their inP bits are set meanwhile; the bins wait in LDS)
    int p;
};

// bin j joins / leaves the passive flags: ONE bit of ONE of the four masks changes (mask 2 (j >> 7) + (j & 1), bit (j >> 1) & 63).  Written
// out as scalar instructions: per mask an OR (AND-NOT), a compare and a conditional move -- 15 instructions; the compiler's selects
// cost 22, and scalar branches to the one mask are rebuilt by the structuriser into four flagged blocks (measured: no better)
__device__ __forceinline__ void set_passive(VoxState &S, int j) {
    unsigned long long t, n;
    int k;
    asm("s_lshr_b32 %[k], %[j], 1\n\t"
        "s_lshl_b64 %[t], 1, %[k]\n\t"
        "s_and_b32 %[k], %[j], 0x81\n\t"
        "s_or_b64 %[n], %[m0], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 0\n\t"
        "s_cmov_b64 %[m0], %[n]\n\t"
        "s_or_b64 %[n], %[m1], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 1\n\t"
        "s_cmov_b64 %[m1], %[n]\n\t"
        "s_or_b64 %[n], %[m2], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 0x80\n\t"
        "s_cmov_b64 %[m2], %[n]\n\t"
        "s_or_b64 %[n], %[m3], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 0x81\n\t"
        "s_cmov_b64 %[m3], %[n]"
        : [m0] "+s"(S.inP[0]), [m1] "+s"(S.inP[1]), [m2] "+s"(S.inP[2]), [m3] "+s"(S.inP[3]), [t] "=&s"(t), [n] "=&s"(n), [k] "=&s"(k)
        : [j] "s"(j)
        : "scc");
}
__device__ __forceinline__ void clear_passive(VoxState &S, int j) {
    unsigned long long t, n;
    int k;
    asm("s_lshr_b32 %[k], %[j], 1\n\t"
        "s_lshl_b64 %[t], 1, %[k]\n\t"
        "s_and_b32 %[k], %[j], 0x81\n\t"
        "s_andn2_b64 %[n], %[m0], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 0\n\t"
        "s_cmov_b64 %[m0], %[n]\n\t"
        "s_andn2_b64 %[n], %[m1], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 1\n\t"
        "s_cmov_b64 %[m1], %[n]\n\t"
        "s_andn2_b64 %[n], %[m2], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 0x80\n\t"
        "s_cmov_b64 %[m2], %[n]\n\t"
        "s_andn2_b64 %[n], %[m3], %[t]\n\t"
        "s_cmp_eq_u32 %[k], 0x81\n\t"
        "s_cmov_b64 %[m3], %[n]"
        : [m0] "+s"(S.inP[0]), [m1] "+s"(S.inP[1]), [m2] "+s"(S.inP[2]), [m3] "+s"(S.inP[3]), [t] "=&s"(t), [n] "=&s"(n), [k] "=&s"(k)
        : [j] "s"(j)
        : "scc");
}

// ---- round 4: the dual with its LDS round trips overlapped ---------------------------------------------------
typedef __attribute__((address_space(3))) const double lds_cdouble;
__device__ __forceinline__ unsigned lds_addr(const double *q) { return (unsigned)(size_t)q; }  // the LDS byte address is the low half of the flat one
__device__ __forceinline__ double lds_at(unsigned a) { return *reinterpret_cast<lds_cdouble *>((size_t)a); }
// x by bin into xbuf (halo zeroed) and the bins by position into ps, padded with kPadBin to the end of the slot
__device__ __forceinline__ void stage_bins(double *xbuf, lds_int *ps, int p, int lane, const double (&x)[kPS], const int (&pidx)[kPS]) {
    const double2 zero2 = {0.0, 0.0};
    double *lo = xbuf + 2 + 2 * lane, *hi = lo + 128;
    *reinterpret_cast<double2 *>(lo) = zero2;
    *reinterpret_cast<double2 *>(hi) = zero2;
    if (lane < 2) *reinterpret_cast<double2 *>(xbuf + 258 * lane) = zero2;
    lds_order();
#pragma unroll
    for (int s = 0; s < kPS; ++s) {
        const int i = lane + kW * s;
        if (kW * s <= p) {  // wave uniform
            // positions >= p of the slot: the padding bin -- whose x (xbuf[2 + kPadBin]) is zero and stays so, because x is zero at
            // those positions (a removal shifts zeros in): one select, then every lane stores without a branch
            const int bin = i < p ? pidx[s] : kPadBin;
            ps[i] = (int)lds_addr(xbuf + 2) + 8 * bin;  // LDS address of x by bin: a gather of x needs no address arithmetic
            xbuf[2 + bin] = x[s];
        }
    }
    lds_order();
}
// the pair of bins at q and its two neighbours on either side: three 16-byte reads
struct Win {
    double2 lo, mid, hi;
};
__device__ __forceinline__ Win load_win(const double *q) {
    Win w;
    w.lo = *reinterpret_cast<const double2 *>(q - 2);
    w.mid = *reinterpret_cast<const double2 *>(q);
    w.hi = *reinterpret_cast<const double2 *>(q + 2);
    return w;
}
// taps d = -2 .. 2 of out_j = sum_d c[d + 2] v[j + d]  (REV: c[2 - d], the transposed band)
template <bool REV, int HB> __device__ __forceinline__ void band_eval(const Win &w, const double (&c)[5], double &o0, double &o1) {
    const double cm1 = REV ? c[3] : c[1], cp1 = REV ? c[1] : c[3], cm2 = REV ? c[4] : c[0], cp2 = REV ? c[0] : c[4];
    double a = cm1 * w.lo.y, b = cm1 * w.mid.x;
    a = fma(c[2], w.mid.x, a);
    b = fma(c[2], w.mid.y, b);
    a = fma(cp1, w.mid.y, a);
    b = fma(cp1, w.hi.x, b);
    if (HB > 1) {
        a = fma(cm2, w.lo.x, a);
        b = fma(cm2, w.lo.y, b);
        a = fma(cp2, w.hi.x, a);
        b = fma(cp2, w.hi.y, b);
    }
    o0 = a;
    o1 = b;
}
template <bool REV> __device__ __forceinline__ void band_eval4(int hb, const Win &wl, const Win &wh, const double (&c)[5], double (&o)[kSlots]) {
    if (hb > 1) {
        band_eval<REV, 2>(wl, c, o[0], o[1]);
        band_eval<REV, 2>(wh, c, o[2], o[3]);
    } else {
        band_eval<REV, 1>(wl, c, o[0], o[1]);
        band_eval<REV, 1>(wh, c, o[2], o[3]);
    }
}
// every lane: (B_P x_P)[m], m = lane & 31, with x read by bin out of xbuf.  Half wave h takes the positions 16 k + 8 h ..
// + 7 of step k (two ds_read_b128 bring their bins); the bins of step k + 1 are requested before the gathers of step k, so
// a step costs one LDS round trip.  `between` runs while the gathers of the first step are in flight.
template <class F>
__device__ __forceinline__ double bx_gather(const double *Bl, const double *xbuf, const lds_int *ps, int p, int lane, F &&between) {
    const int m = lane & 31, h = lane >> 5;
    const unsigned cB = lds_addr(Bl + m * kBStride) - lds_addr(xbuf + 2);  // from x of a bin to this lane's element of the bin's column
    typedef int int4v __attribute__((ext_vector_type(4)));
    const lds_int *pj = ps + 8 * h;
    int4v j0 = *reinterpret_cast<const int4v *>(pj), j1 = *reinterpret_cast<const int4v *>(pj + 4);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    double bv[8], xv[8];
    auto issue = [&](int i) {
        const unsigned jv[8] = {(unsigned)j0.x, (unsigned)j0.y, (unsigned)j0.z, (unsigned)j0.w, (unsigned)j1.x, (unsigned)j1.y, (unsigned)j1.z, (unsigned)j1.w};
        j0 = *reinterpret_cast<const int4v *>(pj + i + 16);  // beyond the staged part: stale entries, never used
        j1 = *reinterpret_cast<const int4v *>(pj + i + 20);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bv[u] = lds_at(jv[u] + cB);
            xv[u] = lds_at(jv[u]);
        }
    };
    auto consume = [&]() {
        a0 = fma(xv[0], bv[0], a0);
        a1 = fma(xv[1], bv[1], a1);
        a2 = fma(xv[2], bv[2], a2);
        a3 = fma(xv[3], bv[3], a3);
        a0 = fma(xv[4], bv[4], a0);
        a1 = fma(xv[5], bv[5], a1);
        a2 = fma(xv[6], bv[6], a2);
        a3 = fma(xv[7], bv[7], a3);
    };
    issue(0);  // p = 0: padding positions only (zeros)
    between();
    consume();
    for (int i = 16; i < p; i += 16) {
        issue(i);
        consume();
    }
    return swap_add32((a0 + a1) + (a2 + a3));
}
// out[s] (bin binof(lane, s)) = sum_m B[m][bin] v[m] as bt_times; `between` runs behind the first eight row reads
// rows >= n_meas of the LDS basis are zero: the product stops at `mrows` = n_meas rounded up to the eight rows of a loop step
typedef double dbl2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const dbl2v glb_cdbl2v;
#ifndef PNX_BLK_DIET
#define PNX_BLK_DIET 1  // round 5, instruction diet of the dual (0: as before; A/B builds)
#endif
// `between` runs behind the first eight row reads and leaves in `start` what the sums start from (the negated regulariser term of the
// dual: four subtractions behind the product become none)
template <class F>
__device__ __forceinline__ void bt_times_h(const double *Bl, const double *v, int mrows, int lane, double (&out)[kSlots], F &&between,
                                           const double (&start)[kSlots]) {
    const double *col = Bl + 2 * lane;
    double2 c0[4], c1[4], d0[4], d1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c0[r] = *reinterpret_cast<const double2 *>(col + r * kBStride);
        c1[r] = *reinterpret_cast<const double2 *>(col + r * kBStride + 128);
    }
    between();
#pragma unroll
    for (int s = 0; s < kSlots; ++s) out[s] = PNX_BLK_DIET ? -start[s] : 0.0;
#pragma unroll 1
    for (int m = 0; m < mrows; m += 8) {
        const double *nx = col + (m + 4) * kBStride;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d0[r] = *reinterpret_cast<const double2 *>(nx + r * kBStride);
            d1[r] = *reinterpret_cast<const double2 *>(nx + r * kBStride + 128);
        }
        {
            const double2 v01 = *reinterpret_cast<const double2 *>(v + m);
            const double2 v23 = *reinterpret_cast<const double2 *>(v + m + 2);
            const double vv[4] = {v01.x, v01.y, v23.x, v23.y};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[0] = fma(c0[r].x, vv[r], out[0]);
                out[1] = fma(c0[r].y, vv[r], out[1]);
                out[2] = fma(c1[r].x, vv[r], out[2]);
                out[3] = fma(c1[r].y, vv[r], out[3]);
            }
        }
        const double *ny = col + ((m + 8) & (kBMeas - 1)) * kBStride;  // the last step re-reads rows 0 .. 3 (unused)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c0[r] = *reinterpret_cast<const double2 *>(ny + r * kBStride);
            c1[r] = *reinterpret_cast<const double2 *>(ny + r * kBStride + 128);
        }
        {
            const double2 v01 = *reinterpret_cast<const double2 *>(v + m + 4);
            const double2 v23 = *reinterpret_cast<const double2 *>(v + m + 6);
            const double vv[4] = {v01.x, v01.y, v23.x, v23.y};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[0] = fma(d0[r].x, vv[r], out[0]);
                out[1] = fma(d0[r].y, vv[r], out[1]);
                out[2] = fma(d1[r].x, vv[r], out[2]);
                out[3] = fma(d1[r].y, vv[r], out[3]);
            }
        }
    }
}
// w = B^T (y - B_P x_P) - R^T (R x), all out of LDS.  LDS round trips in sequence: bins of the first positions -> column
// gathers (one per 16 positions) -> B^T r; the stencil of R rides on the first gathers, the one of R^T on the first row reads.
#ifndef PNX_BLK_RTR
#define PNX_BLK_RTR 1  // half bandwidth 1 (orders 1 and 2): R^T (R x) as one five-point stencil of x (0: two three-point passes through LDS; A/B builds)
#endif
// the pair of bins at the centre of w: sum_e g[|e|] x[j + e], e = -2 .. 2
__device__ __forceinline__ void rtr_eval(const Win &w, double g0, double g1, double g2, double &o0, double &o1) {
    double a = g2 * w.lo.x, b = g2 * w.lo.y;
    a = fma(g1, w.lo.y, a);
    b = fma(g1, w.mid.x, b);
    a = fma(g0, w.mid.x, a);
    b = fma(g0, w.mid.y, b);
    a = fma(g1, w.mid.y, a);
    b = fma(g1, w.hi.x, b);
    a = fma(g2, w.hi.x, a);
    b = fma(g2, w.hi.y, b);
    o0 = a;
    o1 = b;
}
__device__ __forceinline__ void dual_residual_form(const double *Bl, double *xbuf, lds_int *ps, double *rb, const double (&rc)[5], int hb,
                                                   int n, int mrows, int lane, double yreg, const VoxState &S, double (&w)[kSlots]) {
    const int p = __builtin_amdgcn_readfirstlane(S.p);
    lds_order();
    stage_bins(xbuf, ps, p, lane, S.x, S.pidx);
    double *lo = xbuf + 2 + 2 * lane, *hi = lo + 128;
    const Win wl = load_win(lo), wh = load_win(hi);
    // Half bandwidth 1 (orders 1 and 2): (R^T R x)_j = rg[2] (x_{j-2} + x_{j+2}) + rg[1] (x_{j-1} + x_{j+1}) + rg[0] x_j, minus what the rows
    // -1 and n -- which R does not have -- would have added to the bins 0 and n - 1 (x is zero outside [0, n)): one stencil of x, no
    // second pass, no round trip of t through LDS.  Half bandwidth 2 (order 3): t = R x by bin through xbuf, then R^T t.
    const bool merged = PNX_BLK_RTR && hb == 1;  // wave uniform
    double t[kSlots], u[kSlots];
    // the stencil's coefficients are read BEFORE the gathers are issued: scalar loads share their counter with LDS, so a load behind
    // them waits for every gather in flight (it did: the stencil was meant to ride on that round trip and ran behind it)
    double g0 = 0, g1 = 0, g2 = 0, e0 = 0, e1 = 0;
    if (merged) {
        KArgs *K = kargs();
        g0 = K->rg[0], g1 = K->rg[1], g2 = K->rg[2], e0 = K->rg[3], e1 = K->rg[4];
    }
    const double bx = bx_gather(Bl, xbuf, ps, p, lane, [&]() {
        if (merged) {
            rtr_eval(wl, g0, g1, g2, u[0], u[1]);
            rtr_eval(wh, g0, g1, g2, u[2], u[3]);
            // the two end corrections, each on ONE lane of ONE slot: a select on the result (bin 0), and for bin n - 1 -- whose slot is wave
            // uniform -- scalar branches to the one slot concerned (four EXEC-masked blocks with a scalar load each before)
            const double c0 = fma(-e0, wl.mid.x, u[0]);
            u[0] = lane == 0 ? c0 : u[0];
            const int jl = n - 1, sl = slot_of_bin(jl);
            const bool at = lane == lane_of_bin(jl);
            if (sl == 3) {
                const double c = fma(-e1, wh.mid.y, u[3]);
                u[3] = at ? c : u[3];
            } else if (sl == 2) {
                const double c = fma(-e1, wh.mid.x, u[2]);
                u[2] = at ? c : u[2];
            } else if (sl == 1) {
                const double c = fma(-e1, wl.mid.y, u[1]);
                u[1] = at ? c : u[1];
            } else {
                const double c = fma(-e1, wl.mid.x, u[0]);
                u[0] = at ? c : u[0];
            }
        } else
            band_eval4<false>(hb, wl, wh, rc, t);
    });
    lds_order();
    if (lane < kBMeas) rb[lane] = yreg - bx;
    Win ul, uh;
    if (!merged) {
        // t by bin through xbuf: every gather of x has been issued, and the LDS executes a wave's instructions in order
        *reinterpret_cast<double2 *>(lo) = double2{t[0], t[1]};
        *reinterpret_cast<double2 *>(hi) = double2{t[2], t[3]};
        lds_order();
        // rows >= n of R do not exist: x is zero there, so only t_n and t_{n + 1} can be non-zero -- two stores instead of a
        // select on every lane's four values
        if (lane < 2) xbuf[2 + n + lane] = 0.0;
        lds_order();
        ul = load_win(lo);
        uh = load_win(hi);
    }
    lds_order();
    bt_times_h(Bl, rb, mrows, lane, w, [&]() {
        if (!merged) band_eval4<true>(hb, ul, uh, rc, u);  // (R^T t)_j = sum_d c[d + 2] t_{j - d}
    }, u);
    lds_order();
    if (!PNX_BLK_DIET) {
#pragma unroll
        for (int s = 0; s < kSlots; ++s) w[s] -= u[s];
    }
    // (the caller masks the passive bins)
}

// ---- round 5: the dual of a SMALL passive set in Gram form ---------------------------------------------------------
// w = A^T y - G[:, P] x_P: p rows of G (2 KB each, out of L2: their addresses are known when the iteration starts, so all of a
// batch are in flight together) instead of the 64 KB basis out of LDS plus p column gathers -- at p <= kGramP that is fewer bytes,
// a fifth of the vector instructions and one or two round trips instead of eight.  A^T y is this voxel's dual at p = 0 (the residual
// form with x = 0), kept in registers.  Same quantity in another summation order; the factor M already comes from the same G.
#ifndef PNX_BLK_GRAMP
#define PNX_BLK_GRAMP 16  // round 5, final kernel, C4 volume: 16 / 20 / 24: 393.7 / 385.9 / 386.2 ms, 137 / 144 / 151 GB behind the L2 (more rows of G through the L2 evict more rows of M), largest coefficient error of 524 288 voxels 5.1e-7 / 9.0e-7 / 6.0e-7 of the peak: 16 (profiles/r05_nnls_experiments.md, sections 3, 16 and 17)
#endif
constexpr int kGramP = PNX_BLK_GRAMP;
static_assert(kGramP >= 0 && kGramP < kW && kGramP % 4 == 0, "the Gram-form dual reads positions of the first register slot only, in batches of four");
// the bins by position into ps (what stage_bins leaves for the append's prefetch), padded with kPadBin
__device__ __forceinline__ void stage_ps(const double *xbuf, lds_int *ps, int p, int lane, const int (&pidx)[kPS]) {
    lds_order();
    ps[lane] = (int)lds_addr(xbuf + 2) + 8 * (lane < p ? pidx[0] : kPadBin);
    lds_order();
}
// Rows K .. K + 3 of the batch: the bins come out of lane K + u of pidx (v_readlane with a constant lane), the row address is ONE vector
// add on top of the kernel argument (scalar base + 32-bit lane offset), and positions >= p need no select: their x is zero (the removal
// shifts zeros in) and their stale bins are rows of G like any other.
typedef __attribute__((address_space(1))) const char glb_cchar;
template <int K> __device__ __forceinline__ void gram_load4(glb_cchar *G, unsigned lane16, const VoxState &S, dbl2v (&g)[4][2]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned j = (unsigned)__builtin_amdgcn_readlane(S.pidx[0], K + u);
        glb_cchar *row = G + (lane16 + (j << 11));  // row j of G, this lane's pair of bins
        g[u][0] = *reinterpret_cast<glb_cdbl2v *>(row);
        g[u][1] = *reinterpret_cast<glb_cdbl2v *>(row + 1024);
    }
}
template <int K> __device__ __forceinline__ void gram_use4(const dbl2v (&g)[4][2], const VoxState &S, double (&w)[kSlots]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const double xk = -rl(S.x[0], K + u);
        w[0] = fma(xk, g[u][0].x, w[0]);
        w[1] = fma(xk, g[u][0].y, w[1]);
        w[2] = fma(xk, g[u][1].x, w[2]);
        w[3] = fma(xk, g[u][1].y, w[3]);
    }
}
// rows K .. K + 3 are on their way in `cur` (p > K): the next four are requested before these are used.  Unrolled over the batches (the
// chain is at most kGramP / 4 long): every batch has its own registers -- the loop it replaces swapped its two buffers by copying
// them, 16 moves per batch -- and every lane index is a constant.
template <int K>
__device__ __forceinline__ void gram_chain(glb_cchar *G, unsigned lane16, int p, const VoxState &S, const dbl2v (&cur)[4][2], double (&w)[kSlots]) {
    if constexpr (K + 4 < kGramP) {
        if (p > K + 4) {
            dbl2v nxt[4][2];
            gram_load4<K + 4>(G, lane16, S, nxt);
            gram_use4<K>(cur, S, w);
            gram_chain<K + 4>(G, lane16, p, S, nxt, w);
            return;
        }
    }
    gram_use4<K>(cur, S, w);
}
__device__ __forceinline__ void dual_gram_form(const double *xbuf, lds_int *ps, int lane, const double (&w0)[kSlots], const VoxState &S, double (&w)[kSlots]) {
    const int p = __builtin_amdgcn_readfirstlane(S.p);
    stage_ps(xbuf, ps, p, lane, S.pidx);
    glb_cchar *G = (glb_cchar *)kargs()->G;
    const unsigned lane16 = 16u * (unsigned)lane;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) w[s] = w0[s];
    dbl2v first[4][2];
    gram_load4<0>(G, lane16, S, first);
    gram_chain<0>(G, lane16, p, S, first, w);
}

// blocks (I, K), K <= I < NI, of this wave's M: every load is issued before the first use
// The block rows of M that live in the global slab (8 I >= kLdsM), K <= I < NI: every load is issued before the first use.
template <int NI> __device__ __forceinline__ void load_blocks(const MRef &M, int la, int lb, double (&blk)[NI][NI]) {
#pragma unroll
    for (int I = 0; I < NI; ++I) {
        if (8 * I < kLdsM) continue;
        const int base = (I + 1) * (32 * I + 8 * la) + lb;
#pragma unroll
        for (int K = 0; K <= I; ++K) blk[I][K] = M.g[CK(base + 8 * K, kMSlab, 1, I)];
    }
}
// Block row I of M: out of LDS when it is needed (short latency, no register held meanwhile), else the preloaded copy.
template <int NI> __device__ __forceinline__ void block_row(const MRef &M, int I, int la, int lb, const double (&blk)[NI][NI],
                                                            double (&row)[NI]) {
    const int base = (I + 1) * (32 * I + 8 * la) + lb;
#pragma unroll
    for (int K = 0; K < NI; ++K)
        if (K <= I) row[K] = 8 * I < kLdsM ? M.l[base + 8 * K] : blk[I][K];
}

// Column jmax wants to enter.  l = M g (g = G[P, jmax]), lam^2 = G_jj - |l|^2, Lawson-Hanson independence test; when it
// passes: new row of M = [-(l^T M) / lam, 1 / lam], z = x + row * qn (x == M^T q whenever a column enters), q_p = qn.
// Returns false when the column is rejected (nothing changed).
// One sweep over the blocks: block row I gives l_{8 I + a} (FMA per block, butterfly over b), which goes straight into
// the column sums of l^T M, so a block is dead once its row is done; all block loads are issued up front.
// What an append needs that does not depend on the candidate, requested before the arg-max so that its latency hides behind it:
// the bins of the positions in column layout (as byte offsets into a row of G).  (The blocks of M that live in the slab would
// qualify too, but held across the arg-max they push the append out of its registers: 64 -> 256 bytes of scratch, measured.)
template <int NI>
__device__ __forceinline__ void append_prefetch(const lds_int *ps, int lb, unsigned (&goff)[NI]) {
    const int ps_base = (int)lds_addr(reinterpret_cast<const double *>(ps) + kMaxPos / 2 + 2);  // = xbuf + 2 of this wave
#pragma unroll
    for (int K = 0; K < NI; ++K) goff[K] = (unsigned)(ps[8 * K + lb] - ps_base);  // ps holds LDS addresses of x by bin
}
// What an outer iteration's candidate step computes before anything changes: the candidate bin, the independence test, and the new row
// of M up to its scale.  The voxel state is only READ here -- the three instantiations (by the number of block rows in use) merge in
// a handful of fresh values, not in a copy of the state: returning "rejected / accepted" from a function that also updates the state
// cost every outer iteration ~25 register moves (the unchanged parts of the state copied to where the changed ones were computed).
struct AppendOut {
    double a1[kPS];  // l^T M by position (exactly +0 at and behind position p)
    double ilam, qn;
    int jmax;
    bool ok;
};
// lowest bin that attains the maximum `best` of the masked duals: ballots and scalar bit scans (bin = 128 (s >> 1) + 2 lane + (s & 1)).
// s_ff1 answers -1 for an empty mask, so as UNSIGNED numbers the four candidates order themselves, an empty one last: three s_min_u32,
// no select.  Some lane holds the maximum (best > 0 is one of the w), so the result is a bin; the & 255 only keeps a row index of G in
// range whatever happens.
__device__ __forceinline__ unsigned ff1(unsigned long long m) {  // the instruction as it is: __ffsll() - 1 puts a compare and a select behind it for the empty mask
    int r;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return (unsigned)r;
}
__device__ __forceinline__ int argmax_bin(const double (&w)[kSlots], double best) {
    const unsigned long long m0 = __ballot(w[0] == best), m1 = __ballot(w[1] == best);
    const unsigned long long m2 = __ballot(w[2] == best), m3 = __ballot(w[3] == best);
    const unsigned b0 = ff1(m0) << 1;
    const unsigned b1 = (ff1(m1) << 1) | 1u;
    const unsigned b2 = (ff1(m2) << 1) | 128u;
    const unsigned b3 = (ff1(m3) << 1) | 129u;
    unsigned lo = b0 < b1 ? b0 : b1, hi = b2 < b3 ? b2 : b3;
    asm("" : "+s"(lo), "+s"(hi));  // scalar registers: left alone the compiler folds the three minima into one v_min3_u32, and the bin -- a row address of G -- lives in a vector register from then on
    return (int)((lo < hi ? lo : hi) & 255u);
}
template <int NI>
__device__ __forceinline__ void append_prepare(const double *G, const MRef &M, const lds_int *ps, int lane, const double (&w)[kSlots], double wj,
                                               const VoxState &S, AppendOut &A) {
    const int la = lane >> 3, lb = lane & 7;
    unsigned goff[NI];
    append_prefetch<NI>(ps, lb, goff);  // its LDS round trip hides behind the bit scans
    const int jmax = argmax_bin(w, wj);
    A.jmax = jmax;
    const int p = __builtin_amdgcn_readfirstlane(S.p);
    const double *grow = G + (size_t)jmax * kNnlsMaxBins;
    double blk[NI][NI];
    wave_sync();  // this wave's stores to M (previous append / removal) have long landed: the wait is free, the order is kept
    load_blocks<NI>(M, la, lb, blk);
    const double Gjj = uni(grow[CK(jmax, kNnlsMaxBins, 2, p)]);
    // g in column layout: lane (a, b) holds g_{8 K + b}
    double gc[NI];
#pragma unroll
    for (int K = 0; K < NI; ++K) {  // no mask: behind position p the staged bins are 0 and the columns of M are zero
        gc[K] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(grow) + goff[K]);
    }
    double rK[NI];  // column sums of l^T M, lane (a, b) holds the partial sum over its rows of column 8 K + b
#pragma unroll
    for (int K = 0; K < NI; ++K) rK[K] = 0;
    double ll = 0;
#pragma unroll
    for (int I = 0; I < NI; ++I) {
        double acc = 0, row[NI];
        block_row<NI>(M, I, la, lb, blk, row);
#pragma unroll
        for (int K = 0; K <= I; ++K) acc = fma(row[K], gc[K], acc);
        double lr = allreduce_b(acc);  // l_{8 I + a}, the same in the lanes (a, *)
        // rows >= p: zero in the LDS part of M (kept so: no mask); in the global part they may hold a previous voxel's values
        if (8 * I >= kLdsM) lr = (8 * I + la < p) ? lr : 0.0;
        ll = fma(lr, lr, ll);
#pragma unroll
        for (int K = 0; K <= I; ++K) rK[K] = fma(row[K], lr, rK[K]);
    }
    ll = sum_a_uni(ll);
    // rows >= 48: row by row (lanes over the columns), four rows in flight
    double a1[kPS];    // l^T M by position
#pragma unroll
    for (int s = 0; s < kPS; ++s) a1[s] = 0;
    if (NI == kNIMax && p > kRows2D) {
        double g[kPS];
#pragma unroll
        for (int s = 0; s < kPS; ++s) g[s] = (lane + kW * s < p) ? grow[CK(S.pidx[s], kNnlsMaxBins, 4, p)] : 0.0;
        auto one = [&](int i, auto T) {
            constexpr int si = decltype(T)::value;
            const int rbase = moff(i);
            double part = 0, m[si + 1];
#pragma unroll
            for (int s = 0; s <= si; ++s) {
                const int k = lane + kW * s;
                m[s] = (s < si || k <= i) ? M.g[CK(rbase + k, kMSlab, 5, i)] : 0.0;  // past the row end (last slot of the row only): masked
                part = fma(m[s], g[s], part);
            }
            const double li = wave_sum(part);
            ll = fma(li, li, ll);
#pragma unroll
            for (int s = 0; s <= si; ++s) a1[s] = fma(li, m[s], a1[s]);
        };
        auto four = [&](int i, auto T) {
            constexpr int si = decltype(T)::value;
            double m[4][si + 1];
            int mo[4];
            moff_batch<4>(i, lane, mo);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rbase = mo[r];
#pragma unroll
                for (int s = 0; s <= si; ++s) m[r][s] = M.g[CK(rbase + lane + kW * s, kMSlab, 6, i + r)];
            }
            double part[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                part[r] = 0;
#pragma unroll
                for (int s = 0; s <= si; ++s) {
                    m[r][s] = (s < si || lane + kW * s <= i + r) ? m[r][s] : 0.0;  // only the row's last slot reaches beyond its end
                    part[r] = fma(m[r][s], g[s], part[r]);
                }
            }
            // the four wave sums in one go: two 32-lane swaps fold rows (0, 2) and (1, 3) into half waves, a 16-lane swap folds
            // those into quarter waves (row r in lanes 16 r .. 16 r + 15), four butterfly steps finish all four at once --
            // 29 instructions for the four totals instead of 20 per row
            double t = rs16(rs32(part[0], part[2]), rs32(part[1], part[3]));
            t += dppx<0xB1>(t);   // quad_perm [1,0,3,2]
            t += dppx<0x4E>(t);   // quad_perm [2,3,0,1]
            t += dppx<0x141>(t);  // row_half_mirror
            t += dppx<0x140>(t);  // row_mirror
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double li = rl(t, 16 * r);
                ll = fma(li, li, ll);
#pragma unroll
                for (int s = 0; s <= si; ++s) a1[s] = fma(li, m[r][s], a1[s]);
            }
        };
        for_pos4n<kPS>(kRows2D, p, four, one);
    }
    ll = uni(ll);
    // wave-uniform scalar algebra on v_rsq_f64 + Newton (pnx_nnls.hip)
    const double lam2 = Gjj - ll;
    const bool indep = lam2 > 64.0 * 2.220446049250313e-16 * Gjj;
    const double ilam = indep ? rsqrt_nr(lam2) : 0.0;
    double lam = lam2 * ilam;
    lam = fma(0.5 * ilam, fma(-lam, lam, lam2), lam);  // sqrt(lam2) to within an ulp
    // Lawson-Hanson linear-independence test: (|l| + 0.01 lam) - |l| > 0.  It can only fail where 0.01 lam is below half an
    // ulp of |l|, i.e. lam^2 < ~1e-28 |l|^2: |l| = sqrt(ll) is computed on that side of a generous threshold only
    bool ok = true;
    if (!(lam2 > 1e-24 * ll)) {
        const double un = ll > 0 ? ll * rsqrt_nr(ll) : 0.0;
        ok = ((un + lam * 0.01) - un) > 0;
    }
    // ztest = qn / lam with qn = (a_j^T residual) / lam: the residual-form dual w_j IS a_j^T residual
    const double qn = wj * ilam;
    ok = ok && qn > 0;
#ifdef PNX_NNLS_TRACE
    if (lane == 0 && blockIdx.x == 0 && threadIdx.x < 64) printf("A p=%d j=%d lam=%.17g qn=%.17g\n", p, jmax, lam, qn);
#endif
    A.ok = ok;
    A.ilam = ilam;
    A.qn = qn;
    // column sums over a, delivered in position order (position 8 la + lb is column block K = la)
    a1[0] += reduce_scatter_a<NI>(rK, la);
#pragma unroll
    for (int s = 0; s < kPS; ++s) A.a1[s] = a1[s];
}
// The column enters: new row of M = [-(l^T M) / lam, 1 / lam], z = x + row * qn (x == M^T q whenever a column enters), q_p = qn.
// l^T M is exactly +0 at p and behind it (M has no such column yet: those entries are zero), so a -1 written into lane p of it -- one
// v_writelane of the high word -- makes the whole row ONE multiply; q and the bin of the new position are lane writes too (no EXEC
// mask, no copy of the state for the lanes that keep theirs).  x is zero at p already: positions >= p hold zeros (a removal shifts
// them in).
__device__ __forceinline__ void append_commit(const MRef &M, int lane, AppendOut &A, VoxState &S) {
    const int p = __builtin_amdgcn_readfirstlane(S.p);
    const int pbase = moff(p);
    const int width = 8 * ((p >> 3) + 1);
    const int sp = p >> 6, pl = p & 63;
    const double qs = uni(A.qn);
#pragma unroll
    for (int s = 0; s < kPS; ++s) {
        if (kW * s <= p) {  // wave uniform
            if (s == sp) {  // wave uniform
                A.a1[s] = __hiloint2double(pnx_writelane((int)0xBFF00000, pl, __double2hiint(A.a1[s])), __double2loint(A.a1[s]));
                S.q[s] = __hiloint2double(pnx_writelane(__double2hiint(qs), pl, __double2hiint(S.q[s])),
                                          pnx_writelane(__double2loint(qs), pl, __double2loint(S.q[s])));
                S.pidx[s] = pnx_writelane(A.jmax, pl, S.pidx[s]);
            }
            const double rowv = -A.a1[s] * A.ilam;
            if (__builtin_amdgcn_inverse_ballot_w64(lanes_le_nn(width - 1 - kW * s))) M.st(p, pbase + lane + kW * s, rowv);
            S.z[s] = fma(rowv, A.qn, S.x[s]);  // the rank-one update of the solution: z = x + row * qn (x_p = 0)
        }
    }
    set_passive(S, A.jmax);
    S.p = p + 1;
}

constexpr int kMaxRej = 8;  // rejected columns an outer iteration can remember (the four spare doubles behind xbuf's halo)
// A rejected column is what Lawson-Hanson answers with "w_j = 0, take the next largest".  It is rare (none in 28 000 outer
// iterations of the reference workload), so it is not worth a loop around the append -- a loop keeps w[] and a copy of the
// voxel state alive across the append and lets the compiler hoist the append's addresses in front of it: 64 bytes of scratch
// per lane written in every outer iteration, which on this chip is HBM traffic.  Instead the column's passive flag is set for
// the time being, its bin is remembered in LDS, and the outer loop evaluates the dual again (same state, same values; the
// flag masks the column); the flags are taken back when a column enters.

// z = M^T q
template <int NI>
__device__ __forceinline__ void mt_times_q(const MRef &M, double *stg, int lane, int la, int lb, VoxState &S) {
    const int p = __builtin_amdgcn_readfirstlane(S.p);
    double blk[NI][NI];
    wave_sync();
    load_blocks<NI>(M, la, lb, blk);
    lds_order();
    stg[lane] = S.q[0];
    lds_order();
    double qr[NI];
#pragma unroll
    for (int I = 0; I < NI; ++I) {  // LDS rows >= p of M are zero: what q holds there does not matter; global rows are masked
        qr[I] = stg[8 * I + la];
        if (8 * I >= kLdsM) qr[I] = (8 * I + la < p) ? qr[I] : 0.0;
    }
    lds_order();
    double zK[NI];
#pragma unroll
    for (int K = 0; K < NI; ++K) zK[K] = 0;
#pragma unroll
    for (int I = 0; I < NI; ++I) {
        double row[NI];
        block_row<NI>(M, I, la, lb, blk, row);
#pragma unroll
        for (int K = 0; K <= I; ++K) zK[K] = fma(row[K], qr[I], zK[K]);
    }
    S.z[0] = reduce_scatter_a<NI>(zK, la);
#pragma unroll
    for (int s = 1; s < kPS; ++s) S.z[s] = 0;
    if (NI == kNIMax && p > kRows2D) {
        auto one = [&](int i, auto T) {
            constexpr int si = decltype(T)::value;
            const double a = rl(S.q[si], i & 63);
            const int rbase = moff(i);
#pragma unroll
            for (int s = 0; s <= si; ++s) {
                const int k = lane + kW * s;
                const double m0 = M.g[CK(rbase + k, kMSlab, 5, i)];
                S.z[s] += a * ((s < si || k <= i) ? m0 : 0.0);
            }
        };
        auto four = [&](int i, auto T) {
            constexpr int si = decltype(T)::value;
            double m[4][si + 1];
            int mo[4];
            moff_batch<4>(i, lane, mo);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rbase = mo[r];
#pragma unroll
                for (int s = 0; s <= si; ++s) m[r][s] = M.g[CK(rbase + lane + kW * s, kMSlab, 6, i + r)];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double a = rl(S.q[si], (i + r) & 63);
#pragma unroll
                for (int s = 0; s <= si; ++s) S.z[s] += a * ((s < si || lane + kW * s <= i + r) ? m[r][s] : 0.0);
            }
        };
        for_pos4n<kPS>(kRows2D, p, four, one);
    }
}

constexpr int kBail = 2;  // internal status: the passive set wants more than kMaxPos columns, the general kernel redoes the voxel
constexpr int kDone = 3;  // internal status: no dual is positive (KKT satisfied) -- becomes 1 behind the loop

template <bool HOOK> __device__ __forceinline__ void blk_body() {
#ifdef PNX_NNLS_STAMP
    unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    long long cnt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // voxels, outer iterations, rejected candidates, removals, rotated rows, sum of p at the dual, appends with p > 48, p > 64
#endif
    extern __shared__ double dyn_lds[];
    if (kargs()->route && *kargs()->route != kargs()->route_want) return;  // uniform over the grid: the pilot of this call chose the other kernel
    const int lane = threadIdx.x & (kW - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = kargs()->n_bins, nm = kargs()->n_meas, nreg = kargs()->n_reg;
    const int m_total = nm + nreg;
    // the per-wave scratch comes first: its addresses then fit the 16-bit offset field of the DS instructions (behind the
    // 66 KB of B every access needed a VALU add for its address: 18 of them per step of the B x loop alone)
    double *scr = dyn_lds + wave * kScr;
    double *Bl = dyn_lds + kBlkWaves * kScr;
    lds_int *ps = reinterpret_cast<lds_int *>(scr);                    // [128] bin by position
    double *xbuf = scr + kMaxPos / 2;                                  // [kXbuf] x by bin (halo of 2), staging buffer of the M sweeps
    double *rb = xbuf + kXbuf;                                         // [32] residual of the measurements
    MRef M;
    M.g = (glb_double *)(kargs()->Mglob + ((size_t)blockIdx.x * kBlkWaves + wave) * kMSlab);
    M.l = (lds_double *)(dyn_lds + kBlkWaves * kScr + kBMeas * kBStride + wave * kLdsMDoubles);
    for (int e = lane; e < kLdsMDoubles; e += kW) M.l[e] = 0.0;  // rows >= p of M are zero, from the first voxel on
    for (int e = threadIdx.x; e < kBMeas * kBStride; e += kBlkWaves * kW) {
        const int m = e / kBStride, j = e - m * kBStride;
        Bl[e] = (m < nm && j < kNnlsMaxBins) ? kargs()->Bp[(size_t)m * kNnlsMaxBins + j] : 0.0;
    }
    __syncthreads();  // the only workgroup barrier: from here on the waves never meet again

    for (;;) {
        unsigned long long vq = next_voxel(kargs()->queue, lane);
        vq = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(vq >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)vq);
        long long vox = (long long)vq;
        if (kargs()->redo_list) {  // list mode: the voxels another launch handed over
            const unsigned long long nr = (unsigned long long)*kargs()->redo_count, nv = (unsigned long long)kargs()->n_vox;
            if (vq >= (nr < nv ? nr : nv)) break;
            vox = kargs()->redo_list[vq];
        } else if (vq >= (unsigned long long)kargs()->n_vox)
            break;
        const double *yv = kargs()->y + (size_t)vox * nm;

        // y by measurement: lanes 0 .. 31, duplicated in 32 .. 63
        const int ml = lane & 31;
        const double yreg = ml < nm ? yv[ml] : 0.0;
        const bool finite = __all(isfinite(yreg) ? 1 : 0) != 0;
        VoxState S;
#pragma unroll
        for (int s = 0; s < kPS; ++s) {
            S.q[s] = 0;
            S.x[s] = 0;
            S.z[s] = 0;
            S.pidx[s] = 0;
        }
#pragma unroll
        for (int s = 0; s < kSlots; ++s) S.inP[s] = __ballot(binof(lane, s) >= n);
        S.p = 0;
        S.nrej = 0;
        int iteration = 0, status = finite ? 1 : -2;
        const int max_it = kargs()->max_iter;  // once per voxel: read in the inner loop it is a scalar load and a wait for every LDS access in flight per iteration
        double w[kSlots];
        // A^T y of this voxel = B^T y: its dual at p = 0 (x = 0: no gathers, no stencil) and the start of the Gram-form dual of small passive sets
        double w0[kSlots];
        {
            const double zero4[kSlots] = {0, 0, 0, 0};
            lds_order();
            if (lane < kBMeas) rb[lane] = yreg;
            lds_order();
            bt_times_h(Bl, rb, (nm + 7) & ~7, lane, w0, []() {}, zero4);
            lds_order();
        }
        STAMP(0);
        COUNT(0, 1);

        // Two loops.  The inner one is the outer iteration of Lawson-Hanson as the reference workload runs it: dual, candidate, append, inner
        // loop.  Its rare ways out -- KKT satisfied, no room, a rejected column -- LEAVE it with the voxel state untouched, so the state is
        // loop carried along ONE path and updated in place.  With `continue` on those paths the loop's latch merged the unchanged state
        // with the one an append leaves, and the compiler paid for that merge with ~30 register moves in every outer iteration.
        int jrej = -1;  // >= 0: the column the candidate step rejected
        for (;;) {
        while (status == 1 && S.p < n && S.p < m_total) {
            // ---- dual in residual form, all out of LDS: w = B^T (y - B_P x_P) - R^T (R x)
            COUNT(1, 1);
            COUNT(5, S.p);
            COUNT(6, S.p > 48 ? 1 : 0);
            COUNT(7, S.p > 64 ? 1 : 0);
            COUNT(8, S.p == 0 ? 1 : 0);
            COUNT(9, S.p >= 1 && S.p <= 4 ? 1 : 0);
            COUNT(10, S.p >= 5 && S.p <= 8 ? 1 : 0);
            COUNT(11, S.p >= 9 && S.p <= 12 ? 1 : 0);
            COUNT(12, S.p >= 13 && S.p <= 16 ? 1 : 0);
            COUNT(13, S.p >= 17 && S.p <= 24 ? 1 : 0);
            COUNT(14, S.p >= 25 && S.p <= 32 ? 1 : 0);
            {
                const int ld = fresh(lane);
                KArgs *K = kargs();
                const double rc[5] = {K->rc[0], K->rc[1], K->rc[2], K->rc[3], K->rc[4]};
#ifdef PNX_BLK_FULL_ROWS  // (A/B builds) all 32 rows of the LDS basis whatever the plan's number of measurements, as up to round 4
                const int mrows = kBMeas;
#else
                const int mrows = (K->n_meas + 7) & ~7;
#endif
                if (S.p == 0) {  // x = 0: the dual is A^T y, worked out when the voxel was fetched
#pragma unroll
                    for (int s = 0; s < kSlots; ++s) w[s] = w0[s];
                    stage_ps(xbuf, ps, 0, ld, S.pidx);  // what the append's prefetch reads: the padding bin at every position
                } else if (kGramP > 0 && S.p <= kGramP)
                    dual_gram_form(xbuf, ps, ld, w0, S, w);
                else
                    dual_residual_form(Bl, xbuf, ps, rb, rc, K->rhb, n, mrows, ld, yreg, S, w);
                // passive (and rejected, and non-existent) bins leave the arg-max: their dual becomes a huge negative number -- the HIGH word
                // alone is replaced (0xffe00000: -8.99e307 or below whatever the low word holds; one select per bin instead of two for -inf)
#pragma unroll
                for (int s = 0; s < kSlots; ++s) {
                    if (PNX_BLK_DIET) {
                        const int hi = __builtin_amdgcn_inverse_ballot_w64(S.inP[s]) ? (int)0xffe00000 : __double2hiint(w[s]);
                        w[s] = __hiloint2double(hi, __double2loint(w[s]));
                    } else if (__builtin_amdgcn_inverse_ballot_w64(S.inP[s]))
                        w[s] = -INFINITY;
                }
            }
            STAMP(1);

            {
                // largest dual: none positive = KKT satisfied, the voxel is done.  (Tested here, in front of the instantiations of the candidate
                // step: a return value of theirs would merge the unchanged voxel state with the one an append leaves -- 15 register moves
                // in every outer iteration for the sake of the last.)
                const double best = wave_max_pos_uni(max1(max1(w[0], w[1]), max1(w[2], w[3])));
                if (!(best > 0)) {
                    status = kDone;
                    break;
                }
                if (S.p >= kMaxPos) {  // no room for another column in this kernel's registers: the voxel is handed over
                    status = kBail;
                    break;
                }
                const int nI = (S.p >> 3) + 1;  // block rows in use, the one the new row goes to included
                const int lc = fresh(lane);
                const double *Gp = kargs()->G;
                AppendOut A;
                if (nI <= 2)
                    append_prepare<2>(Gp, M, ps, lc, w, best, S, A);
                else if (nI <= 4)
                    append_prepare<4>(Gp, M, ps, lc, w, best, S, A);
                else if (nI <= 6 || kNIMax == 6)
                    append_prepare<6>(Gp, M, ps, lc, w, best, S, A);
                else
                    append_prepare<kNIMax>(Gp, M, ps, lc, w, best, S, A);
                bool rejected = !A.ok;
                if (HOOK) {
                    // test hook (instantiation HOOK only: a kernel of its own, so that the product's outer iteration carries none of it): reject
                    // valid columns, so that the bookkeeping of rejected columns runs although the reference workload never rejects one (the
                    // minimiser is unique with a regulariser: the solve must arrive at the same spectrum by another path)
                    const int tk = kargs()->test_rej_k;
                    rejected = rejected || (tk > 0 && (__builtin_amdgcn_readfirstlane(S.p) % tk) == tk - 1 && __builtin_amdgcn_readfirstlane(S.nrej) < kargs()->test_rej_n);
                }
                lds_int *rejlist = reinterpret_cast<lds_int *>(xbuf + 2 + kNnlsMaxBins + 2);  // the four spare doubles behind the halo
                if (rejected) {  // (the bookkeeping is behind this loop)
                    jrej = A.jmax;
                    break;
                }
                append_commit(M, lc, A, S);
                const int nr = __builtin_amdgcn_readfirstlane(S.nrej);
                for (int k = 0; k < nr; ++k)  // the columns rejected meanwhile may be looked at again
                    clear_passive(S, __builtin_amdgcn_readfirstlane(rejlist[k]));
                S.nrej = 0;
            }
            STAMP(2);
            STAMP(3);

            // ---- inner loop: keep the passive-set solution feasible
            for (;;) {
                iteration += 1;
                if (iteration == max_it) {
                    status = 0;
                    break;
                }
                const int p = __builtin_amdgcn_readfirstlane(S.p);
                // positions with z <= 0 as wave masks (the range test i < p is scalar: no vector compare, and the second
                // register slot is looked at only where it is in use)
                unsigned long long vm[kPS], vany = 0;
#pragma unroll
                for (int s = 0; s < kPS; ++s) {
                    vm[s] = p > kW * s ? (__ballot(S.z[s] <= 0) & lanes_le_nn(p - 1 - kW * s)) : 0ull;
                    vany |= vm[s];
                }
                if (!vany) {  // feasible: x = z (positions >= p hold nothing that is read)
#pragma unroll
                    for (int s = 0; s < kPS; ++s) S.x[s] = S.z[s];
                    break;
                }
                double T[kPS];
#pragma unroll
                for (int s = 0; s < kPS; ++s) {
                    T[s] = INFINITY;
                    if (vm[s]) {  // wave uniform
                        if (__builtin_amdgcn_inverse_ballot_w64(vm[s])) T[s] = -S.x[s] / (S.z[s] - S.x[s]);
                    }
                }
                // smallest step (ties: first position -- Lawson-Hanson keeps the first minimum)
                double tmin = T[0];
#pragma unroll
                for (int s = 1; s < kPS; ++s) tmin = T[s] < tmin ? T[s] : tmin;
                const double alpha = -uni(allreduce_max(-tmin));
                int bpos = kNone;
                {
#pragma unroll
                    for (int s = kPS - 1; s >= 0; --s) {  // the first position that attains it
                        const unsigned long long bs = vm[s] ? (__ballot(T[s] == alpha) & vm[s]) : 0ull;
                        if (bs) bpos = kW * s + __ffsll(bs) - 1;
                    }
                    if (!(alpha < INFINITY)) bpos = kNone;
                }
                if (bpos == kNone) {  // no finite step: as Lawson-Hanson, nothing to remove
#pragma unroll
                    for (int s = 0; s < kPS; ++s) S.x[s] = S.z[s];
                    break;
                }
#pragma unroll
                for (int s = 0; s < kPS; ++s) S.x[s] = S.x[s] + alpha * (S.z[s] - S.x[s]);
                STAMP(4);
                int jj = bpos;
                for (;;) {
                    jj = __builtin_amdgcn_readfirstlane(jj);
                    const int pp = __builtin_amdgcn_readfirstlane(S.p);
                    // ---- position jj leaves the passive set: Givens rotations on adjacent rows of M (column jj
                    // removed) that annihilate m = M[:, jj]; coefficients from the prefix norms of m
                    // Round 4: the rotation coefficients travel through LDS (one broadcast ds_read_b128 per row instead of
                    // four v_readlane), q is rotated in closed form (the carried combination of q is a prefix sum:
                    // carq_i = sum_{k = jj .. i} m_k q_k / a_i), row masks are scalar (no v_cmp), shifts are DPP moves;
                    // batches of rows that lie wholly in LDS or wholly in the slab carry no per-row branch.
                    double mv[kPS], pre[kPS], car[kPS];
                    double carry = 0;
                    wave_sync();
                    int cofs[kPS];  // column read of the next row: column jj dropped
                    {
                        const int jbase = moff(jj);
#pragma unroll
                        for (int s = 0; s < kPS; ++s) {  // column jj (rows jj ..) and row jj (columns < jj): both reads in flight
                            const int i = lane + kW * s;
                            mv[s] = (kW * s < pp && i >= jj && i < pp) ? (i < kLdsM ? M.l[i < kLdsM ? moff(i) + jj : 0] : M.g[CK(moff(i) + jj, kMSlab, 9, i)]) : 0.0;
                            car[s] = (i < jj) ? M.ld(jj, jbase + i) : 0.0;
                            cofs[s] = i < jj ? i : i + 1;
                        }
                    }
                    double tq[kPS];  // prefix sums of m_k q_k
                    {
                        double carry2 = 0;
#pragma unroll
                        for (int s = 0; s < kPS; ++s) {
                            pre[s] = carry;
                            tq[s] = carry2;
                            if (kW * s < pp) {
                                const double sc = wave_incl_scan(mv[s] * mv[s]);
                                pre[s] = sc + carry;
                                carry += rl(sc, 63);
                                const double sq = wave_incl_scan(mv[s] * S.q[s]);
                                tq[s] = sq + carry2;
                                carry2 += rl(sq, 63);
                            }
                        }
                    }
                    double mnext[kPS], prenext[kPS], qsh[kPS];
                    shift_down_dpp(mv, mnext);
                    shift_down_dpp(pre, prenext);
                    shift_down_dpp(S.q, qsh);  // qsh[i] = q[i + 1]
                    const int bin_out = get_at_i<kPS>(S.pidx, jj);
                    double2 *cf = reinterpret_cast<double2 *>(xbuf);  // (c_i, s_i) by position: xbuf is free during a removal
                    lds_order();
#pragma unroll
                    for (int s = 0; s < kPS; ++s) {
                        const int i = lane + kW * s;
                        if (kW * s < pp) {  // wave uniform
                            double c_ = 1.0, s_ = 0.0;
                            if (i >= jj && i < pp - 1) {
                                // a = sqrt(pre) (the first carried value keeps its sign), r = sqrt(prenext): c = m_next / r, s = a / r
                                const double ir = prenext[s] > 0 ? rsqrt_nr(prenext[s]) : 0.0;
                                const double ip = pre[s] > 0 ? rsqrt_nr(pre[s]) : 0.0;
                                const double a = (i == jj) ? mv[s] : pre[s] * ip;
                                if (prenext[s] > 0) {
                                    c_ = mnext[s] * ir;
                                    s_ = a * ir;
                                }
                                const double cq = (i == jj) ? S.q[s] : tq[s] * ip;
                                S.q[s] = c_ * cq - s_ * qsh[s];
                            }
                            cf[i] = double2{c_, s_};
                        }
                    }
                    lds_order();
                    STAMP(11);
                    COUNT(3, 1);
                    COUNT(4, pp - 1 - jj);
                    {
                        // row i of the new factor from the carried combination and old row i + 1 (column jj dropped); the
                        // loads of a batch of rows and their coefficients are in flight before its first rotation
                        auto rows = [&](int i, auto T, auto NB, auto RG) {
                            constexpr int si = decltype(T)::value;
                            constexpr int nb = decltype(NB)::value;
                            constexpr int rg = decltype(RG)::value;  // 0: rows i .. i + nb in LDS, 1: all in the slab, 2: decided per row
                            double nx[nb][si + 1];
                            double2 c2[nb];
                            // moff(i) .. moff(i + nb): lane r works out moff(i + r) -- six vector instructions for the batch and a v_readlane per
                            // row instead of seven scalar instructions per row (a scalar instruction costs what a vector instruction costs)
                            int mo[nb + 1];
                            moff_batch<nb + 1>(i, lane, mo);
#pragma unroll
                            for (int r = 0; r < nb; ++r) {
                                const int nbase = mo[r + 1];
#pragma unroll
                                for (int s = 0; s <= si; ++s) {
                                    const int idx = nbase + cofs[s];
                                    nx[r][s] = rg == 0 ? M.l[idx] : (rg == 1 ? M.g[CK(idx, kMSlab, 10, i + r)] : M.ld(i + r + 1, idx));
                                }
                            }
#pragma unroll
                            for (int r = 0; r < nb; ++r) c2[r] = cf[i + r];
                            // row i + r ends at position i + r: only its LAST register slot is masked (lanes <= i + r - 64 si), and the mask of the
                            // next row is this one shifted by a lane -- two scalar instructions per row instead of six
                            unsigned long long rm = lanes_le_nn(i - kW * si);
#pragma unroll
                            for (int r = 0; r < nb; ++r) {
                                const double c_ = c2[r].x, s_ = c2[r].y;
                                const int obase = mo[r];
                                auto rot = [&](int s) {
                                    const double outv = c_ * car[s] - s_ * nx[r][s];
                                    // car = fma(s, car, c nx) IN PLACE (the three-operand form): the compiler takes v_fmac into the product's register
                                    // and copies the result back, because the update sits under an EXEC mask
                                    const double cn = c_ * nx[r][s];
                                    asm("v_fma_f64 %0, %1, %0, %2" : "+v"(car[s]) : "v"(s_), "v"(cn));
                                    const int oidx = obase + lane + kW * s;
                                    if (rg == 0)
                                        M.l[oidx] = outv;
                                    else if (rg == 1)
                                        M.g[CK(oidx, kMSlab, 11, i + r)] = outv;
                                    else
                                        M.st(i + r, oidx, outv);
                                };
#pragma unroll
                                for (int s = 0; s < si; ++s) rot(s);
                                if (__builtin_amdgcn_inverse_ballot_w64(rm)) rot(si);
                                rm = (rm << 1) | 1ull;
                            }
                        };
                        const int hi = pp - 1;
                        const int e = hi < kW ? hi : kW;
                        int i = jj;
                        while (i < e) {  // rows < 64: one register slot
                            const int left = e - i;
                            if (i >= kLdsM) {
                                if (left >= 8) {
                                    rows(i, SlotTag<0>{}, SlotTag<8>{}, SlotTag<1>{});
                                    i += 8;
                                } else if (left >= 4) {
                                    rows(i, SlotTag<0>{}, SlotTag<4>{}, SlotTag<1>{});
                                    i += 4;
                                } else {
                                    rows(i, SlotTag<0>{}, SlotTag<1>{}, SlotTag<1>{});
                                    i += 1;
                                }
                            } else if (left >= 8 && i + 8 < kLdsM) {
                                rows(i, SlotTag<0>{}, SlotTag<8>{}, SlotTag<0>{});
                                i += 8;
                            } else if (left >= 4 && i + 4 < kLdsM) {
                                rows(i, SlotTag<0>{}, SlotTag<4>{}, SlotTag<0>{});
                                i += 4;
                            } else if (i + 1 < kLdsM) {
                                rows(i, SlotTag<0>{}, SlotTag<1>{}, SlotTag<0>{});
                                i += 1;
                            } else {
                                rows(i, SlotTag<0>{}, SlotTag<1>{}, SlotTag<2>{});
                                i += 1;
                            }
                        }
                        auto upper = [&](auto T) {  // rows kW s .. kW s + 63: s + 1 register slots per row, all in the slab
                            constexpr int sl = decltype(T)::value;
                            if (hi > kW * sl) {
                                const int end = hi < kW * (sl + 1) ? hi : kW * (sl + 1);
                                int k = jj > kW * sl ? jj : kW * sl;
                                for (; k + 4 <= end; k += 4) rows(k, T, SlotTag<4>{}, SlotTag<1>{});
                                for (; k < end; ++k) rows(k, T, SlotTag<1>{}, SlotTag<1>{});
                            }
                        };
                        upper(SlotTag<1>{});
                        if constexpr (kPS > 2) {
                            upper(SlotTag<2>{});
                            upper(SlotTag<3>{});
                        }
                    }
                    // ---- drop position jj from x / pidx
                    {
                        double xsh[kPS];
                        int psh[kPS];
                        shift_down_dpp(S.x, xsh);
                        shift_down_dpp_i(S.pidx, psh);
#pragma unroll
                        for (int s = 0; s < kPS; ++s) {
                            const int i = lane + kW * s;
                            if (i >= jj) {  // also the positions behind the passive set: they hold zeros (x) / stale bins, and so does what moves in
                                S.x[s] = xsh[s];
                                S.pidx[s] = psh[s];
                            }
                        }
                        clear_passive(S, bin_out);
                    }
                    if (pp - 1 < kLdsM) {  // the vacated last row: LDS rows >= p of M stay zero (the block sweeps mask global rows only)
                        const int vbase = moff(pp - 1);
                        const int width = 8 * (((pp - 1) >> 3) + 1);
#pragma unroll
                        for (int s = 0; s < kPS; ++s)
                            if (lane + kW * s < width) M.st(pp - 1, vbase + lane + kW * s, 0.0);
                    }
                    S.p = pp - 1;
                    // ---- round-off clean-up: any remaining x <= 0 leaves too (first position first)
                    int bad;
                    {
                        const int pn = pp - 1;
                        bad = kNone;
#pragma unroll
                        for (int s = kPS - 1; s >= 0; --s) {
                            const unsigned long long bs = pn > kW * s ? (__ballot(S.x[s] <= 0) & lanes_le_nn(pn - 1 - kW * s)) : 0ull;
                            if (bs) bad = kW * s + __ffsll(bs) - 1;
                        }
                    }
                    if (bad == kNone) break;
                    jj = bad;
                }
                STAMP(5);
                // ---- z = M^T q
                {
                    const int nI = (S.p + 7) >> 3;
                    const int lm = fresh(lane);
                    if (nI <= 2)
                        mt_times_q<2>(M, xbuf, lm, lm >> 3, lm & 7, S);
                    else if (nI <= 4)
                        mt_times_q<4>(M, xbuf, lm, lm >> 3, lm & 7, S);
                    else if (nI <= 6 || kNIMax == 6)
                        mt_times_q<6>(M, xbuf, lm, lm >> 3, lm & 7, S);
                    else
                        mt_times_q<kNIMax>(M, xbuf, lm, lm >> 3, lm & 7, S);
                }
                STAMP(6);
            }
        }
        if (jrej < 0) break;
        // a rejected column: its passive flag is set for now, its bin is remembered, and the dual is evaluated again (same state, same
        // values; the flag masks the column)
        {
            COUNT(2, 1);
#ifdef PNX_NNLS_STAMP
            if (lane == 0) atomicAdd(&g_blk_rejects, 1ULL);
#endif
            lds_int *rejlist = reinterpret_cast<lds_int *>(xbuf + 2 + kNnlsMaxBins + 2);
            const int nr = __builtin_amdgcn_readfirstlane(S.nrej);
            if (nr >= kMaxRej) {  // more rejections in one outer iteration than the list holds: the general kernel
                status = kBail;
                break;
            }
            if (lane == 0) rejlist[nr] = jrej;
            S.nrej = nr + 1;
            set_passive(S, jrej);
            jrej = -1;
        }
        }
        if (status == kDone) status = 1;
        STAMP(7);

        // ---- the next voxel starts from an all-zero LDS part of M
        {
            const int pe = __builtin_amdgcn_readfirstlane(S.p) < kLdsM ? __builtin_amdgcn_readfirstlane(S.p) : kLdsM;
            if (pe > 0) {  // the LDS rows only: the whole area in 16-byte stores (five per lane) instead of a store per row
                const double2 zero2 = {0.0, 0.0};
                for (int e = 2 * lane; e < kLdsMDoubles; e += 2 * kW) *reinterpret_cast<double2 *>((double *)M.l + e) = zero2;
            }
        }
        // ---- outputs: x by bin, rnorm = || [B; reg] x - [y; 0] ||_2 evaluated directly
        double xb[kSlots] = {0, 0, 0, 0};
        double rn;
        if (status == 1) {
            double tt = 0;
            KArgs *K = kargs();
            const double rc[5] = {K->rc[0], K->rc[1], K->rc[2], K->rc[3], K->rc[4]};
            const int pf = __builtin_amdgcn_readfirstlane(S.p);
            lds_order();
            stage_bins(xbuf, ps, pf, lane, S.x, S.pidx);
            const Win wl = load_win(xbuf + 2 + 2 * lane), wh = load_win(xbuf + 130 + 2 * lane);
            double t[kSlots];
            const double bx = bx_gather(Bl, xbuf, ps, pf, lane, [&]() { band_eval4<false>(K->rhb, wl, wh, rc, t); });
            lds_order();
#pragma unroll
            for (int s = 0; s < kSlots; ++s)
                if (binof(lane, s) < n) tt = fma(t[s], t[s], tt);  // rows >= n of R do not exist
            xb[0] = wl.mid.x;
            xb[1] = wl.mid.y;
            xb[2] = wh.mid.x;
            xb[3] = wh.mid.y;
            const double r = lane < kBMeas ? yreg - bx : 0.0;
            rn = sqrt(wave_sum(fma(r, r, tt)));
        } else
            rn = sqrt(wave_sum(lane < kBMeas ? yreg * yreg : 0.0));  // reference failure path: zeros, ||y_ext|| (nnls_solver.py:205-210)
        {
            KArgs *K = kargs();
            if (status == kBail) {  // nothing is written: the general kernel solves this voxel from scratch
                if (lane == 0) K->bail[atomicAdd(K->n_bail, 1)] = (int32_t)(K->vox_base + vox);
            } else {
                double *cv = K->coeff + (size_t)vox * n;
#pragma unroll
                for (int s = 0; s < kSlots; ++s) {
                    const int j = binof(lane, s);
                    if (j < n) cv[j] = xb[s];
                }
                if (lane == 0) {
                    K->rnorm[vox] = rn;
                    if (K->status) K->status[vox] = (int8_t)status;
                    if (K->iters) K->iters[vox] = iteration;
                }
            }
        }
        STAMP(8);
    }
#ifdef PNX_NNLS_STAMP
    if (threadIdx.x == 0 && blockIdx.x == 7)
        printf("STAMP setup=%llu dual_bx=%llu dual_bt=%llu dual_reg=%llu cand+append=%llu sync=%llu alpha=%llu removal_head=%llu removal_rows=%llu mtq=%llu tail=%llu out=%llu\n", seg[0], seg[9], seg[10], seg[1], seg[2], seg[3], seg[4], seg[11], seg[5], seg[6], seg[7], seg[8]);
    if (threadIdx.x == 0 && blockIdx.x == 0) printf("REJECTS so far=%llu\n", g_blk_rejects);
    if (threadIdx.x == 0 && blockIdx.x == 7)
        printf("COUNT voxels=%lld outer=%lld rejects=%lld removals=%lld rot_rows=%lld sum_p=%lld p_gt48=%lld p_gt64=%lld\n", cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7]);
    if (threadIdx.x == 0 && blockIdx.x == 7)
        printf("PHIST p=0:%lld 1-4:%lld 5-8:%lld 9-12:%lld 13-16:%lld 17-24:%lld 25-32:%lld\n", cnt[8], cnt[9], cnt[10], cnt[11], cnt[12], cnt[13], cnt[14]);
#endif
}
__global__ void __launch_bounds__(kBlkWaves *kW) PNX_BLK_KERNEL(const BlkArgs) { blk_body<false>(); }  // the arguments are read through kargs()
#if PNX_BLK_PS == 2
__global__ void __launch_bounds__(kBlkWaves *kW) PNX_BLK_KERNEL_HOOK(const BlkArgs) { blk_body<true>(); }  // with the test hook (tests only)
#endif
// what the host side needs to know about this instantiation
struct Variant {
    typedef BlkArgs Args;
    static constexpr int waves = kBlkWaves, mslab = kMSlab, max_pos = kMaxPos;
    static const void *kernel() { return (const void *)PNX_BLK_KERNEL; }
#if PNX_BLK_PS == 2
    static const void *kernel_hook() { return (const void *)PNX_BLK_KERNEL_HOOK; }
#else
    static const void *kernel_hook() { return nullptr; }  // the hand-over target runs without the hook
#endif
    static void launch(dim3 grid, size_t lds, hipStream_t stream, const BlkArgs &a) {
#if PNX_BLK_PS == 2
        if (a.test_rej_k > 0) {
            hipLaunchKernelGGL(PNX_BLK_KERNEL_HOOK, grid, dim3(kBlkWaves * kW), lds, stream, a);
            return;
        }
#endif
        hipLaunchKernelGGL(PNX_BLK_KERNEL, grid, dim3(kBlkWaves * kW), lds, stream, a);
    }
    static size_t lds_bytes() { return sizeof(double) * ((size_t)kBMeas * kBStride + (size_t)kBlkWaves * (kScr + kLdsMDoubles)); }
};

}  // namespace PNX_BLK_NS
}  // namespace pnx
