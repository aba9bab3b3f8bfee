// pnx_nnls_dev.hpp -- device-side helpers shared by the NNLS kernels (pnx_nnls.hip: Gram form, general;
// pnx_nnls_blk.hip: basis resident in LDS, block-distributed inverse factor): wave-level reductions on DPP, EXEC-masked
// FMAs, position-indexed register vectors (position i lives in lane i & 63, slot i >> 6), the bin-ordered stencil of the
// reference's banded regularisers (model_functions/nnls.py:46-85).
#pragma once
#include <hip/hip_runtime.h>

#include "pnx_nnls.hpp"

namespace pnx {

constexpr int kW = 64;
constexpr int kSlots = kNnlsMaxBins / kW;  // 4 (the wide instantiations of the Gram- and QR-form kernels: kNnlsWideBins / kW = 8)
// v_writelane_b32 (value and lane in scalar registers): clang has no builtin of that name, the LLVM intrinsic is reached through its name
extern "C" __device__ int pnx_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
constexpr int kNone = 1 << 30;

// ---- cross-lane primitives ----------------------------------------------------------------------
// DPP (row_shr 1/2/4/8, row_bcast 15/31): a handful of VALU ops with register-file latency instead of a dozen
// ds_bpermute round trips per fp64 reduction.
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

// v_readlane of a double; `l` must be wave-uniform
__device__ inline double rl(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
// inclusive prefix sum over the 64 lanes (lane 63 holds the total)
__device__ inline double wave_incl_scan(double v) {
    v += dpp_mov<kShr1, 0xf, true>(v);
    v += dpp_mov<kShr2, 0xf, true>(v);
    v += dpp_mov<kShr4, 0xf, true>(v);
    v += dpp_mov<kShr8, 0xf, true>(v);
    v += dpp_mov<kBc15, 0xa, true>(v);
    v += dpp_mov<kBc31, 0xc, true>(v);
    return v;
}
__device__ inline double wave_sum(double v) { return rl(wave_incl_scan(v), 63); }
__device__ inline double wave_max(double v) {
    v = fmax(v, dpp_mov<kShr1, 0xf, false>(v));
    v = fmax(v, dpp_mov<kShr2, 0xf, false>(v));
    v = fmax(v, dpp_mov<kShr4, 0xf, false>(v));
    v = fmax(v, dpp_mov<kShr8, 0xf, false>(v));
    v = fmax(v, dpp_mov<kBc15, 0xa, false>(v));
    v = fmax(v, dpp_mov<kBc31, 0xc, false>(v));
    return rl(v, 63);
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

// acc += a * b on the lanes of `mask` only (wave-uniform mask).  One VALU instruction: the mask goes through EXEC on the
// scalar unit instead of a v_cmp + two v_cndmask per fp64 value.  EXEC is saved and restored inside the statement, so the
// compiler never sees it changed.
#ifndef PNX_NNLS_ASM_MASK
#define PNX_NNLS_ASM_MASK 1
#endif
__device__ inline void fma_on(double &acc, double a, double b, unsigned long long mask) {
#if PNX_NNLS_ASM_MASK
    unsigned long long keep;
    asm volatile("s_and_saveexec_b64 %1, %4\n\tv_fma_f64 %0, %2, %3, %0\n\ts_mov_b64 exec, %1"
                 : "+v"(acc), "=&s"(keep)
                 : "v"(a), "v"(b), "s"(mask)
                 : "scc");
#else
    if (__builtin_amdgcn_inverse_ballot_w64(mask)) acc = fma(a, b, acc);
#endif
}
// 1 / sqrt(a) for a wave-uniform a > 0: v_rsq_f64 (~2^-26) + two Newton steps instead of the IEEE sqrt + divide
// expansions (~60 VALU instructions per candidate column)
__device__ inline double rsqrt_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);
    const double h = 0.5 * a;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

__device__ inline void wave_sync() {
    // one wave per workgroup: "everything I wrote (LDS / my global scratch slab) is visible to my other lanes" is a wait
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// Bin ownership: lane l holds bins {2l, 2l+1, 128+2l, 128+2l+1} (and 256+2l, ... with eight bins per lane): one row of G or B is
// two (four) 16-byte loads per lane.
__device__ inline int binof(int lane, int s) { return ((s >> 1) << 7) + 2 * lane + (s & 1); }

// ---- position-indexed register vectors (position i lives in lane i & 63, slot i >> 6) --------------
template <int S> struct SlotTag { static constexpr int value = S; };
// f(i, SlotTag<S>) for every position i in [lo, hi), slot known at compile time
template <int S, int UNROLL, class F> __device__ inline void pos_range(int lo, int hi, F &&f) {
    const int a = lo > kW * S ? lo : kW * S;
    const int b = hi < kW * S + kW ? hi : kW * S + kW;
#pragma unroll UNROLL
    for (int i = a; i < b; ++i) f(i, SlotTag<S>{});
}
template <int S, int NS, int UNROLL, class F> __device__ inline void for_pos_from(int lo, int hi, F &&f) {
    pos_range<S, UNROLL>(lo, hi, f);
    if constexpr (S + 1 < NS) {
        if (hi > (S + 1) * kW) for_pos_from<S + 1, NS, UNROLL>(lo, hi, f);
    }
}
// vectors of NS slots (positions < 64 NS)
template <int NS, int UNROLL, class F> __device__ inline void for_posn(int lo, int hi, F &&f) { for_pos_from<0, NS, UNROLL>(lo, hi, f); }
template <int UNROLL, class F> __device__ inline void for_pos(int lo, int hi, F &&f) { for_pos_from<0, kSlots, UNROLL>(lo, hi, f); }
// the same in groups of four: f4(i, SlotTag<S>) covers positions i .. i + 3 (all in slot S), f1 the ragged rest.  Loops
// around v_readlane / DPP are never unrolled by the compiler, so "several rows in flight" has to be spelled out.
template <int S, class F4, class F1> __device__ inline void pos_range4(int lo, int hi, F4 &&f4, F1 &&f1) {
    const int a = lo > kW * S ? lo : kW * S;
    const int b = hi < kW * S + kW ? hi : kW * S + kW;
    int i = a;
    for (; i + 4 <= b; i += 4) f4(i, SlotTag<S>{});
    for (; i < b; ++i) f1(i, SlotTag<S>{});
}
template <int S, int NS, class F4, class F1> __device__ inline void for_pos4_from(int lo, int hi, F4 &&f4, F1 &&f1) {
    pos_range4<S>(lo, hi, f4, f1);
    if constexpr (S + 1 < NS) {
        if (hi > (S + 1) * kW) for_pos4_from<S + 1, NS>(lo, hi, f4, f1);
    }
}
// the same for vectors of NS slots (positions < 64 NS)
template <int NS, class F4, class F1> __device__ inline void for_pos4n(int lo, int hi, F4 &&f4, F1 &&f1) {
    for_pos4_from<0, NS>(lo, hi, f4, f1);
}
template <class F4, class F1> __device__ inline void for_pos4(int lo, int hi, F4 &&f4, F1 &&f1) { for_pos4_from<0, kSlots>(lo, hi, f4, f1); }
// value at the (wave-uniform) position pos of a position-indexed vector
template <int NS> __device__ inline double get_at(const double (&a)[NS], int pos) {
    double r = rl(a[0], pos & 63);
#pragma unroll
    for (int s = 1; s < NS; ++s)
        if ((pos >> 6) == s) r = rl(a[s], pos & 63);
    return r;
}
template <int NS> __device__ inline int get_at_i(const int (&a)[NS], int pos) {
    int r = __builtin_amdgcn_readlane(a[0], pos & 63);
#pragma unroll
    for (int s = 1; s < NS; ++s)
        if ((pos >> 6) == s) r = __builtin_amdgcn_readlane(a[s], pos & 63);
    return r;
}
// write v at (uniform) position pos.  Written as per-lane selects on purpose: the obvious "if ((pos >> 6) == s) a[s] = v"
// chain is folded by the optimiser into ONE store with a run-time index, which demotes the whole array from registers
// to scratch memory (and every later access to it onto the vector-memory pipe, the busiest unit of this kernel).
template <int NS> __device__ inline void put(double (&a)[NS], int pos, double v, int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = (pos == lane + kW * s) ? v : a[s];
}
template <int NS> __device__ inline void put_i(int (&a)[NS], int pos, int v, int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = (pos == lane + kW * s) ? v : a[s];
}
// position i receives the value of position i + 1 (the last position receives garbage)
template <int NS> __device__ inline void shift_down(const double (&a)[NS], double (&out)[NS], int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double v = __shfl_down(a[s], 1);
        const double nxt0 = (s + 1 < NS) ? rl(a[s + 1 < NS ? s + 1 : s], 0) : 0.0;
        out[s] = (lane == 63) ? nxt0 : v;
    }
}
template <int NS> __device__ inline void shift_down_i(const int (&a)[NS], int (&out)[NS], int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        int v = __shfl_down(a[s], 1);
        const int nxt0 = (s + 1 < NS) ? __builtin_amdgcn_readlane(a[s + 1 < NS ? s + 1 : s], 0) : 0;
        out[s] = (lane == 63) ? nxt0 : v;
    }
}

// Lanes of one wave talk through LDS without barriers (the hardware executes a wave's LDS instructions in order).  The
// COMPILER, however, reasons per thread: "I store xbuf[lane] and later load xbuf[lane + 1]: no alias, the load may move
// up".  lds_order() is a compiler-only fence (no instruction) that pins the program order of memory operations.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// t = R v and u = R^T t through the bin-ordered scratch (zero boundary: R is n x n, R[i][j] = c[j - i + 2]);
// v by position (x, pidx, p).  Returns u by bin in `u`, and sum t^2 of this lane's bins in *tt.  A lane owns the bin
// pairs (2 l, 2 l + 1) and (128 + 2 l, 129 + 2 l): three 16-byte reads per pair bring the pair and its two neighbours
// on either side.
template <bool REV, int HB> __device__ __forceinline__ void band5(const double *q, const double (&c)[5], double &o0, double &o1) {
    // q points at the pair; taps d = -2 .. 2 of out_j = sum_d c[d + 2] v[j + d]  (REV: c[2 - d], the transposed band)
    const double2 lo = *reinterpret_cast<const double2 *>(q - 2), mid = *reinterpret_cast<const double2 *>(q),
                  hi = *reinterpret_cast<const double2 *>(q + 2);
    const double cm1 = REV ? c[3] : c[1], cp1 = REV ? c[1] : c[3], cm2 = REV ? c[4] : c[0], cp2 = REV ? c[0] : c[4];
    double a = cm1 * lo.y, b = cm1 * mid.x;
    a = fma(c[2], mid.x, a);
    b = fma(c[2], mid.y, b);
    a = fma(cp1, mid.y, a);
    b = fma(cp1, hi.x, b);
    if (HB > 1) {
        a = fma(cm2, lo.x, a);
        b = fma(cm2, lo.y, b);
        a = fma(cp2, hi.x, a);
        b = fma(cp2, hi.y, b);
    }
    o0 = a;
    o1 = b;
}
template <bool WANT_U, int HB, int NS, int KS>
__device__ __forceinline__ void reg_terms_hb(double *xbuf, const double (&c)[5], int n, int p, int lane, const double (&x)[NS],
                                    const int (&pidx)[NS], double (&u)[KS], double *tt) {
    // bin b lives at xbuf[2 + b]; two zero doubles in front of bin 0 and behind the last bin (64 KS - 1: 255 with four bins per lane)
    const double2 zero2 = {0.0, 0.0};
    double *lo = xbuf + 2 + 2 * lane;  // pair h of this lane: lo + 128 h
#pragma unroll
    for (int h = 0; h < KS / 2; ++h) *reinterpret_cast<double2 *>(lo + 128 * h) = zero2;
    if (lane < 2) *reinterpret_cast<double2 *>(xbuf + (kW * KS + 2) * lane) = zero2;
    lds_order();
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (lane + kW * s < p) xbuf[2 + pidx[s]] = x[s];
    lds_order();
    double t[KS];
#pragma unroll
    for (int h = 0; h < KS / 2; ++h) band5<false, HB>(lo + 128 * h, c, t[2 * h], t[2 * h + 1]);
    double acc = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        t[s] = (binof(lane, s) < n) ? t[s] : 0.0;  // rows >= n of R do not exist
        acc = fma(t[s], t[s], acc);
    }
    if (tt) *tt = acc;
    lds_order();
    if (!WANT_U) return;
#pragma unroll
    for (int h = 0; h < KS / 2; ++h) *reinterpret_cast<double2 *>(lo + 128 * h) = double2{t[2 * h], t[2 * h + 1]};
    lds_order();
#pragma unroll
    for (int h = 0; h < KS / 2; ++h) band5<true, HB>(lo + 128 * h, c, u[2 * h], u[2 * h + 1]);  // (R^T t)_j = sum_d c[d + 2] t_{j - d}
    lds_order();
}
// hb (wave uniform): half bandwidth of the regulariser, 1 (orders 1 and 2 of the reference) or 2 (order 3)
template <bool WANT_U, int NS, int KS>
__device__ __forceinline__ void reg_terms(double *xbuf, const double (&c)[5], int hb, int n, int p, int lane, const double (&x)[NS],
                                 const int (&pidx)[NS], double (&u)[KS], double *tt) {
    if (hb > 1)
        reg_terms_hb<WANT_U, 2>(xbuf, c, n, p, lane, x, pidx, u, tt);
    else
        reg_terms_hb<WANT_U, 1>(xbuf, c, n, p, lane, x, pidx, u, tt);
}

}  // namespace pnx
