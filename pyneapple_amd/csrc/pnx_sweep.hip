// pnx_sweep.hip -- residual / Jacobian / normal-equation sweep at given parameters (one LM inner-loop
// pass as a standalone, HBM-streaming kernel), fp32 and fp64.
//
// For every voxel: read the signal row (n_b values) and the parameter vector (n_all), write
//   cost = 0.5 * ||model - y||^2,   g = J^T r  (n_all),   upper triangle of J^T J  (n_all (n_all + 1) / 2)
// with the analytic Jacobian of the reference's models (models/{monoexp,biexp,triexp}.py jacobian()).
// Algorithmic bytes per voxel-sweep: (n_b + n_all) in, (n_tri + n_all + 1) out, times sizeof(T);
// triexp fp32: (32 + 5 + 15 + 5 + 1) * 4 = 232 B.
//
// Mapping: one lane per voxel; a wave stages its 64 x n_b signal tile through LDS with fully coalesced
// 16-byte loads (the (n_vox, n_b) matrix is row-major, a lane-per-row global read would touch 64 cache lines per
// instruction), rows padded by one element so the per-lane row reads are bank-conflict free; parameters and
// all outputs are parameter-major (k, n_vox), i.e. coalesced across the wavefront axis.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnx_curvefit_kernel.hpp"
#include "pnx_internal.hpp"

namespace pnx {

template <typename T> struct SweepArgs {
    const T *y;       // (n_vox, n_b)
    const T *params;  // (n_all, n_vox)
    T *cost;          // (n_vox)
    T *g;             // (n_all, n_vox)
    T *jtj;           // (n_tri, n_vox)
    long long n_vox;
    long long v_first;  // generic kernel: first voxel to process (the tiles before it belong to the full-tile kernel)
    int n_b;
    T b[kMaxB];
};

template <typename T> __device__ inline T fast_exp(T x);
template <> __device__ inline float fast_exp<float>(float x) { return __expf(x); }  // v_exp_f32 (native 2^x)
template <> __device__ inline double fast_exp<double>(double x) { return exp(x); }

// Model<MODEL> works on doubles; a thin generic restatement of signal/jac for T (same formulas)
template <int MODEL, typename T> struct ModelT {
    using M = Model<MODEL>;
    static constexpr int NALL = M::NALL, NC = M::NC;
    __device__ static void eval(const T *p, T bb, T &sig, T *ja) {
        T E[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) E[c] = fast_exp<T>(-bb * p[M::dpos(c)]);
        if constexpr (MODEL == 0) {
            sig = p[0] * E[0];
            ja[0] = E[0];
            ja[1] = -bb * p[0] * E[0];
        } else if constexpr (MODEL == 1) {
            sig = p[0] * E[0] + (1 - p[0]) * E[1];
            ja[0] = E[0] - E[1];
            ja[1] = -bb * p[0] * E[0];
            ja[2] = -bb * (1 - p[0]) * E[1];
        } else if constexpr (MODEL == 2) {
            const T inner = p[0] * E[0] + (1 - p[0]) * E[1];
            sig = p[3] * inner;
            ja[0] = p[3] * (E[0] - E[1]);
            ja[1] = -bb * p[3] * p[0] * E[0];
            ja[2] = -bb * p[3] * (1 - p[0]) * E[1];
            ja[3] = inner;
        } else if constexpr (MODEL == 3) {
            sig = p[0] * E[0] + p[2] * E[1];
            ja[0] = E[0];
            ja[1] = -bb * p[0] * E[0];
            ja[2] = E[1];
            ja[3] = -bb * p[2] * E[1];
        } else if constexpr (MODEL == 4) {
            const T f3 = 1 - p[0] - p[2];
            sig = p[0] * E[0] + p[2] * E[1] + f3 * E[2];
            ja[0] = E[0] - E[2];
            ja[1] = -bb * p[0] * E[0];
            ja[2] = E[1] - E[2];
            ja[3] = -bb * p[2] * E[1];
            ja[4] = -bb * f3 * E[2];
        } else if constexpr (MODEL == 5) {
            const T f3 = 1 - p[0] - p[2];
            const T inner = p[0] * E[0] + p[2] * E[1] + f3 * E[2];
            sig = p[5] * inner;
            ja[0] = p[5] * (E[0] - E[2]);
            ja[1] = -bb * p[5] * p[0] * E[0];
            ja[2] = p[5] * (E[1] - E[2]);
            ja[3] = -bb * p[5] * p[2] * E[1];
            ja[4] = -bb * p[5] * f3 * E[2];
            ja[5] = inner;
        } else {
            sig = p[0] * E[0] + p[2] * E[1] + p[4] * E[2];
            ja[0] = E[0];
            ja[1] = -bb * p[0] * E[0];
            ja[2] = E[1];
            ja[3] = -bb * p[2] * E[1];
            ja[4] = E[2];
            ja[5] = -bb * p[4] * E[2];
        }
    }
};

typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT> __device__ inline float4 ld16(const float4 *p) {
    if constexpr (NT & 1) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <int NT, typename T> __device__ inline T ld_in(const T *p) {
    if constexpr (NT & 4) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT, typename T> __device__ inline void st_out(T *p, T v) {
    if constexpr (NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int MODEL, typename T, int NT>
__global__ void __launch_bounds__(256) sweep_kernel(const SweepArgs<T> A) {
    using MT = ModelT<MODEL, T>;
    constexpr int NALL = MT::NALL;
    constexpr int NTRI = NALL * (NALL + 1) / 2;
    constexpr int VEC = 16 / sizeof(T);  // elements per 16-byte load
    extern __shared__ unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    const int n_b = A.n_b;
    const int stride = n_b + 1;  // padded row: lane-per-row reads hit 64 different banks
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T *tile = smem + kMaxB + (size_t)wave * kWave * stride;
    T *bsh = smem;
    for (int i = threadIdx.x; i < n_b; i += blockDim.x) bsh[i] = A.b[i];
    __syncthreads();
    const long long n_tiles = (A.n_vox + kWave - 1) / kWave;
    const long long tiles_per_block = blockDim.x >> 6;
    const long long tstep = (long long)gridDim.x * tiles_per_block;
    const long long t_first = A.v_first / kWave;
    const bool vec_ok = (n_b % VEC) == 0 && (reinterpret_cast<uintptr_t>(A.y) % 16) == 0;
    constexpr int PF = 8;  // 16-byte chunks per lane held in registers for the NEXT tile (8 KiB per wave)
    const bool pow2 = (n_b & (n_b - 1)) == 0;
    const int sh = __ffs(n_b) - 1;  // e / n_b as a shift when n_b is a power of two (no integer division)
    float4 pre[PF];
    long long t = t_first + (long long)blockIdx.x * tiles_per_block + wave;
    if (vec_ok && t < n_tiles) {
        const long long v0 = t * kWave;
        const int nv = (A.n_vox - v0) < kWave ? (int)(A.n_vox - v0) : kWave;
        const float4 *src4 = reinterpret_cast<const float4 *>(A.y + (size_t)v0 * n_b);
        const int total4 = nv * n_b / VEC;
#pragma unroll
        for (int c = 0; c < PF; ++c)
            if (lane + kWave * c < total4) pre[c] = ld16<NT>(src4 + lane + kWave * c);
    }
    for (; t < n_tiles; t += tstep) {
        const long long v0 = t * kWave;
        const int nv = (A.n_vox - v0) < kWave ? (int)(A.n_vox - v0) : kWave;
        // ---- the tile is one contiguous block of nv * n_b elements: coalesced 16-byte loads (already in flight /
        // landed in `pre` for the first PF chunks), scattered into the padded LDS rows
        const T *src = A.y + (size_t)v0 * n_b;
        const int total = nv * n_b;
        if (vec_ok) {
#pragma unroll
            for (int c = 0; c < PF; ++c) {
                const int e = (lane + kWave * c) * VEC;
                if (e < total) {
                    T tmp[VEC];
                    *reinterpret_cast<float4 *>(tmp) = pre[c];
                    const int v = pow2 ? (e >> sh) : e / n_b, i = e - v * n_b;
#pragma unroll
                    for (int u = 0; u < VEC; ++u) tile[v * stride + i + u] = tmp[u];
                }
            }
            for (int e = (lane + kWave * PF) * VEC; e < total; e += kWave * VEC) {
                T tmp[VEC];
                *reinterpret_cast<float4 *>(tmp) = ld16<NT>(reinterpret_cast<const float4 *>(src + e));
                const int v = e / n_b, i = e - v * n_b;
#pragma unroll
                for (int u = 0; u < VEC; ++u) tile[v * stride + i + u] = tmp[u];
            }
            // next tile's loads fly while this one is computed
            const long long tn = t + tstep;
            if (tn < n_tiles) {
                const long long v0n = tn * kWave;
                const int nvn = (A.n_vox - v0n) < kWave ? (int)(A.n_vox - v0n) : kWave;
                const float4 *src4 = reinterpret_cast<const float4 *>(A.y + (size_t)v0n * n_b);
                const int total4 = nvn * n_b / VEC;
#pragma unroll
                for (int c = 0; c < PF; ++c)
                    if (lane + kWave * c < total4) pre[c] = ld16<NT>(src4 + lane + kWave * c);
            }
        } else {
            for (int e = lane; e < total; e += kWave) {
                const int v = e / n_b, i = e - v * n_b;
                tile[v * stride + i] = src[e];
            }
        }
        const long long vox = v0 + lane;
        const bool live = lane < nv;
        T p[NALL];
#pragma unroll
        for (int k = 0; k < NALL; ++k) p[k] = live ? ld_in<NT>(A.params + (size_t)k * A.n_vox + vox) : T(0);
        T cost = 0, g[NALL], H[NTRI];
#pragma unroll
        for (int k = 0; k < NALL; ++k) g[k] = 0;
#pragma unroll
        for (int k = 0; k < NTRI; ++k) H[k] = 0;
        const T *row = tile + lane * stride;
#pragma unroll 4
        for (int i = 0; i < n_b; ++i) {
            T sig, ja[NALL];
            MT::eval(p, bsh[i], sig, ja);
            const T r = sig - row[i];
            cost += r * r;
            int q = 0;
#pragma unroll
            for (int a = 0; a < NALL; ++a) {
                g[a] += ja[a] * r;
#pragma unroll
                for (int c = a; c < NALL; ++c) H[q++] += ja[a] * ja[c];
            }
        }
        if (live) {
            st_out<NT>(A.cost + vox, T(0.5) * cost);
#pragma unroll
            for (int k = 0; k < NALL; ++k) st_out<NT>(A.g + (size_t)k * A.n_vox + vox, g[k]);
#pragma unroll
            for (int k = 0; k < NTRI; ++k) st_out<NT>(A.jtj + (size_t)k * A.n_vox + vox, H[k]);
        }
    }
}


#ifndef PNX_SWEEP_PACKED
#define PNX_SWEEP_PACKED 1
#endif
typedef float v2f __attribute__((ext_vector_type(2)));

// Row pass of the fp32 triexp (reduced) sweep with the instruction count cut for the VALU: at 45 instructions per row the
// pass costs more issue time than its 232 B per voxel cost HBM time.  Here: diffusivities pre-multiplied by -log2(e)
// (v_exp_f32 is 2^x), b-values straight from the kernel argument (scalar registers, no LDS reads), and the 21 accumulations
// -- the upper triangle of v v^T for v = [J_f1, J_f2, J_D1, J_D2, J_D3, r] -- as 9 packed FMAs on register pairs that the
// evaluation produces as pairs + 3 scalar ones: 26 VALU instructions per row.  Outputs in the ABI's parameter order
// [f1, D1, f2, D2, D3].
template <int NB>
__device__ __forceinline__ void tri_rows_packed(const float *p, const float *row, const float *b, float &cost, float *g, float *H) {
    constexpr float L2E = 1.4426950408889634f;
    const v2f d12 = {-p[1] * L2E, -p[3] * L2E};
    const float d3 = -p[4] * L2E;
    const v2f f12 = {p[0], p[2]};
    const float f3 = 1.0f - p[0] - p[2];
    const v2f z = {0.0f, 0.0f};
    v2f a00 = z, a02 = z, a04 = z, a12 = z, a14 = z, a22 = z, a24 = z, a34 = z, a44 = z;
    float s11 = 0.0f, s33 = 0.0f, srr = 0.0f;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float bb = b[i];
        const v2f x12 = d12 * bb;
        const float x3 = d3 * bb;
        v2f e12;
        e12.x = __builtin_amdgcn_exp2f(x12.x);
        e12.y = __builtin_amdgcn_exp2f(x12.y);
        const float e3 = __builtin_amdgcn_exp2f(x3);
        const v2f t12 = f12 * e12;
        const float t3 = f3 * e3;
        const float r = (t12.x + t12.y) + t3 - row[i];
        const v2f v01 = e12 - e3;            // J_f1, J_f2
        const v2f v23 = t12 * (-bb);         // J_D1, J_D2
        v2f v4r;
        v4r.x = t3 * (-bb);                  // J_D3
        v4r.y = r;
        a00 += v01.x * v01;
        a02 += v01.x * v23;
        a04 += v01.x * v4r;
        s11 += v01.y * v01.y;
        a12 += v01.y * v23;
        a14 += v01.y * v4r;
        a22 += v23.x * v23;
        a24 += v23.x * v4r;
        s33 += v23.y * v23.y;
        a34 += v23.y * v4r;
        a44 += v4r.x * v4r;
        srr += r * r;
    }
    // internal order [f1, f2, D1, D2, D3] -> output order [f1, D1, f2, D2, D3]; H row-major upper triangle of 5 x 5
    cost = srr;
    H[0] = a00.x;  H[2] = a00.y;            // (f1,f1) (f1,f2)
    H[1] = a02.x;  H[3] = a02.y;            // (f1,D1) (f1,D2)
    H[4] = a04.x;  g[0] = a04.y;            // (f1,D3)
    H[9] = s11;                             // (f2,f2)
    H[6] = a12.x;  H[10] = a12.y;           // (D1,f2) (f2,D2)
    H[11] = a14.x; g[2] = a14.y;            // (f2,D3)
    H[5] = a22.x;  H[7] = a22.y;            // (D1,D1) (D1,D2)
    H[8] = a24.x;  g[1] = a24.y;            // (D1,D3)
    H[12] = s33;                            // (D2,D2)
    H[13] = a34.x; g[3] = a34.y;            // (D2,D3)
    H[14] = a44.x; g[4] = a44.y;            // (D3,D3)
}


// Full-tile fast path: n_b is the compile-time NB, every tile holds 64 voxels, so the tile copy is CH = NB / VEC
// unpredicated 16-byte loads per lane.  Software pipeline per wave: the loads of tile t+1 (signal chunks AND the
// parameter vector) are issued right after tile t has been scattered into LDS and before its row loop, so nothing the
// row loop or the result stores wait on sits behind them in the in-order vmcnt queue.
template <int MODEL, typename T, int NB>
__global__ void __launch_bounds__(256) sweep_full_kernel(const SweepArgs<T> A, const long long n_full) {
    using MT = ModelT<MODEL, T>;
    constexpr int NALL = MT::NALL;
    constexpr int NTRI = NALL * (NALL + 1) / 2;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CH = NB / VEC;
    constexpr int STRIDE = NB + 1;
    constexpr int NT = 3;
    extern __shared__ unsigned char smem_raw[];
    T *smem = reinterpret_cast<T *>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T *tile = smem + kMaxB + (size_t)wave * kWave * STRIDE;
    T *bsh = smem;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) bsh[i] = A.b[i];
    __syncthreads();
    const long long tstep = (long long)gridDim.x * (blockDim.x >> 6);
    long long t = (long long)blockIdx.x * (blockDim.x >> 6) + wave;
    float4 pre[CH];
    T pn[NALL];
    if (t < n_full) {
        const float4 *src4 = reinterpret_cast<const float4 *>(A.y + (size_t)t * kWave * NB) + lane;
#pragma unroll
        for (int c = 0; c < CH; ++c) pre[c] = ld16<NT>(src4 + kWave * c);
#pragma unroll
        for (int k = 0; k < NALL; ++k) pn[k] = A.params[(size_t)k * A.n_vox + t * kWave + lane];
    }
    // drain the prologue loads here: otherwise the loop header has to merge "loads outstanding" (entry) with "stores
    // outstanding" (back edge) and the compiler settles on vmcnt(0) per iteration, i.e. waits for the previous tile's
    // result stores
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (; t < n_full; t += tstep) {
        const long long vox = t * kWave + lane;
        T p[NALL];
#pragma unroll
        for (int k = 0; k < NALL; ++k) p[k] = pn[k];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int e = (lane + kWave * c) * VEC;
            const int v = e / NB, i = e - v * NB;
            T tmp[VEC];
            *reinterpret_cast<float4 *>(tmp) = pre[c];
#pragma unroll
            for (int u = 0; u < VEC; ++u) tile[v * STRIDE + i + u] = tmp[u];
        }
        const long long tn = t + tstep;
        if (tn < n_full) {
            const float4 *src4 = reinterpret_cast<const float4 *>(A.y + (size_t)tn * kWave * NB) + lane;
#pragma unroll
            for (int c = 0; c < CH; ++c) pre[c] = ld16<NT>(src4 + kWave * c);
#pragma unroll
            for (int k = 0; k < NALL; ++k) pn[k] = A.params[(size_t)k * A.n_vox + tn * kWave + lane];
        }
        T cost = 0, g[NALL], H[NTRI];
        const T *row = tile + lane * STRIDE;
        if constexpr (MODEL == 4 && sizeof(T) == 4 && PNX_SWEEP_PACKED) {
            tri_rows_packed<NB>(p, row, A.b, cost, g, H);
        } else {
#pragma unroll
            for (int k = 0; k < NALL; ++k) g[k] = 0;
#pragma unroll
            for (int k = 0; k < NTRI; ++k) H[k] = 0;
#pragma unroll 8
            for (int i = 0; i < NB; ++i) {
                T sig, ja[NALL];
                MT::eval(p, bsh[i], sig, ja);
                const T r = sig - row[i];
                cost += r * r;
                int q = 0;
#pragma unroll
                for (int a = 0; a < NALL; ++a) {
                    g[a] += ja[a] * r;
#pragma unroll
                    for (int c = a; c < NALL; ++c) H[q++] += ja[a] * ja[c];
                }
            }
        }
        st_out<NT>(A.cost + vox, T(0.5) * cost);
#pragma unroll
        for (int k = 0; k < NALL; ++k) st_out<NT>(A.g + (size_t)k * A.n_vox + vox, g[k]);
#pragma unroll
        for (int k = 0; k < NTRI; ++k) st_out<NT>(A.jtj + (size_t)k * A.n_vox + vox, H[k]);
    }
}

template <int MODEL, typename T, int NB>
static int launch_sweep_full(const SweepArgs<T> &a, long long n_full, int cus, int bpc, hipStream_t st) {
    const int block = 256;
    const size_t shmem = sizeof(T) * (kMaxB + (size_t)(block / kWave) * kWave * (NB + 1));
    long long want = (n_full + 3) / 4, cap = (long long)cus * bpc;
    const int grid = (int)(want < cap ? want : cap);
    static bool attr_done[64] = {false};  // function attributes are per device
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool &attr_set = attr_done[cur_dev & 63];
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute((const void *)sweep_full_kernel<MODEL, T, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return set_error(PNX_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(ea));
        attr_set = true;
    }
    hipLaunchKernelGGL((sweep_full_kernel<MODEL, T, NB>), dim3(grid), dim3(block), shmem, st, a, n_full);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PNX_ERR_HIP, "sweep launch: %s", hipGetErrorString(e));
    return PNX_OK;
}

template <int MODEL, typename T, int NT>
static int launch_sweep_nt(const SweepArgs<T> &a, int cus, hipStream_t st) {
    const int block = 256;
    const size_t shmem = sizeof(T) * (kMaxB + (size_t)(block / kWave) * kWave * (a.n_b + 1));
    const long long n_tiles = (a.n_vox + kWave - 1) / kWave - a.v_first / kWave;
    long long want = (n_tiles + 3) / 4;
    static const int bpc = dev_getenv("PNX_SWEEP_BLOCKS_PER_CU") ? atoi(dev_getenv("PNX_SWEEP_BLOCKS_PER_CU")) : 32;
    long long cap = (long long)cus * bpc;  // memory-bound: ~2048 blocks, grid-stride the rest
    int grid = (int)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    static bool attr_done[64] = {false};  // function attributes are per device
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool &attr_set = attr_done[cur_dev & 63];
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute((const void *)sweep_kernel<MODEL, T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ea != hipSuccess) return set_error(PNX_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(ea));
        attr_set = true;
    }
    if (shmem > 160 * 1024) return set_error(PNX_ERR_UNSUPPORTED, "n_b=%d does not fit the LDS tile", a.n_b);
    hipLaunchKernelGGL((sweep_kernel<MODEL, T, NT>), dim3(grid), dim3(block), shmem, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PNX_ERR_HIP, "sweep launch: %s", hipGetErrorString(e));
    return PNX_OK;
}

template <int MODEL, typename T> static int launch_sweep(SweepArgs<T> a, int cus, hipStream_t st) {
    static const bool generic_only = dev_getenv("PNX_SWEEP_GENERIC") != nullptr;
    // 64 blocks of 4 waves per CU: at C3 every wave gets one tile (measured 4 / 8 / 16 / 32 / 64 / 128: 5.51 / 5.59 / 5.59 /
    // 5.52 / 5.70 / 5.74 TB/s -- flat; the hardware's own wave scheduling hides as much as the software pipeline does)
    static const int bpc = dev_getenv("PNX_SWEEP_BLOCKS_PER_CU") ? atoi(dev_getenv("PNX_SWEEP_BLOCKS_PER_CU")) : 64;
    const long long n_full = a.n_vox / kWave;
    const bool aligned = (reinterpret_cast<uintptr_t>(a.y) % 16) == 0;
    a.v_first = 0;
    if (!generic_only && aligned && n_full > 0 && (a.n_b == 16 || a.n_b == 32)) {
        const int rc = a.n_b == 32 ? launch_sweep_full<MODEL, T, 32>(a, n_full, cus, bpc, st)
                                   : launch_sweep_full<MODEL, T, 16>(a, n_full, cus, bpc, st);
        if (rc != PNX_OK) return rc;
        a.v_first = n_full * kWave;  // ragged last tile, if any, goes through the generic kernel
        if (a.v_first == a.n_vox) return PNX_OK;
    }
    return launch_sweep_nt<MODEL, T, 3>(a, cus, st);
}

template <typename T>
static int sweep_impl(int model, int64_t n_vox, int n_b, const T *b_host, const T *y, const T *params, T *out_cost,
                      T *out_g, T *out_jtj, int device, void *stream) {
    if (model < 0 || model > 6) return set_error(PNX_ERR_INVALID, "unknown model %d", model);
    if (n_b < 1 || n_b > kMaxB) return set_error(PNX_ERR_INVALID, "n_b=%d out of range", n_b);
    if (n_vox < 0 || !b_host || (n_vox && (!y || !params || !out_cost || !out_g || !out_jtj)))
        return set_error(PNX_ERR_INVALID, "NULL pointer");
    if (n_vox == 0) return PNX_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return set_error(PNX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return set_error(PNX_ERR_INVALID, "device %d out of range", device);
    if (hipSetDevice(device) != hipSuccess) return set_error(PNX_ERR_HIP, "hipSetDevice failed");
    static int cu_cache[64] = {0};
    if (device < 64 && cu_cache[device] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return set_error(PNX_ERR_HIP, "hipGetDeviceProperties failed");
        cu_cache[device] = prop.multiProcessorCount;
    }
    SweepArgs<T> a;
    a.y = y;
    a.params = params;
    a.cost = out_cost;
    a.g = out_g;
    a.jtj = out_jtj;
    a.n_vox = n_vox;
    a.n_b = n_b;
    for (int i = 0; i < n_b; ++i) a.b[i] = b_host[i];
    hipStream_t st = (hipStream_t)stream;
    const int cus = device < 64 ? cu_cache[device] : 256;
    switch (model) {
    case 0: return launch_sweep<0, T>(a, cus, st);
    case 1: return launch_sweep<1, T>(a, cus, st);
    case 2: return launch_sweep<2, T>(a, cus, st);
    case 3: return launch_sweep<3, T>(a, cus, st);
    case 4: return launch_sweep<4, T>(a, cus, st);
    case 5: return launch_sweep<5, T>(a, cus, st);
    default: return launch_sweep<6, T>(a, cus, st);
    }
}

}  // namespace pnx

extern "C" {
int pnx_sweep_f32(int model, int64_t n_vox, int n_b, const float *b_host, const float *y, const float *params,
                  float *out_cost, float *out_g, float *out_jtj, int device, void *stream) {
    return pnx::sweep_impl<float>(model, n_vox, n_b, b_host, y, params, out_cost, out_g, out_jtj, device, stream);
}
int pnx_sweep_f64(int model, int64_t n_vox, int n_b, const double *b_host, const double *y, const double *params,
                  double *out_cost, double *out_g, double *out_jtj, int device, void *stream) {
    return pnx::sweep_impl<double>(model, n_vox, n_b, b_host, y, params, out_cost, out_g, out_jtj, device, stream);
}
}
