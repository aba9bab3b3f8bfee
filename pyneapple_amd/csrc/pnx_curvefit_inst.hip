// pnx_curvefit_inst.hip -- explicit instantiations of the TRF kernel for ONE model (compile with
// -DPNX_MODEL=<0..6>); one translation unit per model keeps the build parallel.
#include "pnx_curvefit_kernel.hpp"
#include "pnx_internal.hpp"

#ifndef PNX_MODEL
#error "compile with -DPNX_MODEL=<0..6>"
#endif

namespace pnx {

template <int N> static int launch_pcov(const CurvefitArgs &args, const ColPerm &cp, hipStream_t stream) {
    const int pb = 256;
    hipLaunchKernelGGL(pcov_kernel<N>, dim3((unsigned)((args.n_vox + pb - 1) / pb)), dim3(pb), 0, stream, args.pcov,
                       (const int8_t *)args.status, (const double *)args.cost, args.n_vox, args.n_b, cp, args.absolute_sigma);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(PNX_ERR_HIP, "pcov launch: %s", hipGetErrorString(e));
    return PNX_OK;
}

template <int MODEL, int N, bool FD, bool PV, bool T1, bool STREAM = false>
static int launch_one(const CurvefitArgs &args, int device_cus, hipStream_t stream) {
    ColPerm cp;
    constexpr int NP = Model<MODEL>::NALL + (T1 ? 1 : 0);
    for (int k = 0; k < kMaxP; ++k) cp.p[k] = (N == NP && k < Model<MODEL>::NALL) ? colperm<MODEL>(k) : k;
    if (args.phase == 2) return args.pcov ? launch_pcov<N>(args, cp, stream) : PNX_OK;
    auto kern = curvefit_kernel<MODEL, N, FD, PV, T1, STREAM>;
    // LDS per block: b-value table + per wave: [n_b][64] signal tile, parked R factor and singular vectors.
    // Up to 4 waves per block; fewer when that would not fit 160 KiB.
    int waves = PNX_CF_BLOCK_WAVES;
    auto bytes = [&](int w) { return sizeof(double) * ((args.use_sigma ? 2 * kMaxB : kMaxB) + (size_t)w * Park<N>::per_wave(args.n_b)); };  // + the 1 / sigma table
    while (waves > 1 && bytes(waves) > 160 * 1024) --waves;
    if (bytes(waves) > 160 * 1024) return set_error(PNX_ERR_UNSUPPORTED, "n_b=%d does not fit the LDS tile", args.n_b);
    const int block = waves * kWave;
    const size_t shmem = bytes(waves);
    static bool attr_done[64] = {false};  // function attributes are per device
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool &attr_set = attr_done[cur_dev & 63];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return set_error(PNX_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, block, shmem);
    if (e != hipSuccess || occ < 1) return set_error(PNX_ERR_HIP, "occupancy query failed (shmem=%zu): %s", shmem, hipGetErrorString(e));
    // A streamed launch shares the chip with its own uploads and downloads (the runtime's copy kernels), the covariance
    // epilogue and the float32 conversions, all of which must become resident WHILE it runs -- the kernel waits for their
    // data.  One block per CU leaves at least 132 of the 512 registers per lane free on every SIMD (two blocks of the
    // 246-register mono kernel would leave 20, and the uploads would wait for the kernel that waits for them).
    if (STREAM && occ > 1) occ = 1;
    long long want = (args.n_vox + block - 1) / block;
    long long cap = (long long)occ * device_cus;
    int grid = (int)(want < cap ? want : cap);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), shmem, stream, args);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(PNX_ERR_HIP, "curvefit launch: %s", hipGetErrorString(e));
    if (args.pcov && args.phase == 0) return launch_pcov<N>(args, cp, stream);
    return PNX_OK;
}

template <int MODEL, int N, bool FD, bool T1> static int launch_pv(const CurvefitArgs &args, int cus, hipStream_t st) {
    if (args.ctl && args.phase != 2) {
        // streamed launch (host-array calls): shared p0 / bounds with any set of fixed parameters, per-voxel p0 / bounds with
        // every parameter free (the combination of both stays with the chunk ring: it would double the build once more).
        // Not for the instantiations that need (almost) all 512 registers of a lane -- six or seven free parameters, or four
        // and more with the T1 factor: 402-512 -- because the runtime's copy kernels, the conversions and the epilogue must
        // become resident beside the fit.  (Leaving CUs free does not help below one per shader engine: with 8 of 256 CUs
        // free the upload still sat behind the kernel until its poll limit, with 64 it ran; profiles/stream_tight_probe.py.)
        constexpr int NP = Model<MODEL>::NALL + (T1 ? 1 : 0);
        constexpr bool kTight = (N >= 6) || (N >= 4 && T1);
        if constexpr (kTight) {
            return set_error(PNX_ERR_UNSUPPORTED, "streamed launch: %d free parameters%s leave no registers for the copies", N, T1 ? " + T1" : "");
        } else {
            if (!args.per_voxel) return launch_one<MODEL, N, FD, false, T1, true>(args, cus, st);
            if constexpr (N == NP) return launch_one<MODEL, N, FD, true, T1, true>(args, cus, st);
            return set_error(PNX_ERR_UNSUPPORTED, "streamed launch: per-voxel p0 / bounds together with fixed parameters");
        }
    }
    return args.per_voxel ? launch_one<MODEL, N, FD, true, T1>(args, cus, st) : launch_one<MODEL, N, FD, false, T1>(args, cus, st);
}

// Built per model: all parameters free (FD or analytic Jacobian) and every proper subset size of fixed parameters
// (N = 1 .. NP - 1 free, analytic Jacobian like the reference: curvefit.py:274-288, models/base.py:145-230 allow any
// subset), each without and with the T1 / STEAM factor.  WHICH parameters are free is run-time data (free_idx / fixed_idx).
template <int MODEL, bool T1, int N> static int launch_fixed(int n_free, const CurvefitArgs &args, int cus, hipStream_t st) {
    if constexpr (N >= 1) {
        if (n_free == N) return launch_pv<MODEL, N, false, T1>(args, cus, st);
        return launch_fixed<MODEL, T1, N - 1>(n_free, args, cus, st);
    } else {
        return set_error(PNX_ERR_INVALID, "model %d%s: %d free parameters", MODEL, T1 ? "+T1" : "", n_free);
    }
}

template <int MODEL, bool T1> static int launch_t1(int n_free, int jac_mode, const CurvefitArgs &args, int cus, hipStream_t st) {
    constexpr int NP = Model<MODEL>::NALL + (T1 ? 1 : 0);
    if (n_free == NP) {
        if (jac_mode == PNX_JAC_FD) return launch_pv<MODEL, NP, true, T1>(args, cus, st);
        return launch_pv<MODEL, NP, false, T1>(args, cus, st);
    }
    return launch_fixed<MODEL, T1, NP - 1>(n_free, args, cus, st);
}

template <int MODEL> static int launch_model(int n_free, int jac_mode, const CurvefitArgs &args, int cus, hipStream_t st) {
    return args.t1_mode ? launch_t1<MODEL, true>(n_free, jac_mode, args, cus, st)
                        : launch_t1<MODEL, false>(n_free, jac_mode, args, cus, st);
}

}  // namespace pnx

#define PNX_CAT2(a, b) a##b
#define PNX_CAT(a, b) PNX_CAT2(a, b)
extern "C" int PNX_CAT(pnx_launch_curvefit_m, PNX_MODEL)(int n_free, int jac_mode, const pnx::CurvefitArgs *args, int cus,
                                                          void *stream) {
    return pnx::launch_model<PNX_MODEL>(n_free, jac_mode, *args, cus, (hipStream_t)stream);
}
