#!/usr/bin/env python3
"""bench.py -- voxels/s of the per-voxel fitting hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload triexp|biexp|mono|nnls] [--no-secondary]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one batched fit through the C ABI, device-pointer mode) over one
synthetic volume that is already resident in HBM.  Default workload: BASELINE.json configs[2]
(triexp bounded TRF, 256x256x64 voxels x 32 b-values), computed in fp64 with SciPy's 2-point finite
difference Jacobian and with the covariance output, i.e. exactly what the reference's pixelwise fitter
asks of its solver.  Voxels shard with no collective: every rank fits its own volume (weak scaling);
value = (voxels of all ranks) / (max over ranks of the timed region).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable)
METRIC = "voxels/sec (pixelwise triexp LM & 250-bin NNLS) at 1/2/4/8 MI355X"


def host_cores() -> int:
    """Threads for the CPU baseline: the box's CPU share (16 per GPU on the pool), not the host's 256 cores."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return int(os.environ.get("PNX_CPU_THREADS", min(n, 16)))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="triexp", choices=["triexp", "biexp", "mono", "nnls"])
    ap.add_argument("--jac", default="fd", choices=["fd", "analytic"])
    ap.add_argument("--no-pcov", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the NNLS leg that is reported beside triexp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--voxels", type=int, default=0, help="override voxels per GPU (debug; marks the line invalid)")
    return ap.parse_args()


class CurvefitLeg:
    def __init__(self, workload, device, jac, want_pcov, n_vox_override=0, seed=0):
        import torch
        from pyneapple_amd import api, synth

        self.torch = torch
        self.api = api
        model, n_b, shape = synth.WORKLOADS[workload]
        self.model, self.n_b, self.shape = model, n_b, shape
        self.n_vox = n_vox_override or int(np.prod(shape))
        self.names, self.p0, self.lo, self.hi = synth.shared_arrays(model)
        n = len(self.names)
        self.n = n
        self.b, self.y = synth.make_torch(model, self.n_vox, n_b, device, sigma=0.01, seed=synth.SEED + seed)
        self.opts = api.make_opts(model, n_b, max_nfev=250, ftol=1e-8, jac=jac)
        self.popt = torch.empty((n, self.n_vox), dtype=torch.float64, device=device)
        self.pcov = torch.empty((self.n_vox, n, n), dtype=torch.float64, device=device) if want_pcov else None
        self.status = torch.empty(self.n_vox, dtype=torch.int8, device=device)
        self.nfev = torch.empty(self.n_vox, dtype=torch.int32, device=device)
        self.cost = torch.empty(self.n_vox, dtype=torch.float64, device=device)
        self.device = device
        # algorithmic HBM bytes per voxel: signal in, popt/status/nfev/cost (+pcov) out  
        self.bytes_per_voxel = n_b * 8 + n * 8 + 1 + 4 + 8 + (n * n * 8 if want_pcov else 0)
        self.dtype = "f64"
        self.kernel = f"curvefit_kernel<{model},{jac}>"

    def step(self):
        stream = self.torch.cuda.current_stream().cuda_stream
        self.api.curvefit_device(self.opts, self.n_vox, self.b, self.y, self.p0, self.lo, self.hi, None, self.popt,
                                 self.pcov, self.status, self.nfev, self.cost, self.device.index, stream)

    def check(self):
        ok = (self.status > 0).double().mean().item()
        return {"converged_frac": ok, "mean_nfev": self.nfev.double().mean().item()}

    def cpu_baseline(self, seconds_target=15.0):
        """The oracle (C restatement of SciPy TRF) on the host cores, bounded sample of the same workload."""
        from oracle import pnx_oracle as O
        from pyneapple_amd import synth

        cores = host_cores()
        n = 32768 * cores  # ~10-20 s of CPU work on 16 threads
        b, y, _ = synth.make_numpy(self.model, n, self.n_b, sigma=0.01)
        O.curvefit(self.model, b, y[:256], self.p0, self.lo, self.hi, n_threads=cores)  # warm-up / build
        t = time.perf_counter()
        r = O.curvefit(self.model, b, y, self.p0, self.lo, self.hi, max_nfev=250, ftol=1e-8, jac="fd",
                       n_threads=cores)
        dt = time.perf_counter() - t
        return {"value": n / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
                "sample": f"{n} voxels of the same synthetic {self.model} volume, oracle/pnx_oracle_trf.c, "
                          f"OpenMP over voxels, {dt:.2f} s", "converged_frac": float((r['status'] > 0).mean())}


class NnlsLeg:
    def __init__(self, device, n_vox_override=0, seed=0):
        import torch
        from pyneapple_amd import api, synth

        self.torch = torch
        _, n_b, shape = synth.WORKLOADS["nnls"]
        self.n_b = n_b
        self.n_vox = n_vox_override or int(np.prod(shape))
        cfg = synth.NNLS_CFG
        self.cfg = cfg
        self.bins, self.basis, self.reg = synth.nnls_matrices(n_b, cfg)
        self.plan = api.NnlsPlan(self.basis, self.reg, device.index)
        _, self.y = synth.make_torch("tri_reduced", self.n_vox, n_b, device, sigma=0.01, seed=synth.SEED + seed,
                                     scale=1000.0)
        nb = cfg["n_bins"]
        self.coeff = torch.empty((self.n_vox, nb), dtype=torch.float64, device=device)
        self.rnorm = torch.empty(self.n_vox, dtype=torch.float64, device=device)
        self.status = torch.empty(self.n_vox, dtype=torch.int8, device=device)
        self.iters = torch.empty(self.n_vox, dtype=torch.int32, device=device)
        self.bytes_per_voxel = n_b * 8 + nb * 8 + 8 + 1 + 4
        self.dtype = "f64"
        self.kernel = "nnls_kernel"
        self.model = "nnls"

    def step(self):
        stream = self.torch.cuda.current_stream().cuda_stream
        self.plan.solve_device(self.n_vox, self.y, self.cfg["max_iter"], self.coeff, self.rnorm, self.status,
                               self.iters, stream)

    def check(self):
        return {"converged_frac": (self.status == 1).double().mean().item(),
                "mean_iters": self.iters.double().mean().item()}

    def cpu_baseline(self):
        from oracle import pnx_oracle as O
        from pyneapple_amd import synth

        cores = host_cores()
        n = 1536 * cores  # ~10-15 s of CPU work on 16 threads
        _, y, _ = synth.make_numpy("tri_reduced", n, self.n_b, sigma=0.01, scale=1000.0)
        O.nnls(self.basis, self.reg, y[:cores], self.cfg["max_iter"], n_threads=cores)
        t = time.perf_counter()
        O.nnls(self.basis, self.reg, y, self.cfg["max_iter"], n_threads=cores)
        dt = time.perf_counter() - t
        return {"value": n / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
                "sample": f"{n} voxels, oracle/pnx_oracle_nnls.c (Lawson-Hanson, Householder/Givens), "
                          f"OpenMP over voxels, {dt:.2f} s"}


def timed(leg, steps, warmup, world, dist, torch):
    for _ in range(warmup):
        leg.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        leg.step()
        b.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in ev]  # HIP events on the launch stream
    from pyneapple_amd.sharding import max_over_ranks

    dt = max_over_ranks(dt, dist if world > 1 else None, device="cuda")
    return dt, kernel_ms


def pmc_traffic(kernel_prefix: str):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.sh -> profiles/r01_j_traffic.json;
    FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as is).  bench.py cannot run rocprofv3 around itself,
    so the figure is the one measured on this kernel and this full-size workload when the profile was taken."""
    try:
        with open(os.path.join(HERE, "profiles", "r01_j_traffic.json")) as f:
            t = json.load(f)
        for name, v in t.items():
            if kernel_prefix in name:
                return v["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = fp64 matrix peak (MI355X_MICROARCH.md: half the 157.3 TF fp32 rate)


def pmc_flops(kernel_prefix: str):
    """fp64 flop per voxel ISSUED by the kernel (64 lanes per fp64 VALU instruction, FMA = 2), from the committed
    PMC pass (profiles/r01_j_pmc_*.txt -> profiles/r01_j_flops.json)."""
    try:
        with open(os.path.join(HERE, "profiles", "r01_j_flops.json")) as f:
            t = json.load(f)
        for name, v in t.items():
            if kernel_prefix in name:
                return v
    except Exception:
        pass
    return None


def mfma_roofline(device, torch, reps=20):
    """The NNLS Gram step (aty = y . basis on v_mfma_f64_16x16x4) alone, 2^20 voxels x 32 b-values x 250 bins:
    algorithmic flops 2 * n_vox * n_b * n_bins per launch against the fp64 matrix peak."""
    from pyneapple_amd import api, synth

    n_vox, n_b = 1 << 20, 32
    _, basis, reg = synth.nnls_matrices(n_b)
    plan = api.NnlsPlan(basis, reg, device.index)
    _, y = synth.make_torch("tri_reduced", n_vox, n_b, device, sigma=0.01, scale=1000.0)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.aty_device(n_vox, y, None, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.aty_device(n_vox, y, None, stream)
    e1.record()
    torch.cuda.synchronize()
    plan.close()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * n_vox * n_b * basis.shape[1]
    ach = flops / (ms * 1e-3) / 1e12
    out_bytes = n_vox * 256 * 8 + n_vox * n_b * 8
    return {"bound": "mfma", "kernel": "nnls_aty_mfma_kernel (v_mfma_f64_16x16x4_f64)", "achieved": ach,
            "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": pmc_traffic("nnls_aty_mfma_kernel"),
            "algorithmic_flop_per_launch": flops, "kernel_ms_avg": ms, "dtype": "f64",
            "hbm_GBps": out_bytes / (ms * 1e-3) / 1e9,
            "note": "fp64 because an fp32 Gram product breaks NNLS parity (DESIGN.md 4.3); the (n_vox,256) fp64 product is "
                    "materialised, so the step is co-bound by its HBM write"}


def nnls_traffic(n_vox):
    """PMC HBM bytes of one NNLS step: the per-launch figure (one 2^20-voxel chunk) times the chunks per step."""
    t = pmc_traffic("nnls_kernel")
    return None if t is None else t * ((n_vox + (1 << 20) - 1) >> 20)


def sweep_roofline(device, torch, n_vox_override=0, reps=20):
    """The LM residual/Jacobian/normal-equation sweep as a standalone HBM-streaming kernel (pnx_sweep_f32),
    triexp on the C3 volume: 232 algorithmic bytes per voxel-sweep against the HBM roofline."""
    from pyneapple_amd import api, synth

    model, n_b, shape = synth.WORKLOADS["triexp"]
    n_vox = n_vox_override or int(np.prod(shape))
    b, y64 = synth.make_torch(model, n_vox, n_b, device, sigma=0.01)
    y = y64.float()
    del y64
    names, p0, _, _ = synth.shared_arrays(model)
    n = len(names)
    ntri = n * (n + 1) // 2
    params = torch.tensor(p0, dtype=torch.float32, device=device)[:, None].repeat(1, n_vox).contiguous()
    params *= 1.0 + 0.05 * torch.rand_like(params)
    cost = torch.empty(n_vox, dtype=torch.float32, device=device)
    g = torch.empty((n, n_vox), dtype=torch.float32, device=device)
    h = torch.empty((ntri, n_vox), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        api.sweep_device(model, n_vox, b, y, params, cost, g, h, device.index, stream)
    torch.cuda.synchronize()
    # one HIP-event pair per launch (on the launch stream): the average launch duration, free of host-side gaps
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in evs:
        e0.record()
        api.sweep_device(model, n_vox, b, y, params, cost, g, h, device.index, stream)
        e1.record()
    torch.cuda.synchronize()
    ms_pair = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
    # the same launches back to back under one event pair: the figure rocprofv3's per-kernel average agrees with
    # (an event pair per launch adds ~25 us of marker serialisation to a 0.24 ms kernel)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for _ in range(reps):
        api.sweep_device(model, n_vox, b, y, params, cost, g, h, device.index, stream)
    eb.record()
    torch.cuda.synchronize()
    ms = ea.elapsed_time(eb) / reps
    bytes_per = (n_b + n + ntri + n + 1) * 4
    ach = bytes_per * n_vox / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "sweep_kernel<tri_reduced,f32>", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": None if n_vox_override else pmc_traffic("sweep_"),
            "algorithmic_bytes_per_launch": bytes_per * n_vox, "algorithmic_bytes_per_voxel": bytes_per,
            "kernel_ms_avg": ms, "per_launch_event_pair_ms_avg": ms_pair, "voxel_sweeps_per_s": n_vox / (ms * 1e-3), "dtype": "f32"}


def main():
    args = parse()
    import torch

    from pyneapple_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # RCCL; only used for the barrier and the max-reduce
    _lib.load()

    want_pcov = not args.no_pcov
    if args.workload == "nnls":
        leg = NnlsLeg(device, args.voxels, seed=rank)
    else:
        leg = CurvefitLeg(args.workload, device, args.jac, want_pcov, args.voxels, seed=rank)
    dt, kernel_ms = timed(leg, args.steps, args.warmup, world, dist, torch)
    total_vox = leg.n_vox * world * args.steps
    value = total_vox / dt
    k_avg = float(np.mean(kernel_ms)) * 1e-3
    achieved = leg.bytes_per_voxel * leg.n_vox / k_avg / 1e9
    out = {
        "metric": METRIC, "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": leg.dtype, "data": "synthetic",
        "config": {"workload": {"triexp": "triexp bounded LM (SciPy-TRF parity), 256x256x64x32, fp64, FD Jacobian" if args.jac == "fd" else "triexp bounded LM, 256x256x64x32, fp64, analytic Jacobian",
                                "biexp": "biexp bounded LM, 128x128x32x24, fp64",
                                "mono": "monoexp curvefit, 32x32x1x16, fp64",
                                "nnls": "NNLS reg_order=2 mu=0.02 n_bins=250, 256x256x64x32, fp64"}[args.workload],
                   "voxels_per_gpu": leg.n_vox, "jac": args.jac if args.workload != "nnls" else None,
                   "pcov": want_pcov if args.workload != "nnls" else None, "parallelism": f"voxel-shard x{world}",
                   "full_size": not args.voxels},
        "roofline": {"bound": "hbm", "kernel": leg.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic("curvefit_kernel<4, 5, true, false, false>") if (args.workload == "triexp" and args.jac == "fd" and not args.voxels) else None,
                     "algorithmic_bytes_per_launch": leg.bytes_per_voxel * leg.n_vox,
                     "algorithmic_bytes_per_voxel": leg.bytes_per_voxel, "kernel_ms_avg": k_avg * 1e3,
                     "note": "whole-fit kernel is fp64-VALU/transcendental bound, not HBM bound (DESIGN.md section 4)"},
        "check": leg.check(),
    }
    fl = pmc_flops("curvefit_kernel<4, 5, true, false, false>") if (args.workload == "triexp" and args.jac == "fd") else None
    if fl:  # the bound that actually applies: fp64 VALU issue (flop count from the committed PMC pass, time live)
        tf = fl["fp64_flop_per_voxel_issued"] * leg.n_vox / k_avg / 1e12
        out["roofline"]["valu_f64"] = {"achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                                       "fp64_flop_per_voxel_issued": fl["fp64_flop_per_voxel_issued"],
                                       "lane_utilisation": fl["lane_utilisation"], "source": "profiles/r01_j_flops.json"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = leg.cpu_baseline()
    if args.workload == "triexp" and not args.no_secondary:
        del leg
        torch.cuda.empty_cache()
        leg2 = NnlsLeg(device, args.voxels, seed=rank)
        dt2, k2 = timed(leg2, max(1, min(args.steps, 2)), 1 if args.warmup else 0, world, dist, torch)
        steps2 = max(1, min(args.steps, 2))
        k2avg = float(np.mean(k2)) * 1e-3
        ach2 = leg2.bytes_per_voxel * leg2.n_vox / k2avg / 1e9
        sec = {"workload": "NNLS reg_order=2 mu=0.02 n_bins=250, 256x256x64x32, fp64", "value": leg2.n_vox * world * steps2 / dt2,
               "unit": "voxels/s", "steps": steps2, "ms_per_step": dt2 / steps2 * 1e3, "check": leg2.check(),
               "roofline": {"bound": "hbm", "kernel": "nnls_kernel", "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach2 / HBM_PEAK_GBS, "traffic": nnls_traffic(leg2.n_vox) if not args.voxels else None,
                            "algorithmic_bytes_per_voxel": leg2.bytes_per_voxel, "kernel_ms_avg": k2avg * 1e3,
                            "note": "active-set loop is VALU-issue bound: 4 waves per SIMD keep the issue slots 74 % busy, 24 % of the instructions are fp64 arithmetic (DESIGN.md 4.3)"}}
        fl2 = pmc_flops("nnls_kernel")
        if fl2:  # the active-set kernel is VALU-issue bound (74 % of the SIMD issue slots, a quarter of them fp64)
            tf2 = fl2["fp64_flop_per_voxel_issued"] * leg2.n_vox / k2avg / 1e12
            sec["roofline"]["valu_f64"] = {"achieved": tf2, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf2 / FP64_PEAK_TFLOPS,
                                           "fp64_flop_per_voxel_issued": fl2["fp64_flop_per_voxel_issued"],
                                           "fp64_share_of_valu_instructions": fl2["fp64_share_of_valu_instructions"],
                                           "lane_utilisation": fl2["lane_utilisation"], "source": "profiles/r01_j_flops.json"}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            sec["cpu_baseline"] = leg2.cpu_baseline()
        out["secondary"] = sec
    if args.workload == "triexp" and not args.no_secondary:
        out["roofline_sweep"] = sweep_roofline(device, torch, args.voxels)
        out["roofline_mfma"] = mfma_roofline(device, torch)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
