#!/usr/bin/env python3
"""bench.py -- voxels/s of the per-voxel fitting hot path on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload triexp|biexp|mono|nnls] [--no-secondary]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one batched fit through the C ABI, device-pointer mode) over ONE seed-fixed
synthetic volume that is already resident in HBM.  Default workload: BASELINE.json configs[2] (triexp bounded TRF,
256x256x64 voxels x 32 b-values), computed in fp64 with SciPy's 2-point finite difference Jacobian and with the
covariance output, i.e. exactly what the reference's pixelwise fitter asks of its solver.

Multi-GPU (north star: "voxels shard embarrassingly across the GPUs of one node"): the ONE volume is split into
contiguous row ranges, rank r fits sharding.shard_range(n_vox, r, N) -- strong scaling, no data-path collective;
RCCL only carries the barrier and the MAX-reduce of the timed region.  value = voxels of the whole volume x steps /
(max over ranks of the timed region).  `--gpus N` without a torchrun environment starts the N ranks itself: fresh
child processes, spawned before anything in this process touches the GPU.  Rank 0 prints ONE JSON line.

`value` is the device-resident rate (inputs in HBM when the timed region starts, as the bench contract asks);
`host_mode` (every rank on its shard, max over ranks) is the PCIe-inclusive rate of the same work through PNX_MEM_HOST --
numpy arrays in, numpy arrays out, what the reference's fitter hands its solver (fitters/pixelwise.py:91-96).
`noise_sweep` (N = 1): both fits (the whole volume / its first 2^20 voxels) at 0 / 1 / 5 % noise -- the rates depend on the data (the
benchmark's volume carries 1 %): the curve fit slows down with the noise (longer trust-region walks, lanes out of step), the
NNLS speeds up (smaller supports).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = fp64 matrix peak (half the 157.3 TF fp32 rate of MI355X_MICROARCH.md)
METRIC = "voxels/sec (pixelwise triexp LM & 250-bin NNLS) at 1/2/4/8 MI355X"
CURVEFIT_C3_KERNEL = "curvefit_kernel<4, 5, true, false, false, false>"  # rocprofv3's name of the C3 instantiation
NNLS_KERNEL = "nnls_blk_kernel"  # the C4 plan (banded regulariser, 32 measurements): pnx_nnls_blk.hip


def host_cores() -> int:
    """Threads for the CPU baseline: the box's CPU share (16 per GPU on the pool), not the host's 256 cores."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return int(os.environ.get("PNX_CPU_THREADS", min(n, 16)))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="triexp", choices=["triexp", "biexp", "mono", "nnls"])
    ap.add_argument("--jac", default="fd", choices=["fd", "analytic"])
    ap.add_argument("--no-pcov", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the NNLS leg that is reported beside triexp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-mode", action="store_true", help="skip the PCIe-inclusive host-pointer legs")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-passes-in-flight leg of the curve fit")
    ap.add_argument("--no-noise-sweep", action="store_true", help="skip the 0 / 1 / 5 % noise sweep of both fits (N = 1, 2^20 voxels each)")
    ap.add_argument("--voxels", type=int, default=0, help="override the voxels of the volume (debug; marks the line invalid)")
    ap.add_argument("--fail-rank", type=int, default=-1, help="(launcher test) this rank exits with code 3 at start-up")
    ap.add_argument("--launch-check", action="store_true",
                    help="every rank prints its rank / world / shard as JSON and exits without touching the GPU")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` outside torchrun starts N fresh ranks before this process touches the GPU
def launch(args, argv) -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile

    procs, outs_f = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # a rank's stdout goes to a temporary file, not a pipe: nothing drains a pipe while the ranks are polled, and a rank
        # that writes more than the pipe buffer holds would block in write() for ever
        f = tempfile.TemporaryFile() if (r == 0 or args.launch_check) else None
        outs_f.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=f if f is not None else subprocess.DEVNULL))
    # poll every rank: the first one that exits non-zero ends the run (a rank that dies at start-up would otherwise leave
    # the others waiting in init_process_group / the barrier until the backend's own timeout)
    rc = 0
    alive = set(range(len(procs)))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write(f"bench.py: rank {r} exited with code {code}; stopping the other ranks\n")
                for q in alive:
                    procs[q].terminate()
        if alive:
            time.sleep(0.05)
            if rc != 0:
                deadline = time.time() + 10.0
                while alive and time.time() < deadline:
                    alive = {q for q in alive if procs[q].poll() is None}
                    time.sleep(0.05)
                for q in alive:
                    procs[q].kill()
                alive = set()
    for p in procs:
        p.wait()
    outs = []
    for f in outs_f:
        if f is None:
            outs.append("")
        else:
            f.seek(0)
            outs.append(f.read().decode())
            f.close()
    sys.stdout.write("".join(outs) if args.launch_check else outs[0])
    sys.stdout.flush()
    return rc


def _cpulist(text):
    from pyneapple_amd.sharding import cpulist

    return cpulist(text)


def _fmt_cpus(cpus):
    from pyneapple_amd.sharding import fmt_cpus

    return fmt_cpus(cpus)


def gpu_numa_cpus(index):
    """CPUs local to the `index`-th GPU, read from sysfs only (pyneapple_amd/sharding.py): nothing here touches the GPU, so it
    can run before the affinity is set and before HIP starts its helper threads."""
    from pyneapple_amd.sharding import gpu_numa_cpus as g

    return g(index)


def pin_rank_to_gpu_numa(local_rank, local_world, share_gpu=False):
    """Restrict this rank (and every thread it starts later: the HIP runtime's, the library's IN / OUT / page-touch helpers) to
    CPUs on the NUMA node of its GPU, in-process and before anything touches the GPU -- never through numactl / taskset
    wrappers, which a profiler's preloaded library turns into an exec after GPU initialisation.  Ranks whose GPUs share a
    node split that node's CPUs between them.  Returns what was done, for the bench line; never raises."""
    try:
        have = os.sched_getaffinity(0)
    except AttributeError:
        return {"cpus": None, "source": "no sched_getaffinity"}
    try:
        if os.environ.get("PNX_BENCH_NO_PIN") == "1":
            return {"cpus": _fmt_cpus(have), "source": "PNX_BENCH_NO_PIN"}
        dev = 0 if share_gpu else local_rank
        node, local = gpu_numa_cpus(dev)
        source = "gpu numa node (sysfs)"
        if not local or not (local & have):
            node, local, source = None, set(have), "affinity mask split by rank (no NUMA information)"
        mine = sorted(local & have)
        # ranks that land on the same CPU set take disjoint slices of it
        peers = []
        for r in range(local_world):
            n2, l2 = (node, local) if (share_gpu or source.startswith("affinity")) else gpu_numa_cpus(r)
            if l2 and sorted(l2 & have) == mine:
                peers.append(r)
        if local_rank in peers and len(peers) > 1 and len(mine) >= len(peers):
            k = peers.index(local_rank)
            per = len(mine) // len(peers)
            mine = mine[k * per:(k + 1) * per] if k < len(peers) - 1 else mine[k * per:]
        if mine:
            os.sched_setaffinity(0, mine)
        return {"cpus": _fmt_cpus(mine), "n_cpus": len(mine), "numa_node": node, "source": source}
    except Exception as e:  # a bench must not fail over an affinity nicety
        return {"cpus": None, "source": f"unpinned ({type(e).__name__})"}


def volume_rows(workload, args, rank, world):
    """(n_vox of the whole volume, [start, stop) of this rank): contiguous row ranges of ONE volume."""
    from pyneapple_amd import synth
    from pyneapple_amd.sharding import shard_range

    n_vox = args.voxels or int(np.prod(synth.WORKLOADS[workload][2]))
    return n_vox, shard_range(n_vox, rank, world)


class CurvefitLeg:
    def __init__(self, workload, device, jac, want_pcov, n_vox_total, rows, sigma=0.01):
        import torch
        from pyneapple_amd import api, synth

        self.torch = torch
        self.api = api
        model, n_b, shape = synth.WORKLOADS[workload]
        self.model, self.n_b, self.shape = model, n_b, shape
        self.n_vox_total, self.rows = n_vox_total, rows
        self.n_vox = rows[1] - rows[0]
        self.names, self.p0, self.lo, self.hi = synth.shared_arrays(model)
        n = len(self.names)
        self.n = n
        self.jac, self.want_pcov = jac, want_pcov
        self.b, self.y = synth.make_torch_rows(model, rows[0], rows[1], n_b, device, sigma=sigma)
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

    def pipelined(self, steps, world, dist):
        """The same K passes with TWO in flight: launches alternate between two streams and two sets of result buffers, as a
        caller with a queue of independent batches would issue them.  A single pass ends in a straggler tail (a handful of
        voxels that need 150-250 evaluations at ~36 us each, started wherever they sit in the volume: profiles/curvefit_scale.py);
        the next launch fills the SIMDs that tail leaves idle.  Same barriers / max over ranks as the primary timing."""
        torch = self.torch
        dev = self.device
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        n = self.n
        second = (torch.empty_like(self.popt), torch.empty_like(self.pcov) if self.pcov is not None else None,
                  torch.empty_like(self.status), torch.empty_like(self.nfev), torch.empty_like(self.cost))
        outs = [(self.popt, self.pcov, self.status, self.nfev, self.cost), second]

        def launch(k):
            po, pc, st, nf, co = outs[k % 2]
            self.api.curvefit_device(self.opts, self.n_vox, self.b, self.y, self.p0, self.lo, self.hi, None, po, pc, st, nf, co,
                                     dev.index, streams[k % 2].cuda_stream)

        main = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(main)
        for k in range(2):
            launch(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            launch(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        from pyneapple_amd.sharding import max_over_ranks

        dt = max_over_ranks(dt, dist if world > 1 else None, device="cuda" if dist is None or dist.get_backend() == "nccl" else "cpu")
        same = bool((second[0] == self.popt).all().item()) if steps >= 2 else True
        del second, outs
        return {"workload": "the same K passes, two in flight on alternating streams (independent batches in a queue)",
                "value": self.n_vox_total * steps / dt, "unit": "voxels/s", "steps": steps, "steps_in_flight": 2,
                "ms_per_step": dt / steps * 1e3, "results_identical_across_buffers": same}

    def host_mode(self, reps=None):
        """The same fit through PNX_MEM_HOST: numpy signal in, numpy popt / pcov / status / nfev / cost out (PCIe inclusive).
        Three timed calls for the big volumes, six for a call of a few ms; the figure is the median and every call is in
        ms_reps (one call in a few is slow on the box right after another leg's buffers were released: 59 instead of 39 ms
        for C3, 12 instead of 4.5 ms for C2)."""
        if reps is None:
            reps = 3 if self.n_vox >= (1 << 21) else 6
        y = self.y.cpu().numpy()
        kw = dict(max_nfev=250, ftol=1e-8, jac=self.jac, want_pcov=self.want_pcov, device=self.device.index)
        r = self.api.curvefit(self.model, self.b, y, self.p0, self.lo, self.hi, **kw)  # warm-up (slots, first-touch)
        ts = []
        for _ in range(reps):
            del r  # the previous result is released before the clock starts (unmapping 1 GB costs tens of ms)
            t = time.perf_counter()
            r = self.api.curvefit(self.model, self.b, y, self.p0, self.lo, self.hi, **kw)
            ts.append(time.perf_counter() - t)
        dt = float(np.median(ts))
        d2h = sum(a.nbytes for a in r.values() if a is not None)
        same = bool((self.torch.from_numpy(r["popt"]).to(self.popt.device) == self.popt).all().item())
        return {"workload": "same volume, host (numpy) arrays in and out through PNX_MEM_HOST", "value": self.n_vox / dt,
                "unit": "voxels/s", "ms_per_step": dt * 1e3, "ms_reps": [t * 1e3 for t in ts], "h2d_bytes": int(y.nbytes), "d2h_bytes": int(d2h),
                "pcie_GBps": (y.nbytes + d2h) / dt / 1e9, "steps": reps, "equals_device_resident_result": same}

    def host_mode_f32(self, reps=3):
        """BASELINE's "fp32" configuration as a Pyneapple user runs it: float32 signal in, float32 popt / pcov / cost out through
        pnx_curvefit_batch_f32 (fp64 arithmetic on the widened values; half of the PCIe bytes of host_mode)."""
        y = self.y.cpu().numpy().astype(np.float32)
        kw = dict(max_nfev=250, ftol=1e-8, jac=self.jac, want_pcov=self.want_pcov, device=self.device.index)
        r = self.api.curvefit(self.model, self.b, y, self.p0, self.lo, self.hi, **kw)
        ts = []
        for _ in range(reps):
            del r
            t = time.perf_counter()
            r = self.api.curvefit(self.model, self.b, y, self.p0, self.lo, self.hi, **kw)
            ts.append(time.perf_counter() - t)
        dt = float(np.median(ts))
        d2h = sum(a.nbytes for a in r.values() if a is not None)
        return {"workload": "same volume as float32 numpy arrays in and out (pnx_curvefit_batch_f32, PNX_MEM_HOST)",
                "value": self.n_vox / dt, "unit": "voxels/s", "ms_per_step": dt * 1e3, "h2d_bytes": int(y.nbytes),
                "d2h_bytes": int(d2h), "pcie_GBps": (y.nbytes + d2h) / dt / 1e9, "steps": reps, "ms_reps": [t * 1e3 for t in ts],
                "result_dtype": str(r["popt"].dtype), "converged_frac": float((r["status"] > 0).mean())}

    def cpu_baseline(self):
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
                "sample": f"{n} voxels of the same synthetic {self.model} distribution, oracle/pnx_oracle_trf.c, "
                          f"OpenMP over voxels, {dt:.2f} s", "converged_frac": float((r['status'] > 0).mean())}


class NnlsLeg:
    def __init__(self, device, n_vox_total, rows, sigma=0.01):
        import torch
        from pyneapple_amd import api, synth

        self.torch = torch
        self.api = api
        _, n_b, shape = synth.WORKLOADS["nnls"]
        self.n_b = n_b
        self.n_vox_total, self.rows = n_vox_total, rows
        self.n_vox = rows[1] - rows[0]
        cfg = synth.NNLS_CFG
        self.cfg = cfg
        self.bins, self.basis, self.reg = synth.nnls_matrices(n_b, cfg)
        self.plan = api.NnlsPlan(self.basis, self.reg, device.index)
        _, self.y = synth.make_torch_rows("tri_reduced", rows[0], rows[1], n_b, device, sigma=sigma, scale=1000.0)
        nb = cfg["n_bins"]
        self.coeff = torch.empty((self.n_vox, nb), dtype=torch.float64, device=device)
        self.rnorm = torch.empty(self.n_vox, dtype=torch.float64, device=device)
        self.status = torch.empty(self.n_vox, dtype=torch.int8, device=device)
        self.iters = torch.empty(self.n_vox, dtype=torch.int32, device=device)
        self.bytes_per_voxel = n_b * 8 + nb * 8 + 8 + 1 + 4
        self.dtype = "f64"
        self.kernel = NNLS_KERNEL
        self.model = "nnls"
        self.device = device

    def step(self):
        stream = self.torch.cuda.current_stream().cuda_stream
        self.plan.solve_device(self.n_vox, self.y, self.cfg["max_iter"], self.coeff, self.rnorm, self.status,
                               self.iters, stream)

    def check(self):
        return {"converged_frac": (self.status == 1).double().mean().item(),
                "mean_iters": self.iters.double().mean().item()}

    def host_mode(self, reps=3):
        """PNX_MEM_HOST solve: numpy signal in, numpy (n_vox, 250) float64 spectra out.  The whole volume when the host
        has the memory for its 8.4 GB result, else its first 2^20 voxels.  Median of three calls per variant (one call in a
        few runs 10 % slow on the box: 730 instead of 654 ms; every call is in ms_reps)."""
        try:
            import psutil
            avail = psutil.virtual_memory().available
        except Exception:
            avail = 0
        n = self.n_vox if avail > 6 * self.n_vox * self.cfg["n_bins"] * 8 else min(self.n_vox, 1 << 20)
        y = self.y[:n].cpu().numpy()
        r = self.plan.solve(y[: min(n, 1 << 16)], self.cfg["max_iter"])  # warm-up (slots, staging threads)
        ts = []
        for _ in range(reps):
            del r
            t = time.perf_counter()
            r = self.plan.solve(y, self.cfg["max_iter"])
            ts.append(time.perf_counter() - t)
        dt = float(np.median(ts))
        d2h = sum(a.nbytes for a in r.values())
        same = bool((self.torch.from_numpy(r["coefficients"][:65536]).to(self.coeff.device) == self.coeff[:65536]).all().item())
        out = {"workload": f"{n} voxels of the same volume, host (numpy) arrays in and out through PNX_MEM_HOST",
               "value": n / dt, "unit": "voxels/s", "ms_per_step": dt * 1e3, "h2d_bytes": int(y.nbytes),
               "d2h_bytes": int(d2h), "pcie_GBps": (y.nbytes + d2h) / dt / 1e9, "steps": reps, "ms_reps": [t * 1e3 for t in ts],
               "equals_device_resident_result": same}
        del r
        # the same fit with the spectrum post-processing on the device (pnx_nnls_solve_peaks_f64: find_spectrum_peaks +
        # apply_cutoffs of utility/spectrum.py per voxel): a peak table comes back instead of 2 KB per voxel
        cuts = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
        kw = dict(max_iter=self.cfg["max_iter"], height=0.1, regularized=True, max_peaks=8, cutoffs=cuts)
        rp = self.plan.solve_peaks(y[: min(n, 1 << 16)], self.bins, **kw)
        tp = []
        for _ in range(reps):
            del rp
            t = time.perf_counter()
            rp = self.plan.solve_peaks(y, self.bins, **kw)
            tp.append(time.perf_counter() - t)
        dtp = float(np.median(tp))
        out["with_peak_tables_instead_of_spectra"] = {
            "value": n / dtp, "unit": "voxels/s", "ms_per_step": dtp * 1e3, "ms_reps": [t * 1e3 for t in tp],
            "d2h_bytes": int(sum(a.nbytes for a in rp.values() if a is not None)),
            "mean_peaks_per_voxel": float(rp["n_peaks"].mean())}
        del rp
        # float32 storage (pnx_nnls_solve_f32): the signal goes in and the spectra come back as float32, half the PCIe bytes
        y32 = y.astype(np.float32)
        r = self.plan.solve(y32[: min(n, 1 << 16)], self.cfg["max_iter"])
        t32 = []
        for _ in range(reps):
            del r
            t = time.perf_counter()
            r = self.plan.solve(y32, self.cfg["max_iter"])
            t32.append(time.perf_counter() - t)
        dt32 = float(np.median(t32))
        out["f32"] = {"workload": "the same voxels as float32 arrays in and out (pnx_nnls_solve_f32)", "value": n / dt32,
                      "unit": "voxels/s", "ms_per_step": dt32 * 1e3, "ms_reps": [t * 1e3 for t in t32], "h2d_bytes": int(y32.nbytes),
                      "d2h_bytes": int(sum(a.nbytes for a in r.values())), "result_dtype": str(r["coefficients"].dtype)}
        return out

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


def _gather_floats(vals, dist, device):
    """[per-rank list of floats] on every rank."""
    import torch

    t = torch.tensor(vals, dtype=torch.float64, device=device)
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    return [[float(x) for x in b.cpu()] for b in bufs]


def all_ranks_timed(call, verify, reps, n_total, world, dist, rdev, sync=lambda: None):
    """`reps` barrier-aligned repetitions of `call()` on every rank: value = n_total / median over reps of (max over ranks of the
    call time).  `verify(result)` -> (equals the device-resident result, bytes moved).  A failure on one rank (no host memory
    for its result, a HIP error) must not leave the others waiting at a barrier and take the resident figures of the line with
    it: every rank goes through the same collectives whatever happens to its calls, and the line reports the failure.
    (tests/test_multirank_gloo.py runs this with world_size 2 on the CPU, including a rank that fails.)"""
    err = ""
    r = None
    try:
        r = call()  # warm-up: staging slab / ring slots, first-touch of the helper threads
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    locals_ = []
    for _ in range(reps):
        r = None  # the previous result is released before the clock starts
        sync()
        dist.barrier()
        t = time.perf_counter()
        if not err:
            try:
                r = call()
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
        locals_.append(time.perf_counter() - t)
        dist.barrier()
    same, nbytes = False, 0
    if not err and r is not None:
        same, nbytes = verify(r)
    per_rank = _gather_floats(locals_ + [float(same), float(nbytes), 0.0 if err else 1.0], dist, rdev)
    if any(pr[reps + 2] == 0.0 for pr in per_rank):
        bad = [i for i, pr in enumerate(per_rank) if pr[reps + 2] == 0.0]
        return {"value": None, "unit": "voxels/s", "n_gpus": world, "error": f"host-array call failed on rank(s) {bad}" + (f"; this rank: {err}" if err else "")}
    call_s = [max(pr[i] for pr in per_rank) for i in range(reps)]  # a rep ends when its slowest rank is home
    dt = float(np.median(call_s))
    return {"value": n_total / dt, "unit": "voxels/s", "ms_per_step": dt * 1e3, "ms_reps": [c * 1e3 for c in call_s], "steps": reps,
            "n_gpus": world, "per_rank_ms": [[x * 1e3 for x in pr[:reps]] for pr in per_rank],
            "per_rank_pcie_GBps": [pr[reps + 1] / float(np.median(pr[:reps])) / 1e9 for pr in per_rank],
            "pcie_GBps": sum(pr[reps + 1] for pr in per_rank) / dt / 1e9,
            "equals_device_resident_result": all(pr[reps] == 1.0 for pr in per_rank)}


def host_mode_ranks(leg, dist, world, torch, n_total, reps=3):
    """SURVEY 8(d)'s metric at N ranks: every rank hands ITS shard to the C ABI as host (numpy) arrays at the same moment
    -- barrier, call, barrier -- so the ranks' uploads and downloads meet on the host's memory system as they would in a
    node-wide fit.  value = voxels of the whole volume / median over reps of (max over ranks of the call time); per-rank
    PCIe-inclusive GB/s and CPU sets are reported beside it."""
    rdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    y = leg.y.cpu().numpy()
    if isinstance(leg, NnlsLeg):
        call = lambda: leg.plan.solve(y, leg.cfg["max_iter"])
        ref, key = leg.coeff, "coefficients"
    else:
        kw = dict(max_nfev=250, ftol=1e-8, jac=leg.jac, want_pcov=leg.want_pcov, device=leg.device.index)
        call = lambda: leg.api.curvefit(leg.model, leg.b, y, leg.p0, leg.lo, leg.hi, **kw)
        ref, key = leg.popt, "popt"

    def verify(r):
        nbytes = int(y.nbytes + sum(a.nbytes for a in r.values() if a is not None))
        m = min(65536, ref.shape[-1] if key == "popt" else ref.shape[0])
        got = torch.from_numpy(np.ascontiguousarray(r[key][..., :m] if key == "popt" else r[key][:m])).to(ref.device)
        return bool((got == (ref[..., :m] if key == "popt" else ref[:m])).all().item()), nbytes

    return all_ranks_timed(call, verify, reps, n_total, world, dist, rdev, sync=torch.cuda.synchronize)


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

    dt = max_over_ranks(dt, dist if world > 1 else None, device="cuda" if dist is None or dist.get_backend() == "nccl" else "cpu")
    return dt, kernel_ms


# ---------------------------------------------------------------------------------------------------------------
# PMC counters: bench.py cannot run rocprofv3 around itself, so HBM traffic and the instruction mix are replayed from
# the committed passes (profiles/pmc_*.sh -> profiles/rNN_x_{traffic,flops}.json) -- but only from a profile stamped
# with the source id of the kernel sources this run was built from (pyneapple_amd/_build.py:source_id); a kernel edit
# without a new profile reports null, never stale counters.
def _profile(kind: str, group: str):
    import glob

    from pyneapple_amd import _build

    want = _build.source_id(group)
    for f in sorted(glob.glob(os.path.join(HERE, "profiles", f"r*_{kind}.json")), reverse=True):
        try:
            with open(f) as fh:
                t = json.load(fh)
        except Exception:
            continue
        if t.get("_source_ids", {}).get(group) == want:
            return t, os.path.relpath(f, HERE), want
    return None, None, want


def pmc_traffic(kernel_prefix: str, group: str):
    """HBM bytes per launch (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, separate passes)."""
    t, src, _ = _profile("traffic", group)
    if t:
        for name, v in t.items():
            if kernel_prefix in name:
                return v["hbm_bytes_per_launch"]
    return None


def pmc_flops(kernel_prefix: str, group: str):
    """fp64 flop per voxel ISSUED by the kernel (64 lanes per fp64 VALU instruction, FMA = 2)."""
    t, src, sid = _profile("flops", group)
    if t:
        for name, v in t.items():
            if kernel_prefix in name:
                return dict(v, source=src, source_id=sid)
    return None


# Launches of the two sub-millisecond roofline kernels before their timed regions: a burst of 20 launches from an idle GPU
# (4 ms) runs 10-15 % slower per launch than the sustained rate (groups of 20: 0.181-0.198 ms, groups of 500: 0.171 ms for the
# sweep, profiles/sweep_sustain.py) -- the clocks are still ramping.  30 warm-up launches, then >= 100 timed ones.
WARM_LAUNCHES = 30


def mfma_roofline(device, torch, reps=100):
    """The NNLS Gram step (aty = y . basis on v_mfma_f64_16x16x4) alone, 2^20 voxels x 32 b-values x 250 bins:
    algorithmic flops 2 * n_vox * n_b * n_bins per launch against the fp64 matrix peak."""
    from pyneapple_amd import api, synth

    n_vox, n_b = 1 << 20, 32
    _, basis, reg = synth.nnls_matrices(n_b)
    plan = api.NnlsPlan(basis, reg, device.index)
    _, y = synth.make_torch("tri_reduced", n_vox, n_b, device, sigma=0.01, scale=1000.0)
    aty = torch.empty((n_vox, 256), dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(WARM_LAUNCHES):
        plan.aty_device(n_vox, y, aty, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.aty_device(n_vox, y, aty, stream)
    e1.record()
    torch.cuda.synchronize()
    plan.close()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * n_vox * n_b * basis.shape[1]
    ach = flops / (ms * 1e-3) / 1e12
    out_bytes = n_vox * 256 * 8 + n_vox * n_b * 8
    return {"bound": "mfma", "kernel": "nnls_aty_mfma_kernel (v_mfma_f64_16x16x4_f64)", "achieved": ach,
            "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
            "traffic": pmc_traffic("nnls_aty_mfma_kernel", "nnls"),
            "algorithmic_flop_per_launch": flops, "kernel_ms_avg": ms, "dtype": "f64",
            "hbm_GBps": out_bytes / (ms * 1e-3) / 1e9,
            "note": "fp64 because an fp32 Gram product breaks NNLS parity (DESIGN.md 4.3); the (n_vox,256) fp64 product is "
                    "materialised, so the step is co-bound by its HBM write"}


def sweep_roofline(device, torch, n_vox_override=0, reps=300):
    """The LM residual/Jacobian/normal-equation sweep as a standalone HBM-streaming kernel (pnx_sweep_f32),
    triexp on the C3 volume: 232 algorithmic bytes per voxel-sweep against the HBM roofline."""
    from pyneapple_amd import api, synth

    model, n_b, shape = synth.WORKLOADS["triexp"]
    n_vox = n_vox_override or int(np.prod(shape))
    b, y = synth.make_torch_rows(model, 0, n_vox, n_b, device, sigma=0.01, dtype=torch.float32)
    names, p0, _, _ = synth.shared_arrays(model)
    n = len(names)
    ntri = n * (n + 1) // 2
    params = torch.tensor(p0, dtype=torch.float32, device=device)[:, None].repeat(1, n_vox).contiguous()
    params *= 1.0 + 0.05 * torch.rand_like(params)
    cost = torch.empty(n_vox, dtype=torch.float32, device=device)
    g = torch.empty((n, n_vox), dtype=torch.float32, device=device)
    h = torch.empty((ntri, n_vox), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(WARM_LAUNCHES):
        api.sweep_device(model, n_vox, b, y, params, cost, g, h, device.index, stream)
    torch.cuda.synchronize()
    # one HIP-event pair per launch (on the launch stream): the average launch duration, free of host-side gaps
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(reps, 50))]
    for e0, e1 in evs:
        e0.record()
        api.sweep_device(model, n_vox, b, y, params, cost, g, h, device.index, stream)
        e1.record()
    torch.cuda.synchronize()
    ms_pair = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))
    # the same launches back to back under one event pair: the figure rocprofv3's per-kernel average agrees with
    # (an event pair per launch adds ~25 us of marker serialisation to a 0.24 ms kernel)
    # three bursts of reps / 3 launches, the fastest one counts: a burst during which the host enqueues more slowly than the
    # 0.17 ms kernel runs (one was seen at 0.29 ms per launch right after 10 GB of buffers had been released) measures the
    # host, not the kernel; all bursts are reported
    bursts = []
    for _ in range(3):
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(reps // 3):
            api.sweep_device(model, n_vox, b, y, params, cost, g, h, device.index, stream)
        eb.record()
        torch.cuda.synchronize()
        bursts.append(ea.elapsed_time(eb) / (reps // 3))
    ms = min(bursts)
    bytes_per = (n_b + n + ntri + n + 1) * 4
    ach = bytes_per * n_vox / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "sweep_kernel<tri_reduced,f32>", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": None if n_vox_override else pmc_traffic("sweep_", "sweep"),
            "algorithmic_bytes_per_launch": bytes_per * n_vox, "algorithmic_bytes_per_voxel": bytes_per,
            "kernel_ms_avg": ms, "burst_ms_avg": bursts, "per_launch_event_pair_ms_avg": ms_pair,
            "voxel_sweeps_per_s": n_vox / (ms * 1e-3), "dtype": "f32"}


WORKLOAD_TEXT = {
    "triexp": "triexp bounded LM (SciPy-TRF parity), 256x256x64x32, fp64",
    "biexp": "biexp bounded LM, 128x128x32x24, fp64",
    "mono": "monoexp curvefit, 32x32x1x16, fp64",
    "nnls": "NNLS reg_order=2 mu=0.02 n_bins=250, 256x256x64x32, fp64",
}


NOISE_SWEEP_SIGMAS = (0.0, 0.01, 0.05)
NOISE_SWEEP_VOXELS = 1 << 20      # NNLS passes (~120 ms each)
NOISE_SWEEP_VOXELS_C3 = 1 << 22   # curve-fit passes: the whole C3 volume -- a pass ends in a straggler tail of 4-9 ms whatever its size,
                                  # so a 2^20-voxel pass (9 ms + tail) would understate the rate the headline is quoted at


def noise_sweep(device, torch, jac="fd"):
    """Data dependence of both rates: the benchmark's volume re-drawn at sigma = 0 / 1 % / 5 % (same seed, same ground truth) --
    the whole volume for the curve fit, its first 2^20 voxels for the NNLS --, each fit device resident, one warm-up and the best
    of two passes (~0.4 s of curve fits, ~1.1 s of NNLS solves).  voxels/s."""
    out = {"sigma": list(NOISE_SWEEP_SIGMAS), "c3_voxels": NOISE_SWEEP_VOXELS_C3, "c4_voxels": NOISE_SWEEP_VOXELS, "c3_voxels_per_s": [], "c4_voxels_per_s": []}
    for key, nvx, make in (("c3_voxels_per_s", NOISE_SWEEP_VOXELS_C3, lambda s: CurvefitLeg("triexp", device, jac, True, NOISE_SWEEP_VOXELS_C3, (0, NOISE_SWEEP_VOXELS_C3), sigma=s)),
                           ("c4_voxels_per_s", NOISE_SWEEP_VOXELS, lambda s: NnlsLeg(device, NOISE_SWEEP_VOXELS, (0, NOISE_SWEEP_VOXELS), sigma=s))):
        for s in NOISE_SWEEP_SIGMAS:
            leg = make(s)
            leg.step()
            torch.cuda.synchronize()
            best = float("inf")
            for _ in range(2):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                leg.step()
                b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) * 1e-3)
            out[key].append(nvx / best)
            del leg
            torch.cuda.empty_cache()
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch(args, argv)  # nothing above has touched the GPU (torch is not even imported yet)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; run `python bench.py --gpus {args.gpus}` "
                         f"(it spawns the ranks) or torch.distributed.run --nproc-per-node {args.gpus}\n")
        return 2
    if args.fail_rank == rank:
        return 3
    n_total, rows = volume_rows(args.workload, args, rank, world)
    if args.launch_check:
        # the launcher, the row split and (N > 1) the CPU slice a rank would pin itself to -- without touching the GPU: the N = 8
        # the driver uses can be rehearsed anywhere (a GPU box admits at most six processes on its card)
        aff = pin_rank_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), os.environ.get("PNX_BENCH_SHARE_GPU") == "1") if world > 1 else None
        print(json.dumps({"rank": rank, "world": world, "local_rank": local, "rows": rows, "n_vox_total": n_total, "affinity": aff}), flush=True)
        time.sleep(float(os.environ.get("PNX_BENCH_LAUNCH_HOLD", "0")))  # launcher test: a rank that would wait for its peers
        return 0

    # stdout carries exactly ONE JSON line: whatever libraries print there (gloo / RCCL connection notes, ...) is sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # N > 1: this rank's threads (HIP's, the library's copy / page-touch helpers) stay on the NUMA node of its GPU; set before
    # anything touches the GPU so that every later thread inherits it.  N = 1 keeps the box's own CPU share untouched.
    affinity = None
    if world > 1:
        affinity = pin_rank_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), os.environ.get("PNX_BENCH_SHARE_GPU") == "1")

    import torch

    from pyneapple_amd import _lib
    from pyneapple_amd.sharding import shard_range

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # rehearsal knobs (one-GPU box): PNX_BENCH_SHARE_GPU=1 puts every rank on device 0, PNX_BENCH_BACKEND=gloo carries the
    # barrier / max-reduce over the CPU -- the driver's multi-GPU runs use neither
    share = os.environ.get("PNX_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("PNX_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL; only used for the barrier and the max-reduce
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()
    _lib.load()

    want_pcov = not args.no_pcov
    if args.workload == "nnls":
        leg = NnlsLeg(device, n_total, rows)
    else:
        leg = CurvefitLeg(args.workload, device, args.jac, want_pcov, n_total, rows)
    dt, kernel_ms = timed(leg, args.steps, args.warmup, world, dist, torch)
    value = n_total * args.steps / dt
    k_avg = float(np.mean(kernel_ms)) * 1e-3
    achieved = leg.bytes_per_voxel * leg.n_vox / k_avg / 1e9
    full_c3 = args.workload == "triexp" and args.jac == "fd" and not args.voxels and world == 1
    text = WORKLOAD_TEXT[args.workload]
    if args.workload != "nnls":
        text += ", FD Jacobian" if args.jac == "fd" else ", analytic Jacobian"
    out = {
        "metric": METRIC, "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": leg.dtype, "data": "synthetic",
        "config": {"workload": text, "residency": "device-resident: the volume is in HBM when the timed region starts; "
                                                  "PCIe-inclusive rate in host_mode",
                   "voxels_total": n_total, "voxels_per_rank": [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)],
                   "jac": args.jac if args.workload != "nnls" else None,
                   "pcov": want_pcov if args.workload != "nnls" else None,
                   "parallelism": f"one volume, contiguous voxel shards x{world}, no collective on the data path",
                   "full_size": not args.voxels},
        "roofline": {"bound": "hbm", "kernel": leg.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (pmc_traffic(CURVEFIT_C3_KERNEL, "curvefit") if full_c3 else
                                 nnls_traffic(leg.n_vox) if (args.workload == "nnls" and not args.voxels and world == 1) else None),
                     "algorithmic_bytes_per_launch": leg.bytes_per_voxel * leg.n_vox,
                     "algorithmic_bytes_per_voxel": leg.bytes_per_voxel, "kernel_ms_avg": k_avg * 1e3,
                     "note": "whole-fit kernels are fp64-VALU bound, not HBM bound (DESIGN.md section 4); rank 0's kernel; "
                             "kernel_ms_avg covers the K one-at-a-time launches of the timed region -- the launches of the "
                             "pipelined leg overlap in pairs and last about twice as long each (profiles: --no-pipelined)"},
        "check": leg.check(),
    }
    fl = pmc_flops(CURVEFIT_C3_KERNEL, "curvefit") if (args.workload == "triexp" and args.jac == "fd") else None
    if fl:  # the bound that actually applies: fp64 VALU issue (flop count from the committed PMC pass, time live)
        tf = fl["fp64_flop_per_voxel_issued"] * leg.n_vox / k_avg / 1e12
        out["roofline"]["valu_f64"] = {"achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
                                       "fp64_flop_per_voxel_issued": fl["fp64_flop_per_voxel_issued"],
                                       "lane_utilisation": fl["lane_utilisation"],
                                       # share of a SIMD's cycles in which the VALU issues (PMC pass: SQ_ACTIVE_INST_VALU /
                                       # SQ_WAVE_CYCLES per wave, x resident waves per SIMD): the issue roof this kernel lives under
                                       "valu_issue_busy_per_wave": fl.get("valu_issue_busy"), "waves_per_simd": 1,
                                       "simd_valu_busy": min(1.0, fl.get("valu_issue_busy", 0.0) * 1),
                                       "fp64_share_of_valu_instructions": fl.get("fp64_share_of_valu_instructions"),
                                       "source": fl["source"], "source_id": fl["source_id"]}
    if args.workload != "nnls" and not args.no_pipelined:
        out["pipelined"] = leg.pipelined(max(args.steps, 4), world, dist)  # every rank takes part (barriers)
        # `value` is the strong-scaling figure north_star asks for: ONE volume, one pass at a time, so at N ranks it ends in the
        # straggler tail of an N-th of the volume.  `throughput` is what N ranks sustain on a queue of independent volumes
        # (every rank keeps two passes of its shard in flight; same barriers, max over ranks): it separates "the tail of a
        # short shard" from "the ranks interfere" when the two are compared over N
        out["throughput"] = {"value": out["pipelined"]["value"], "unit": "voxels/s", "n_gpus": world,
                             "mode": "two passes in flight per rank on its shard; voxels of the whole volume x passes / max over ranks"}
    solo = rank == 0 and world == 1
    if solo and not args.no_host_mode:
        out["host_mode"] = leg.host_mode()
        if args.workload != "nnls":
            out["host_mode_f32"] = leg.host_mode_f32()
    if world > 1 and not args.no_host_mode:  # every rank takes part (barriers)
        out["host_mode"] = host_mode_ranks(leg, dist, world, torch, n_total)
    if solo and not args.no_cpu_baseline:
        out["cpu_baseline"] = leg.cpu_baseline()
    if args.workload == "triexp" and not args.no_secondary:
        del leg
        torch.cuda.empty_cache()
        n2, rows2 = volume_rows("nnls", args, rank, world)
        leg2 = NnlsLeg(device, n2, rows2)
        steps2 = 3  # a C4 pass is ~0.45 s: three timed passes whatever --steps says (min / median of the passes are reported beside the mean)
        dt2, k2 = timed(leg2, steps2, 1 if args.warmup else 0, world, dist, torch)
        k2avg = float(np.mean(k2)) * 1e-3
        ach2 = leg2.bytes_per_voxel * leg2.n_vox / k2avg / 1e9
        sec = {"workload": WORKLOAD_TEXT["nnls"], "value": n2 * steps2 / dt2,
               "unit": "voxels/s", "steps": steps2, "ms_per_step": dt2 / steps2 * 1e3,
               "ms_per_step_min": float(np.min(k2)), "ms_per_step_median": float(np.median(k2)), "check": leg2.check(),
               "roofline": {"bound": "hbm", "kernel": NNLS_KERNEL, "achieved": ach2, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach2 / HBM_PEAK_GBS,
                            "traffic": nnls_traffic(leg2.n_vox) if (not args.voxels and world == 1) else None,
                            "algorithmic_bytes_per_voxel": leg2.bytes_per_voxel, "kernel_ms_avg": k2avg * 1e3,
                            "note": "VALU issue (about 75 % busy at 12 waves per CU) and per-iteration latency bound, not HBM bound (DESIGN.md section 4.3)"}}
        fl2 = pmc_flops(NNLS_KERNEL, "nnls")
        if fl2:
            tf2 = fl2["fp64_flop_per_voxel_issued"] * leg2.n_vox / k2avg / 1e12
            sec["roofline"]["valu_f64"] = {"achieved": tf2, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf2 / FP64_PEAK_TFLOPS,
                                           "fp64_flop_per_voxel_issued": fl2["fp64_flop_per_voxel_issued"],
                                           "fp64_share_of_valu_instructions": fl2["fp64_share_of_valu_instructions"],
                                           "lane_utilisation": fl2["lane_utilisation"],
                                           "valu_issue_busy_per_wave": fl2.get("valu_issue_busy"), "waves_per_simd": 3,
                                           "simd_valu_busy": min(1.0, fl2.get("valu_issue_busy", 0.0) * 3),
                                           "source": fl2["source"], "source_id": fl2["source_id"]}
        if solo and not args.no_host_mode:
            sec["host_mode"] = leg2.host_mode()
        if world > 1 and not args.no_host_mode:
            sec["host_mode"] = host_mode_ranks(leg2, dist, world, torch, n2)
        if solo and not args.no_cpu_baseline:
            sec["cpu_baseline"] = leg2.cpu_baseline()
        out["secondary"] = sec
        del leg2
        torch.cuda.empty_cache()
    if args.workload == "triexp" and not args.no_secondary and rank == 0:
        # The two sub-millisecond roofline probes.  Whatever runs right after gigabytes of device memory have been released is
        # slow for a few hundred milliseconds on this box (measured both ways round: the sweep's launch burst at 0.29 ms per
        # enqueue right after the NNLS leg was freed; the host-mode leg at 68-77 ms instead of 51 when it followed the probes'
        # buffers): the frees are given a second to settle, and the sweep reports the best of three bursts.
        torch.cuda.synchronize()
        time.sleep(1.0)
        out["roofline_sweep"] = sweep_roofline(device, torch, args.voxels)
        out["roofline_mfma"] = mfma_roofline(device, torch)
        if world == 1 and not args.no_noise_sweep:
            out["noise_sweep"] = noise_sweep(device, torch, args.jac)
    if world > 1:
        aff = [None] * world
        dist.all_gather_object(aff, affinity)
        out["affinity"] = aff
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(finalize(out, args.workload)) + "\n").encode())
    return 0


HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data")
SCALAR_KEYS = ("nnls_voxels_per_s", "nnls_ms_per_step", "c3_host_voxels_per_s", "c4_host_voxels_per_s", "throughput_voxels_per_s")


def finalize(out: dict, workload: str = "triexp") -> dict:
    """The bench line in the order a truncating reader needs it: the contract's keys, then the five scalars of the metric's
    other half (resident NNLS rate, both PCIe-inclusive rates, the two-in-flight throughput), the noise sweep of both rates and
    the min / median of the NNLS passes, then roofline / cpu_baseline,
    then the detail objects; every prose string (`note`, `workload`, `mode`) moves to ONE trailing `notes` object keyed by its
    path.  Pure function (tests/test_bench_line.py runs it on a committed line)."""
    out = json.loads(json.dumps(out))  # deep copy, JSON types only
    sec = out.get("secondary") or {}
    nn = out if workload == "nnls" else sec
    scal = {
        "nnls_voxels_per_s": nn.get("value"), "nnls_ms_per_step": nn.get("ms_per_step"),
        "c3_host_voxels_per_s": (out.get("host_mode") or {}).get("value") if workload == "triexp" else None,
        "c4_host_voxels_per_s": (nn.get("host_mode") or {}).get("value"),
        "throughput_voxels_per_s": (out.get("throughput") or {}).get("value"),
    }
    notes = {}

    def strip(d, path):
        for k in list(d.keys()):
            v = d[k]
            if isinstance(v, dict):
                strip(v, path + [k])
            elif k in ("note", "workload", "mode") and isinstance(v, str) and path and path != ["config"]:
                notes[".".join(path + [k])] = d.pop(k)

    strip(out, [])
    final = {k: out[k] for k in HEAD_KEYS if k in out}
    final.update(scal)
    if "noise_sweep" in out:  # the data dependence of both rates belongs next to them (inside the first 2 000 bytes)
        final["noise_sweep"] = out["noise_sweep"]
    if nn.get("ms_per_step_min") is not None:
        final["nnls_ms_per_step_min"], final["nnls_ms_per_step_median"] = nn["ms_per_step_min"], nn["ms_per_step_median"]
    for k in ("roofline", "cpu_baseline", "config", "check"):
        if k in out:
            final[k] = out[k]
    for k, v in out.items():
        if k not in final:
            final[k] = v
    final["notes"] = notes
    return final


def nnls_traffic(n_vox):
    """PMC HBM bytes of one NNLS step (per-launch figure x launches per step; see profiles/*_traffic.json:_nnls_launch_voxels)."""
    t, _, _ = _profile("traffic", "nnls")
    if not t:
        return None
    per = int(t.get("_nnls_launch_voxels", 1 << 20))
    v = pmc_traffic(NNLS_KERNEL, "nnls")
    return None if v is None else v * ((n_vox + per - 1) // per)


if __name__ == "__main__":
    sys.exit(main())
