#!/usr/bin/env python3
"""the plugin's n_gpus = N path from numpy arrays -- one host thread per device through the blocking
ABI, each call with its upload / download / page-touch helpers -- rehearsed on one card (PNX_SHARE_DEVICE=1: every shard goes to
device 0).  Prints the fit time, the peak number of threads of the process during the call and the helper budget in use."""
import os as _os; _os.environ.setdefault("PNX_ENABLE_TEST_HOOKS", "1")  # this script drives developer switches of the library (include/pnx.h, "Environment")
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyneapple_amd import synth
from pyneapple_amd.models import TriExpModel
from pyneapple_amd.solvers import HipCurveFitSolver, HipNNLSSolver

os.environ["PNX_SHARE_DEVICE"] = "1"
n = 256 * 256 * 64
b, y, _ = synth.make_numpy("tri_reduced", n, 32, sigma=0.01)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")


def threads_now():
    return len(os.listdir("/proc/self/task"))


def timed(fn):
    peak = [threads_now()]
    stop = threading.Event()
    def sampler():
        while not stop.is_set():
            peak[0] = max(peak[0], threads_now()); time.sleep(0.001)
    th = threading.Thread(target=sampler); th.start()
    t = time.perf_counter(); r = fn(); dt = time.perf_counter() - t
    stop.set(); th.join()
    return dt, peak[0], r


base_threads = threads_now()
for label, env in (("helper budget by calls in flight (default)", {}), ("PNX_HOST_TOUCHERS=2 PNX_STREAM_OUT_THREADS=2 (round 3)", {"PNX_HOST_TOUCHERS": "2", "PNX_STREAM_OUT_THREADS": "2"}),
                   ("PNX_HOST_TOUCHERS=0", {"PNX_HOST_TOUCHERS": "0"})):
    for k in ("PNX_HOST_TOUCHERS", "PNX_STREAM_OUT_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for n_gpus in (1, 2, 3, 4):
        s = HipCurveFitSolver(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)), bounds={k: (a, c) for k, a, c in zip(names, lo, hi)}, n_gpus=n_gpus)
        s.fit(b, y)  # warm-up
        ts = []
        for _ in range(3):
            dt, peak, _ = timed(lambda: s.fit(b, y))
            ts.append(dt * 1e3)
        print(f"{label}: n_gpus={n_gpus} (one card): {min(ts):.1f} ms best of {[round(t, 1) for t in ts]}, peak threads {peak} (process idle: {base_threads})", flush=True)
        s.close()
