#!/usr/bin/env python3
"""Time the *reference's* own solver path, serial and with its joblib multi-processing (survey container only).

TEST / MEASUREMENT INFRASTRUCTURE ONLY: imports darksim33/Pyneapple from /root/reference/src the way
oracle/gen_golden.py does.  joblib's loky workers are fresh interpreters, so the two stand-ins for the absent
`loguru` / `cv2` modules are written as files into a temporary directory that is put on
PYTHONPATH for the children; nothing of it is kept.  Prints voxels/s for the BASELINE workloads (triexp C3 inputs,
NNLS C4 inputs) on a bounded sample; the numbers go into DESIGN.md section 5 by hand.

usage: python3 oracle/time_reference.py [n_triexp_voxels] [n_nnls_voxels] [n_pools]
"""
from __future__ import annotations

import os
import sys
import tempfile
import time

import numpy as np

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n_nnls = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    n_pools = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
    stub = tempfile.mkdtemp(prefix="pnx_refstub_")
    with open(os.path.join(stub, "loguru.py"), "w") as f:
        f.write("class _N:\n    def __getattr__(self, n):\n        return lambda *a, **k: None\nlogger = _N()\n")
    with open(os.path.join(stub, "cv2.py"), "w") as f:
        f.write("INTER_LINEAR = 1\nINTER_CUBIC = 2\n")
    for name in ("nibabel", "h5py"):
        try:
            __import__(name)
        except Exception:
            open(os.path.join(stub, name + ".py"), "w").close()
    os.environ["PYTHONPATH"] = os.pathsep.join([stub, REF_SRC, os.environ.get("PYTHONPATH", "")])
    os.environ.setdefault("PYNEAPPLE_QUIET", "1")
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path[:0] = [stub, REF_SRC, os.path.join(HERE, "..")]
    import warnings

    from pyneapple import CurveFitSolver, NNLSModel, NNLSSolver, TriExpModel

    from pyneapple_amd import synth

    b, y, _ = synth.make_numpy("tri_reduced", n_tri, 32, sigma=0.01)
    names, p0, lo, hi = synth.shared_arrays("tri_reduced")
    kw = dict(model=TriExpModel(), max_iter=250, tol=1e-8, p0=dict(zip(names, p0)),
              bounds={k: (a, c) for k, a, c in zip(names, lo, hi)})
    print(f"cpus available: {len(os.sched_getaffinity(0))}, n_pools = {n_pools}", flush=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ns = max(64, n_tri // 8)
        t = time.perf_counter(); CurveFitSolver(**kw).fit(b, y[:ns]); dt = time.perf_counter() - t
        print(f"reference CurveFitSolver, triexp C3 inputs, serial: {ns / dt:.1f} voxels/s ({ns} voxels, {dt:.1f} s)", flush=True)
        s = CurveFitSolver(multi_threading=True, n_pools=n_pools, **kw)
        s.fit(b, y[:64])  # start the worker pool
        t = time.perf_counter(); s.fit(b, y); dt = time.perf_counter() - t
        print(f"reference CurveFitSolver, triexp C3 inputs, joblib n_pools={n_pools}: {n_tri / dt:.1f} voxels/s ({n_tri} voxels, {dt:.1f} s)", flush=True)
        cfg = synth.NNLS_CFG
        model = NNLSModel(d_range=tuple(cfg["d_range"]), n_bins=cfg["n_bins"])
        yn = y[:n_nnls] * 1000.0
        ns = max(16, n_nnls // 8)
        t = time.perf_counter(); NNLSSolver(model=model, reg_order=cfg["reg_order"], mu=cfg["mu"], max_iter=250).fit(b, yn[:ns]); dt = time.perf_counter() - t
        print(f"reference NNLSSolver, C4 inputs, serial: {ns / dt:.1f} voxels/s ({ns} voxels, {dt:.1f} s)", flush=True)
        s = NNLSSolver(model=model, reg_order=cfg["reg_order"], mu=cfg["mu"], max_iter=250, multi_threading=True, n_pools=n_pools)
        s.fit(b, yn[:16])
        t = time.perf_counter(); s.fit(b, yn); dt = time.perf_counter() - t
        print(f"reference NNLSSolver, C4 inputs, joblib n_pools={n_pools}: {n_nnls / dt:.1f} voxels/s ({n_nnls} voxels, {dt:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
