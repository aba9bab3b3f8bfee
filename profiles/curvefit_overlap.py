#!/usr/bin/env python3
"""Experiment: consecutive whole-volume fits on two alternating streams (independent volumes in a pipeline): does the straggler
tail of one launch overlap the bulk of the next?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyneapple_amd import api, synth, _lib
_lib.load()
dev = torch.device("cuda", 0)
names, p0, lo, hi = synth.shared_arrays("tri_reduced")
nn = len(names)
n = 1 << 22
b, y = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01)
outs = []
for k in range(2):
    outs.append((torch.empty((nn, n), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int8, device=dev),
                 torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float64, device=dev)))
opts = api.make_opts("tri_reduced", 32, [], False, False, 250, 1e-8, 1e-8, 1e-8, "fd", 0, 0.0, 0.0)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
for nstreams in (1, 2):
    for reps in (2, 8):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        main = torch.cuda.current_stream()
        e0.record()
        for s_ in streams:
            s_.wait_stream(main)
        for k in range(reps):
            po, st, nf, cost = outs[k % 2]
            api.curvefit_device(opts, n, b, y, p0, lo, hi, None, po, None, st, nf, cost, 0, streams[k % nstreams].cuda_stream)
        for s_ in streams:
            main.wait_stream(s_)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{nstreams} stream(s), {reps} launches: {ms:.3f} ms per volume  {n / ms / 1e3:.1f} M voxels/s", flush=True)
