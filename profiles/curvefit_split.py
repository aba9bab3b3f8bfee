#!/usr/bin/env python3
"""Experiment: one C3 volume fitted as geometrically shrinking pieces on two alternating streams, so that the straggler tail
of one piece overlaps the bulk of the next (the persistent kernel runs one wave per SIMD: a second launch fills the SIMDs as
the first one's waves retire)."""
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
popt = torch.empty((nn, n), dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); nf = torch.empty(n, dtype=torch.int32, device=dev); cost = torch.empty(n, dtype=torch.float64, device=dev)
opts = api.make_opts("tri_reduced", 32, [], False, False, 250, 1e-8, 1e-8, 1e-8, "fd", 0, 0.0, 0.0)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

def run(fracs):
    edges = np.concatenate([[0], np.cumsum(np.round(np.array(fracs) * n / 64).astype(np.int64) * 64)])
    edges[-1] = n
    main = torch.cuda.current_stream()
    for s_ in streams:
        s_.wait_stream(main)
    for k in range(len(fracs)):
        a, e = int(edges[k]), int(edges[k + 1])
        s_ = streams[k % 2]
        # per-piece outputs: popt is parameter-major (nn, n): write into a piece-local buffer (nn, e - a)
        po = pieces[k]
        api.curvefit_device(opts, e - a, b, y[a:e], p0, lo, hi, None, po, None, st[a:e], nf[a:e], cost[a:e], 0, s_.cuda_stream)
    for s_ in streams:
        main.wait_stream(s_)

for fracs in ([1.0], [0.8, 0.2], [0.8, 0.17, 0.03], [0.7, 0.22, 0.06, 0.02], [0.5, 0.25, 0.125, 0.0625, 0.0625], [0.6, 0.25, 0.1, 0.04, 0.01],
              [0.25, 0.25, 0.25, 0.25]):
    edges = np.concatenate([[0], np.cumsum(np.round(np.array(fracs) * n / 64).astype(np.int64) * 64)]); edges[-1] = n
    pieces = [torch.empty((nn, int(edges[k + 1] - edges[k])), dtype=torch.float64, device=dev) for k in range(len(fracs))]
    for _ in range(2):
        run(fracs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        run(fracs)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    print(fracs, f"{ms:.3f} ms  {n / ms / 1e3:.1f} M voxels/s", flush=True)
