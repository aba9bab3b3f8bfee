#!/usr/bin/env python3
"""Spectrum post-processing on the device: the peak kernel alone (device resident), and solve + peaks against solve with the
spectra coming back, host arrays in and out."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyneapple_amd import api, synth
bins, basis, reg = synth.nnls_matrices(32)
n = 1 << 20
dev = torch.device("cuda", 0)
_, yd = synth.make_torch_rows("tri_reduced", 0, n, 32, dev, sigma=0.01, scale=1000.0)
plan = api.NnlsPlan(basis, reg, 0)
coeff = torch.empty((n, 250), dtype=torch.float64, device=dev); rn = torch.empty(n, dtype=torch.float64, device=dev)
st = torch.empty(n, dtype=torch.int8, device=dev); it = torch.empty(n, dtype=torch.int32, device=dev)
plan.solve_device(n, yd, 250, coeff, rn, st, it, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
cuts = [(0.0008, 0.003), (0.003, 0.02), (0.02, 0.5)]
for rep in range(3):
    t = time.perf_counter(); r = api.spectrum_peaks(coeff, bins, regularized=True, cutoffs=cuts); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"peak kernel, {n} spectra device resident: {dt*1e3:.2f} ms = {n*2000/dt/1e9:.0f} GB/s of spectra", flush=True)
y = yd.cpu().numpy()
plan.solve_peaks(y[:65536], bins, cutoffs=cuts, regularized=True)
for _ in range(2):
    t = time.perf_counter(); r = plan.solve_peaks(y, bins, cutoffs=cuts, regularized=True); dt = time.perf_counter() - t
    print(f"solve_peaks (host in / peak tables out): {n/dt/1e6:.2f} M voxels/s ({dt*1e3:.0f} ms)", flush=True); del r
for _ in range(2):
    t = time.perf_counter(); r = plan.solve(y); dt = time.perf_counter() - t
    print(f"solve (host in / spectra out): {n/dt/1e6:.2f} M voxels/s ({dt*1e3:.0f} ms)", flush=True); del r
