#!/usr/bin/env python3
"""Gram-step (nnls_aty_mfma_kernel) probe: python3 profiles/aty_probe.py [reps] -> one line with the HIP-event average."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyneapple_amd import _lib
_lib.load()
r = bench.mfma_roofline(torch.device("cuda", 0), torch, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 50)
print(json.dumps({k: r[k] for k in ("achieved", "frac", "kernel_ms_avg", "hbm_GBps")}), {k: v for k, v in os.environ.items() if k.startswith("PNX_")})
