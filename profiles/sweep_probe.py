#!/usr/bin/env python3
"""Sweep-kernel probe: python3 profiles/sweep_probe.py [reps] -> one line with the HIP-event average per launch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pyneapple_amd import _lib
_lib.load()
r = bench.sweep_roofline(torch.device("cuda", 0), torch, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 50)
print(json.dumps({k: r[k] for k in ("achieved", "frac", "kernel_ms_avg", "per_launch_event_pair_ms_avg")}), {k: v for k, v in os.environ.items() if k.startswith("PNX_")})
