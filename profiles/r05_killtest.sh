#!/bin/bash
# NOTE: the meeting points (-DPNX_BLK_KILLTEST=k) exist in the sources of commit f09b818 only (removed after the measurement);
# the variants were built with PNX_VARIANT=kt1 PNX_NNLS_FLAGS=-DPNX_BLK_KILLTEST=1 python -m pyneapple_amd._build (and so on).
# Round 5, VERDICT item 1(a): what lock-step costs the NNLS block kernel.  Each variant is the product's blk2 kernel with meeting
# points added (arithmetic untouched): kt1 = one s_barrier per outer iteration (all 12 waves of the CU), kt2 = two; kt3gG = one
# meeting point of G waves through an LDS counter, kt4g4 = two of four waves.
set -e
mkdir -p gpurun_out
for v in "" kt1 kt2 kt3g4 kt4g4 kt3g2 kt3g3 kt3g6; do
  lib=""; [ -n "$v" ] && lib=pyneapple_amd/libpnx_hip.$v.so
  timeout -k 10 150 python profiles/nnls_probe.py $lib >> gpurun_out/r05_killtest.txt 2>&1
  tail -1 gpurun_out/r05_killtest.txt
done
