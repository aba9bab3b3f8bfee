#!/bin/bash
# Instructions per voxel of NNLS block-kernel variants (C4 workload, 2^18 voxels, profiles/nnls_run.py under one --pmc pass per
# variant, counters only): usage  bash profiles/nnls_count.sh <name> product|<variant> ...   -> gpurun_out/<name>.txt
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  lib=""; [ "$v" != "product" ] && lib=$GRAFT_REPO_ROOT/pyneapple_amd/libpnx_hip.$v.so
  PNX_LIB=$lib timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES \
      --output-format csv -d gpurun_out/cnt_$name/$v -- python3 profiles/nnls_run.py > gpurun_out/cnt_$name.$v.log 2>&1
  python3 - gpurun_out/cnt_$name/$v $v >> gpurun_out/$name.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "nnls_blk_kernel" in row["Kernel_Name"]: acc[row["Counter_Name"]] += float(row["Counter_Value"])
nv = 2 * (1 << 18)   # warm-up + one solve
tot = sum(v for k, v in acc.items() if k.startswith("SQ_INSTS"))
print(sys.argv[2], " ".join(f"{k[9:]}={acc[k] / nv:.0f}" for k in sorted(acc) if k.startswith("SQ_INSTS")), f"sum={tot / nv:.0f}", f"wave_cycles={acc['SQ_WAVE_CYCLES'] / nv:.0f}")
PY
  tail -1 gpurun_out/$name.txt
done
