#!/bin/bash
# One counter pass (instruction counts and VALU activity) for an NNLS kernel variant.  usage: PNX_LIB=... bash profiles/pmc_quick.sh <outdir>
set -e
out=$1; shift
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $out/p1 -- python3 profiles/nnls_run.py > $out/p1.log 2>&1
python3 profiles/pmc_summary.py $out nnls_blk > $out/summary.txt
