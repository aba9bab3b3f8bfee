#!/bin/bash
# HBM traffic of an NNLS kernel variant (FETCH_SIZE / WRITE_SIZE in separate passes).  usage: bash profiles/pmc_traffic_nnls.sh <outdir>
set -e
out=$1; shift
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 profiles/nnls_run.py > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 profiles/nnls_run.py > $out/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/l2 -- python3 profiles/nnls_run.py > $out/l2.log 2>&1
python3 profiles/traffic_summary.py $out > $out/traffic.json
python3 profiles/pmc_summary.py $out/l2 nnls_blk > $out/l2.txt
