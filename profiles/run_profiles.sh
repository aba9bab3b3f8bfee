#!/bin/bash
# One profiling round on the GPU box (from the repo root): kernel trace, PMC passes, traffic passes -> gpurun_out/<tag>/...
# usage: bash profiles/run_profiles.sh <tag> <part>   part: kt | cf | nnls | traffic
set -e
tag=$1; part=$2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case $part in
  kt)
    python3 bench.py > $out/bench.json 2> $out/bench.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py --no-cpu-baseline --no-host-mode --no-pipelined --no-noise-sweep > $out/kt.log 2>&1
    ;;
  cf) bash profiles/pmc_curvefit.sh $out/pmc_cf ;;
  nnls) bash profiles/pmc_nnls.sh $out/pmc_nnls ;;
  traffic) bash profiles/pmc_traffic.sh $out/traffic ;;
esac
echo "$part done"
