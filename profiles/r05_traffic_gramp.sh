#!/bin/bash
# Round 5: traffic behind the L2 of the NNLS block kernel by the Gram-form threshold (final kernel; variants -DPNX_BLK_GRAMP=16 / 20
# beside the product's 24): FETCH_SIZE / WRITE_SIZE in separate passes, `bench.py --workload nnls`.  r05_nnls_experiments.md section 17.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r05g_*
ARGS="bench.py --workload nnls --steps 1 --warmup 1 --no-cpu-baseline --no-host-mode"
for v in g16 g20 product; do
  lib=""; [ "$v" != product ] && lib=$PWD/pyneapple_amd/libpnx_hip.$v.so
  export PNX_LIB=$lib; [ -z "$lib" ] && unset PNX_LIB
  out=gpurun_out/r05g_$v
  mkdir -p $out
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $ARGS > $out.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $ARGS > $out.write.log 2>&1
  echo "== $v"; python3 profiles/traffic_summary.py $out | python3 -c "import json,sys; d=json.load(sys.stdin); [print(k[:40], {a: round(b/1e9,2) for a,b in v.items() if a!='launches_averaged'}) for k,v in d.items() if 'blk' in k]"
  grep -h '"value"' $out.write.log | head -1 | cut -c1-200
done
