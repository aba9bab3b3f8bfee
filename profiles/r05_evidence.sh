#!/bin/bash
# Round 5: the ONE stamped evidence set of the round, on the final sources.  Three parts, one gpurun call each
# (a call is limited to 20 minutes):
#   bash profiles/r05_evidence.sh counters   -> gpurun_out/r05/{bench.json,kt,pmc_nnls,pmc_cf,traffic}  -> profiles/make_summaries.sh gpurun_out/r05 r05
#   bash profiles/r05_evidence.sh parity     -> gpurun_out/r05_parity_large.json, r05_fuzz_nnls.json (+ cases), r05_fuzz_curvefit.json
#   bash profiles/r05_evidence.sh hostfuzz   -> gpurun_out/r05_fuzz_nnls_host.json, r05_fuzz_stream_vs_ring.json, r05_fuzz_nnls_wide.json (+ cases)
set -e
part=$1
export PNX_ENABLE_TEST_HOOKS=1   # the fuzzers drive chunk sizes; bench.py runs below unset it again
case $part in
  counters)
    unset PNX_ENABLE_TEST_HOOKS
    for p in kt nnls cf traffic; do bash profiles/run_profiles.sh r05 $p; done
    ;;
  parity)
    timeout -k 10 500 python profiles/parity_large.py --c3 1048576 --c4 524288 --json gpurun_out/r05_parity_large.json > gpurun_out/r05_parity_large.log 2>&1
    tail -2 gpurun_out/r05_parity_large.log
    timeout -k 10 300 python tests/fuzz_gpu_vs_oracle_nnls.py 800 46 --json gpurun_out/r05_fuzz_nnls.json > gpurun_out/r05_fuzz_nnls_cases.txt 2>&1 || true
    tail -1 gpurun_out/r05_fuzz_nnls_cases.txt
    timeout -k 10 400 python tests/fuzz_gpu_vs_oracle.py 800 46 --json gpurun_out/r05_fuzz_curvefit.json > gpurun_out/r05_fuzz_curvefit.log 2>&1 || true
    tail -1 gpurun_out/r05_fuzz_curvefit.log
    ;;
  hostfuzz)
    timeout -k 10 300 python tests/fuzz_nnls_host.py 100 46 --json gpurun_out/r05_fuzz_nnls_host.json > gpurun_out/r05_fuzz_nnls_host.log 2>&1 || true
    tail -1 gpurun_out/r05_fuzz_nnls_host.log
    timeout -k 10 400 python tests/fuzz_stream_vs_ring.py 600 46 --json gpurun_out/r05_fuzz_stream_vs_ring.json > gpurun_out/r05_fuzz_stream_vs_ring.log 2>&1 || true
    tail -1 gpurun_out/r05_fuzz_stream_vs_ring.log
    timeout -k 10 300 python tests/fuzz_gpu_vs_oracle_nnls.py 200 81 --wide --json gpurun_out/r05_fuzz_nnls_wide.json > gpurun_out/r05_fuzz_nnls_wide_cases.txt 2>&1 || true
    tail -1 gpurun_out/r05_fuzz_nnls_wide_cases.txt
    ;;
esac
echo "$part done"
