#!/bin/bash
# HBM traffic of the bench kernels from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), counters only (no trace domains).
# usage: bash profiles/pmc_traffic.sh <outdir>
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-mode --no-pipelined --no-noise-sweep $@"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $ARGS > $out.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $ARGS > $out.write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- python3 $ARGS > $out.mfma.log 2>&1
