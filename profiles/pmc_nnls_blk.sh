#!/bin/bash
# Counter passes for an NNLS kernel variant on the C4 workload.  usage: PNX_LIB=... bash profiles/pmc_nnls_blk.sh <outdir>
set -e
out=$1; shift
root="$GRAFT_REPO_ROOT"; [ -z "$root" ] && root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/p1 -- python3 profiles/nnls_run.py > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $out/p2 -- python3 profiles/nnls_run.py > $out/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT --output-format csv -d $out/p3 -- python3 profiles/nnls_run.py > $out/p3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/p4 -- python3 profiles/nnls_run.py > $out/p4.log 2>&1 || true
python3 profiles/pmc_summary.py $out nnls > $out/summary.txt
