#!/bin/bash
# PMC passes for the curve-fit kernel (separate runs, counters only -- no trace domains).
# usage (on the GPU box, from the repo root): bash profiles/pmc_curvefit.sh <outdir> [bench args...]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-mode --no-pipelined --no-secondary $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/p1 -- python3 $ARGS > $out.p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d $out/p2 -- python3 $ARGS > $out.p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p3 -- python3 $ARGS > $out.p3.log 2>&1
