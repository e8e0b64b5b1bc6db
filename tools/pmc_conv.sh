#!/bin/bash
# Hardware counters (separate passes) for the conv micro-benchmark; run on the GPU box from the repo root.
# usage: tools/pmc_conv.sh <layer-name-filter>      (e.g. blk2.c1, blk4.c2)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { timeout -k 10 120 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py --only $3 --reps 2 > $OUT/$1.log 2>&1; }
run p1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" $1
run p2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" $1
run p3 "FETCH_SIZE GRBM_GUI_ACTIVE" $1
run p4 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $1
run p5 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" $1
run p6 "SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" $1
echo done $1
