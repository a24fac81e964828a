#!/bin/bash
# SQ-counter passes over a widened row (as tools/profile_sq.sh does for the headline step): --pmc only with
# --kernel-trace, 8 SQ slots per pass. usage (through gpurun, from the repo root):
#   bash tools/profile_sq_rows.sh r04 dccrn_bf16 $PWD/tools/prof_dccrn.py 1   (absolute path: the passes run from /tmp)
set -u
TAG=$1; NAME=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqr_a /tmp/sqr_b /tmp/sqr_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/sqr_a -o a -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d /tmp/sqr_b -o b -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/sqr_c -o c -- python3 "$@" > /dev/null 2>&1
python3 $REPO/tools/sq_counters.py $OUT/${TAG}_rows_${NAME}_sq_counters.json /tmp/sqr_a /tmp/sqr_b /tmp/sqr_c | tee $OUT/${TAG}_rows_${NAME}_sq_counters.txt | head -30
