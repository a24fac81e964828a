#!/bin/bash
# SQ-counter passes over the headline step (one-chain: whole-batch launches, a kernel alone on the chip).
# --pmc only with --kernel-trace; 8 SQ slots per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (through gpurun, from the repo root): bash tools/profile_sq.sh r04
set -u
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BRV_CTN_STREAMS=1
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-through-trainer --no-fp32-path --no-other-configs --min-warmup-s 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/sq_a -o a -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_sq_pass_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d /tmp/sq_b -o b -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_sq_pass_b.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/sq_c -o c -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_sq_pass_c.log 2>&1
python3 $REPO/tools/sq_counters.py $OUT/${TAG}_sq_counters.json /tmp/sq_a /tmp/sq_b /tmp/sq_c | tee $OUT/${TAG}_sq_counters.txt
tail -3 $OUT/${TAG}_sq_pass_a.log | cut -c1-300
