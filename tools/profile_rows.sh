#!/bin/bash
# Kernel statistics and PMC HBM traffic of the widened rows on the GPU box (through gpurun from the repo
# root):  bash tools/profile_rows.sh r03
#   fp32 Conv-TasNet training (tools/prof_ctn_f32.py), DCCRN training under use_amp and in fp32 (tools/prof_dccrn.py 1 / 0),
#   TF-GridNet training under use_amp (tools/prof_train.py), SGMSE+ use_amp inference at batch 1 and 8
#   (tools/prof_sgmse.py): rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in separate
#   --pmc passes (never combined with other trace domains); tools/rows_roofline.py turns each triple into
#   a `roofline` object of the row's dominant kernel (measured traffic per launch / average duration).
set -u
TAG=${1:-r04}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# ROWS="dccrn_bf16 sgmse_b8" bash tools/profile_rows.sh r05: only those rows (their roofline lines are appended)
run() {   # name, program, args...
  local name=$1; shift
  if [ -n "${ROWS:-}" ] && [[ " $ROWS " != *" $name "* ]]; then return; fi
  rm -rf /tmp/rw_$name /tmp/rwf_$name /tmp/rww_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rw_$name -o s -- python3 "$@" > /dev/null 2>&1
  cp $(find /tmp/rw_$name -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rows_${name}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/rwf_$name -o f -- python3 "$@" > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/rww_$name -o w -- python3 "$@" > /dev/null 2>&1
  python3 $REPO/tools/pmc_traffic.py /tmp/rwf_$name /tmp/rww_$name $OUT/${TAG}_rows_${name}_pmc_hbm_traffic.json > /dev/null
  python3 $REPO/tools/rows_roofline.py $name $OUT/${TAG}_rows_${name}_kernel_stats.csv $OUT/${TAG}_rows_${name}_pmc_hbm_traffic.json >> $OUT/${TAG}_rows_roofline.json
}
[ -z "${ROWS:-}" ] && rm -f $OUT/${TAG}_rows_roofline.json
run ctn_fp32 $REPO/tools/prof_ctn_f32.py
# (per-kernel durations of the DCCRN use_amp step are taken IN ORDER on one stream: with the parameter gradients on
# the side stream, overlapping kernels share the chip and each reads up to twice its own duration)
export BRV_DCCRN_WGRAD_SIDE=0
run dccrn_bf16 $REPO/tools/prof_dccrn.py 1
unset BRV_DCCRN_WGRAD_SIDE
run dccrn_fp32 $REPO/tools/prof_dccrn.py 0
run tfgridnet_bf16 $REPO/tools/prof_train.py tfgridnet 1 3
run sgmse_b1 $REPO/tools/prof_sgmse.py 1
run sgmse_b8 $REPO/tools/prof_sgmse.py 8
cat $OUT/${TAG}_rows_roofline.json | cut -c1-600
