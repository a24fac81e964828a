set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BRV_DCCRN_WGRAD_SIDE=0
for v in 1 0; do
  export BRV_DCCRN_BF16_ACT=$v
  rm -rf /tmp/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o s -- python3 $REPO/tools/prof_dccrn.py 1 > /dev/null 2>&1
  cp $(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1) $OUT/dccrn_act$v.csv
done
