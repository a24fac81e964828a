#!/bin/bash
# Profile of the SGMSE+ use_amp score network (channels-last fp16 path) on the GPU box, run through
# gpurun from the repo root:  bash tools/profile_sgmse.sh r02
#   * rocprofv3 --kernel-trace --stats of 2 x 6 network evaluations at batch 1 and 8 (tools/prof_sgmse.py)
#   * HBM traffic per kernel launch at batch 8 (FETCH_SIZE / WRITE_SIZE, separate PMC passes)
#   * tools/bin/convbench2: the 3x3 convolution alone -- check against a CPU loop, timings,
#     compile-time ablations, in-kernel cycle stamps (build them first: see the header of
#     tools/convbench2.hip; the binaries travel with the snapshot)
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sg$b -o sg -- python3 $REPO/tools/prof_sgmse.py $b > /dev/null 2>&1
  cp $(find /tmp/sg$b -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_sgmse_b${b}_kernel_stats.csv
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sgpmc_f -o f -- python3 $REPO/tools/prof_sgmse.py 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/sgpmc_w -o w -- python3 $REPO/tools/prof_sgmse.py 8 > /dev/null 2>&1
python3 $REPO/tools/pmc_traffic.py /tmp/sgpmc_f /tmp/sgpmc_w $OUT/${TAG}_sgmse_b8_pmc_hbm_traffic.json
cd $REPO/tools/bin
{
  echo "== convbench2 1 (checks + batch-1 timings)"; ./convbench2 1
  echo "== convbench2 8"; ./convbench2 8 | grep time
  for a in 16 20 48 19 24 51; do echo "== ablation CN_ABL=$a (1 no patch DMA, 2 no weight DMA, 4 no MFMA, 8 no transform, 16 no stores, 32 no fragment reads)"; ./convbench2_$a 8 | grep time | head -2; done
  echo "== in-kernel stamps (CN_DIAG), batch 8"; ./convbench2_diag 8 | grep -v check | head -24
  echo "== previous kernel (conv_mfma.hip, fp32 NCHW activations), batch 8"; ./convbench 8
} > $OUT/${TAG}_conv_nhwc_bench.txt 2>&1
tail -5 $OUT/${TAG}_conv_nhwc_bench.txt
