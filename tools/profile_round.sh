#!/bin/bash
# Round profile of the headline benchmark on the GPU box (run through gpurun from the repo root):
#   kernel statistics (rocprofv3 --kernel-trace --stats), the bench JSON line of that run, the
#   per-label HIP-event table, and HBM traffic per kernel from two PMC passes (FETCH_SIZE,
#   WRITE_SIZE: separate runs, --pmc never combined with other trace domains).
# usage: bash tools/profile_round.sh r02
set -u
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-through-trainer --no-fp32-path --no-other-configs --min-warmup-s 0"
# The default step runs two half-batch kernel chains concurrently (a launch's wall time then includes
# the other chain's workgroups, and a launch covers half the batch). The per-kernel numbers -- this
# statistics pass, the PMC passes and the `roofline` object of bench.py -- are those of the ONE-chain
# step (BRV_CTN_STREAMS=1): whole-batch launches, a kernel alone on the chip. The two-chain statistics
# are kept next to them.
export BRV_CTN_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o main -- python3 $REPO/bench.py $ARGS --kernel-table > $OUT/${TAG}_rocprofv3_bench_line.json 2> $OUT/${TAG}_kernel_table.txt
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats.csv
unset BRV_CTN_STREAMS
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats2 -o main -- python3 $REPO/bench.py $ARGS > $OUT/${TAG}_rocprofv3_bench_line_two_chains.json 2> /dev/null
cp $(find /tmp/prof_stats2 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_rocprofv3_kernel_stats_two_chains.csv
export BRV_CTN_STREAMS=1
PMCARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-through-trainer --no-fp32-path --no-other-configs --min-warmup-s 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python3 $REPO/bench.py $PMCARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python3 $REPO/bench.py $PMCARGS > /dev/null 2>&1
python3 $REPO/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w $OUT/${TAG}_pmc_hbm_traffic.json
# the same two PMC passes for the switchable variants of the first-conv backward (csrc/pw1_bwd.cuh):
# stored z1 / dz1 (round-3 kernels) and dz1 never stored (weight gradient rebuilds it too)
for V in stored:BRV_PW1_RC=0 rc_wgrad:BRV_PW1_RC_WGRAD=1; do
  NAME=${V%%:*}; export ${V#*:}
  rm -rf /tmp/pmc_fv /tmp/pmc_wv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fv -o f -- python3 $REPO/bench.py $PMCARGS > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_wv -o w -- python3 $REPO/bench.py $PMCARGS > /dev/null 2>&1
  python3 $REPO/tools/pmc_traffic.py /tmp/pmc_fv /tmp/pmc_wv $OUT/${TAG}_pmc_hbm_traffic_${NAME}.json
  VAR=${V#*:}; unset ${VAR%%=*}
done
# the complete default line without a profiler attached (what the driver runs)
unset BRV_CTN_STREAMS
cd $REPO && python3 bench.py > $OUT/${TAG}_bench_line.json 2> /dev/null
grep -v "^/opt" $OUT/${TAG}_kernel_table.txt | grep -E "^[a-z_0-9]+ +[0-9]" | head -12
cut -c1-400 $OUT/${TAG}_bench_line.json
