REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/f32prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/f32prof -o f32 -- python3 $REPO/tools/bench_rows.py --rows convtasnet_fp32 > /tmp/f32prof.log 2>&1
tail -2 /tmp/f32prof.log
cp $(find /tmp/f32prof -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/r3_f32b_kernel_stats.csv
