#!/bin/bash
# Hardware counters of one layer's row-convolution launches (tools/cconv_bench.py with LAYER=<name>), per library:
#   bash tools/cconv_counters.sh <tag> <layer> [lib.so]     (through gpurun, from the repo root)
set -u
TAG=$1; LAYER=$2; LIB=${3:-}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LAYER
[ -n "$LIB" ] && export BRV_LIB_PATH=$LIB
N=0
# (SQ counters only: a pass with TA_* / TCP_* counters never returned on this pool and ran into gpurun's limit; every
# pass is bounded by its own timeout)
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$((N + 1))
  rm -rf /tmp/cc_$N
  timeout 150 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/cc_$N -o p -- python3 $REPO/tools/cconv_bench.py bf16 > /tmp/cc_$N.log 2>&1
  f=$(find /tmp/cc_$N -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $OUT/${LAYER}_pass$N.csv; else echo "pass $N failed: $(tail -2 /tmp/cc_$N.log | cut -c1-200)"; fi
done
python3 - $OUT $LAYER <<'PY'
import csv, sys, glob, collections
out, layer = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob(f'{out}/{layer}_pass*.csv')):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if 'cconv' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[(k, r['Dispatch_Id'])] += 1
    for (k, _d) in seen: cnt[(k, f)] += 1
for k in acc:
    n = max(v for (kk, f), v in cnt.items() if kk == k)
    print(k, 'dispatches', n)
    for c, v in sorted(acc[k].items()): print(f'   {c:45s} {v/n:16.0f}')
PY
