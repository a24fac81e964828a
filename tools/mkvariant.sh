#!/bin/bash
# Build a variant of libbrever_hip.so HERE (hipcc cross-compiles gfx950) into tools/_v/<tag>/ -- the .so travels to the
# GPU box with the snapshot (git-ignored, not gpurun-ignored), so no box time goes into compiling:
#   tools/mkvariant.sh <tag> [file.hip] [-DFLAG ...]      (file defaults to convtasnet.hip)
# then on the box:  python tools/ab_step.py base= v=BRV_LIB_PATH=tools/_v/<tag>/libbrever_hip.so --labels ...
set -e
TAG=$1; shift
SRC=convtasnet.hip
if [[ "${1:-}" == *.hip ]]; then SRC=$1; shift; fi
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/brever_amd/csrc
OUT=$ROOT/tools/_v/$TAG
mkdir -p $OUT
OBJ=$OUT/${SRC%.hip}.o
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -c $CSRC/$SRC -o $OBJ
OTHERS=$(ls $CSRC/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -shared --offload-arch=gfx950 -o $OUT/libbrever_hip.so $OBJ $OTHERS
rm -f $OBJ
ls -la $OUT/libbrever_hip.so | awk '{print $5, $9}'
