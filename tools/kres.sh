#!/bin/bash
# registers / spills / LDS of the kernels whose mangled name matches $1 (device-only compile of csrc/$2, default convtasnet.hip)
PAT=$1; SRC=${2:-convtasnet.hip}; shift; shift
cd $(dirname $0)/../brever_amd/csrc
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $SRC -o /dev/null 2>&1 \
  | grep -A11 "Function Name: .*$PAT" | grep -E "Function Name|VGPRs|AGPRs|Scratch|LDS Size|Occupancy|SGPRs:" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
