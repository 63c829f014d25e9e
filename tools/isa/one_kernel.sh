#!/bin/bash
# Compile ONE specialisation of the grid kernel (default: the north-star / config-5 one) to gfx950 assembly and print its
# resource usage + loop summary.  usage: one_kernel.sh [G BUNDLE OBST WPE WSPLIT] [extra hipcc flags ...]
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
G=${1:-1}; B=${2:-false}; O=${3:-true}; W=${4:-3}; WS=${5:-false}; shift 5 2>/dev/null || true
out=/tmp/fx_one; mkdir -p $out
cat > $out/one.hip <<EOT
#include <hip/hip_ext.h>
#include "fx_eval_kernel.h"
#include "fx_eval_grid_kernel.h"
template __global__ void fx_eval_grid_kernel<$G, $B, $O, $W, $WS>(const DevProblem *__restrict__, const FuseArgs);
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function \
  -I$root/frenetix-motion-planner_amd/csrc -I$root/include --cuda-device-only -S -o $out/one.s $out/one.hip \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep -E "VGPRs:|Spill|ScratchSize|Occupancy|SGPRs:" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste -sd' '
python3 $here/kernel_isa.py $out/one.s fx_eval_grid_kernel --dump $out/kernel.s --min 30
