#!/bin/bash
# product library + the probe build (-DFX_PROBE=3 -> tools/probe_build/libfxplan_p3.so), side by side
cd "$(dirname "$0")/../frenetix-motion-planner_amd/csrc" || exit 1
mkdir -p ../../tools/probe_build
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -DFX_PROBE=${FX_PROBE:-3} -shared -o ../../tools/probe_build/libfxplan_p3.so fx_kernels.hip fx_api.hip fx_api_step.hip fx_api_exchange.hip fx_api_host.hip 2>&1 | grep -E "error" ; echo probe built) &
make libfxplan.so 2>&1 | grep -E "error|Error"; echo lib built
wait
