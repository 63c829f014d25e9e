#!/bin/bash
# AddressSanitizer + UBSan run of the product's HOST C (CPU box only): _fxhost (csrc/fx_host_ext.c) and the host half of libfxplan.
# usage (repo root): bash tools/asan_host.sh [> profiles/r6/asan_host.log]
set -e
make -C frenetix-motion-planner_amd/csrc -s asan
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python3 tools/asan_host.py
