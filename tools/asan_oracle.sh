#!/bin/bash
# AddressSanitizer + UBSan run of the CPU oracle (sanitizers are CPU-only on this pool): 120 seeded random scenarios through
# fxo_plan_step and the threaded fxo_plan_range_mt.   usage (repo root): bash tools/asan_oracle.sh
make -C oracle -s asan
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 python3 tools/asan_oracle.py
