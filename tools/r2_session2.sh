#!/bin/bash
O=gpurun_out/r2s2; mkdir -p $O
timeout 1200 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1
tail -15 $O/gpu_tests.txt
timeout 600 python3 tools/obst_sweep.py > $O/obst_sweep.txt 2>&1
cat $O/obst_sweep.txt | tail -20
