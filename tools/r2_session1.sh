#!/bin/bash
# GPU session 1 of round 2: baseline sweeps + phase stamps before the kernel work
O=gpurun_out/r2s1; mkdir -p $O
python3 tools/quick.py > $O/quick.txt 2>&1
python3 tools/probe_phases.py > $O/phases.txt 2>&1
python3 tools/e2e_step.py > $O/e2e.json 2>&1
python3 bench.py --workload config3 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_c3.json 2>&1
tail -5 $O/quick.txt $O/phases.txt
