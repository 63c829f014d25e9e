#!/bin/bash
# final evidence of the round: profiles (kernel stats + PMC passes) and the bench lines of every workload on one box
cd $GRAFT_REPO_ROOT 2>/dev/null || true
bash tools/collect_profiles_r2.sh prof_r2c > gpurun_out/prof_r2c.log 2>&1
tail -3 gpurun_out/prof_r2c.log
mkdir -p gpurun_out/bench_r2c
cp gpurun_out/prof_r2c/summary.json profiles/r2/summary.json
timeout 600 python bench.py > gpurun_out/bench_r2c/bench_config3.json 2>/dev/null
timeout 600 python bench.py --workload config2 --no-north-star > gpurun_out/bench_r2c/bench_config2.json 2>/dev/null
timeout 600 python bench.py --workload config2 --select-only --no-north-star --no-cpu-baseline > gpurun_out/bench_r2c/bench_config2_modeA.json 2>/dev/null
timeout 600 python bench.py --workload config1 --no-cpu-baseline > gpurun_out/bench_r2c/bench_config1.json 2>/dev/null
timeout 600 python bench.py --workload config4 --no-cpu-baseline --steps 30 --warmup 6 > gpurun_out/bench_r2c/bench_config4.json 2>/dev/null
timeout 600 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/bench_r2c/bench_config5.json 2>/dev/null
timeout 300 python tools/upload_step.py > gpurun_out/bench_r2c/upload_step.json 2>/dev/null
timeout 300 python tools/exchange_overhead.py 2>&1 | grep "p50" > gpurun_out/bench_r2c/exchange_overhead.txt
timeout 300 python tools/closed_loop_timing.py 2>&1 | grep level > gpurun_out/bench_r2c/closed_loop.txt
ls gpurun_out/bench_r2c
