#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
bash tools/collect_profiles_r2.sh prof_r2b > gpurun_out/prof_r2b.log 2>&1
tail -5 gpurun_out/prof_r2b.log
mkdir -p gpurun_out/bench_r2b
cp gpurun_out/prof_r2b/summary.json profiles/r2/summary.json   # so the bench lines below read this box's counters
timeout 600 python bench.py > gpurun_out/bench_r2b/bench_config3.json 2> gpurun_out/bench_r2b/bench_config3.err; tail -c 3000 gpurun_out/bench_r2b/bench_config3.json
timeout 600 python bench.py --workload config2 --no-north-star > gpurun_out/bench_r2b/bench_config2.json 2>/dev/null
timeout 600 python bench.py --workload config2 --select-only --no-north-star --no-cpu-baseline > gpurun_out/bench_r2b/bench_config2_modeA.json 2>/dev/null
timeout 600 python bench.py --workload config1 --no-cpu-baseline > gpurun_out/bench_r2b/bench_config1.json 2>/dev/null
timeout 600 python bench.py --workload config4 --no-cpu-baseline --steps 30 --warmup 6 > gpurun_out/bench_r2b/bench_config4.json 2>/dev/null
timeout 600 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/bench_r2b/bench_config5.json 2>/dev/null
timeout 300 python tools/upload_step.py > gpurun_out/bench_r2b/upload_step.json 2>/dev/null
ls -la gpurun_out/bench_r2b
