#!/bin/bash
# the bench lines and measurement tools of tools/r3_final.sh without the rocprofv3 collection (profiles/r3/summary.json stays)
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out/r3final; mkdir -p $O
timeout 900 python bench.py > $O/bench_config3.json 2>$O/bench_config3.err
timeout 600 python bench.py --workload config2 --no-north-star > $O/bench_config2.json 2>/dev/null
timeout 600 python bench.py --workload config2 --select-only --no-north-star --no-cpu-baseline > $O/bench_config2_modeA.json 2>/dev/null
timeout 600 python bench.py --workload config1 --no-cpu-baseline > $O/bench_config1.json 2>/dev/null
timeout 600 python bench.py --workload config4 --no-cpu-baseline --steps 30 --warmup 6 > $O/bench_config4.json 2>/dev/null
timeout 600 python bench.py --workload config5 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_config5.json 2>/dev/null
timeout 300 python tools/upload_step.py > $O/upload_step.json 2>/dev/null
timeout 300 python tools/exchange_overhead.py 2>&1 | grep "p50" > $O/exchange_overhead.txt
timeout 300 python tools/closed_loop_timing.py 2>&1 | grep level > $O/closed_loop.txt
timeout 300 python tools/closed_loop_segments.py 2>&1 | grep -v amdgpu > $O/closed_loop_segments.txt
timeout 300 python tools/seg_config4.py 2>&1 | grep -v amdgpu > $O/config4_segments.txt
timeout 300 python tools/adapter_matrix_timing.py 2>&1 | grep -v amdgpu > $O/adapter_matrix.txt
ls $O | wc -l
