#!/bin/bash
# The round's evidence on ONE box: usage tools/final_evidence.sh <round, e.g. r4> [--no-profiles]
# rocprofv3 kernel-trace stats + per-kernel PMC passes (tools/collect_profiles.sh), then the bench lines of every workload and the
# small measurement tools; everything lands in gpurun_out/<round>final -- copy what is judged into profiles/<round>/.
cd $GRAFT_REPO_ROOT 2>/dev/null || true
R=${1:-r5}
O=gpurun_out/${R}final; mkdir -p $O profiles/$R
if [ "$2" != "--no-profiles" ]; then
  bash tools/collect_profiles.sh ${R}final_prof > $O/collect.log 2>&1
  tail -3 $O/collect.log
  cp gpurun_out/${R}final_prof/summary.json profiles/$R/summary.json      # bench.py reads the PMC figures of THIS build
  cp gpurun_out/${R}final_prof/kernel_stats.csv profiles/$R/kernel_stats.csv
  cp gpurun_out/${R}final_prof/summary.json gpurun_out/${R}final_prof/kernel_stats.csv $O/
fi
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
FX_SPLIT_STEPS=2,3,5 timeout 600 python tools/c3_split.py c3B c5B c4 m1oB 2>&1 | grep -v amdgpu | grep -v '^{' > $O/obstacle_stage_ab.txt
timeout 300 python tools/cull_stats.py m1o c5 c3A sparse 2>&1 | grep -v amdgpu > $O/cull_stats.txt || true
timeout 300 python tools/generic_kernel_cases.py 2>&1 | grep -v amdgpu > $O/generic_kernel_cases.txt || true
if [ -f tools/probe_build/libfxplan_p2.so ]; then timeout 300 python tools/probe_timeline.py c1 c1_noobs zam630 l4 c4agent 2>&1 | grep -v amdgpu > $O/probe_timeline.txt || true; fi
ls -la $O
