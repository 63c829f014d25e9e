#!/bin/bash
O=gpurun_out/r2b; mkdir -p $O
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python3 -c "
import json; d=json.load(open('$O/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','resident_step_p50_ms','plan_step_p50_ms','with_upload_ms_per_step','eval_kernel_ms','device_ms_per_step','winner')})
print(d['roofline']); print(d['roofline_hbm']['frac'])
for k,v in d['north_star'].items(): print(k, v['eval_kernel_ms'], v['step_ms'], v['roofline']['frac'], v['winner'])
"
python3 bench.py --no-cpu-baseline --no-north-star --workload config2 > $O/bench_c2.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/bench_c2.json'))
print({k:d[k] for k in ('value','ms_per_step','resident_step_p50_ms','plan_step_p50_ms','with_upload_ms_per_step','eval_kernel_ms')}, d['roofline']['frac'])"
timeout 900 python3 -m pytest tests/test_bench_contract.py -m gpu -x -q 2>&1 | tail -5
