#!/bin/bash
# quick loop on the GPU box: the parity tests, then kernel times
O=gpurun_out/r2q; mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -6 $O/gpu_tests.txt
timeout 300 python3 tools/c3.py "$@" 2>&1 | tail -1
timeout 300 python3 bench.py --workload config3 --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('ms_per_step','plan_step_p50_ms','device_ms_per_step','eval_kernel_ms','winner')})"
