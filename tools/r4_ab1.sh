#!/bin/bash
# round 4, first A/B: pre-hull wave cull + five-group prediction loop + shared LDS tail vs round 3's build; GPU parity tests of the touched paths
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
timeout 900 python tools/ab.py --builds cur,prev --rounds 2 m1o c5 c3 2>&1 | tee gpurun_out/ab1.log | grep -v "^{" | tail -20
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_obstacle_kernel.py tests/test_stress_config5.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/ab1_tests.log
