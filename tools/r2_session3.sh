#!/bin/bash
# package tests + tuning sweep + closed-loop host timing
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 900 python -m pytest tests/test_package.py tests/test_hip_planner.py tests/test_multiagent.py tests/test_distributed_gloo.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python tools/sweep_tuning.py c3B c3A 2>&1 | tail -110
timeout 300 python tools/closed_loop_timing.py 2>&1 | tail -20
