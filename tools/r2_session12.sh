#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 900 python -m pytest tests/test_distributed_gloo.py tests/test_package.py tests/test_multiagent.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/exchange_overhead.py 2>&1 | tail -5
timeout 300 python tools/closed_loop_timing.py 2>&1 | grep level
