#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 900 python -m pytest tests/test_package.py tests/test_hip_planner.py tests/test_multiagent.py tests/test_distributed_gloo.py tests/test_adapter_replay.py -x -q -m gpu 2>&1 | tail -8
echo "--- new build"; timeout 300 python tools/c3.py c3B c3A c2B c2A m1o 2>&1 | tail -6
echo "--- previous build"; FXPLAN_SO=$PWD/tools/probe_build/libfxplan_prev.so timeout 300 python tools/c3.py c3B c3A c2B c2A m1o 2>&1 | tail -6
echo "--- new build again"; timeout 300 python tools/c3.py c3B c3A c2B c2A 2>&1 | tail -5
echo "--- probe"; timeout 300 python tools/probe_phases.py c3B c3A c2B c3B_g4 2>&1 | tail -20
echo "--- closed loop"; timeout 300 python tools/closed_loop_timing.py 2>&1 | grep level
