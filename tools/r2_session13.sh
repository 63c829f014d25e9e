#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
echo "--- new build"; timeout 300 python tools/c3.py c3B c3A c2B c2A 2>&1 | tail -5
echo "--- previous build"; FXPLAN_SO=$PWD/tools/probe_build/libfxplan_prev.so timeout 300 python tools/c3.py c3B c3A c2B c2A 2>&1 | tail -5
echo "--- new build again"; timeout 300 python tools/c3.py c3B c3A c2B c2A 2>&1 | tail -5
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_math.py -x -q -m gpu 2>&1 | tail -3
