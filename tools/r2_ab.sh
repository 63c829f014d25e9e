#!/bin/bash
# same-box A/B of the current build against tools/probe_build/libfxplan_prev.so
cd $GRAFT_REPO_ROOT 2>/dev/null || true
W="${@:-c3B c3A c2B c2A}"
echo "--- new build"; timeout 300 python tools/c3.py $W 2>&1 | grep -v "^ " | tail -7
echo "--- previous build"; FXPLAN_SO=$PWD/tools/probe_build/libfxplan_prev.so timeout 300 python tools/c3.py $W 2>&1 | grep -v "^ " | tail -7
echo "--- new build again"; timeout 300 python tools/c3.py $W 2>&1 | grep -v "^ " | tail -7
