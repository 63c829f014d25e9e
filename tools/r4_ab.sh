#!/bin/bash
# round 4 A/B driver on the GPU box: tools/r4_ab.sh "<ab.py args>" [pytest args ...]
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
timeout 1200 python tools/ab.py $1 2>&1 | tee gpurun_out/ab.log | grep -v "^{" | tail -24
shift
if [ $# -gt 0 ]; then timeout 2400 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/ab_tests.log; fi
