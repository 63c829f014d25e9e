#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 120 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "grid_synchronised" 2>&1 | tail -6
echo "--- grid sync on"; timeout 300 python tools/c3.py c3B c3A c5 2>&1 | tail -4
echo "--- grid sync off"; FX_GRID_SYNC=0 timeout 300 python tools/c3.py c3B c3A c5 2>&1 | tail -4
echo "--- grid sync on"; timeout 300 python tools/c3.py c3B c3A 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_soak_parity.py tests/test_multiagent.py tests/test_package.py tests/test_distributed_gloo.py -x -q -m gpu 2>&1 | tail -4
