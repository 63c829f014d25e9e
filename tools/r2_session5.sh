#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_soak_parity.py tests/test_package.py -x -q -m gpu 2>&1 | tail -8
echo "--- new build"; timeout 300 python tools/c3.py c3B c3A m1o c5 2>&1 | tail -5
echo "--- previous build"; FXPLAN_SO=$PWD/tools/probe_build/libfxplan_prev.so timeout 300 python tools/c3.py c3B c3A m1o c5 2>&1 | tail -5
echo "--- new build again"; timeout 300 python tools/c3.py c3B c3A 2>&1 | tail -3
echo "--- upload step (kernel staging)"; timeout 300 python tools/upload_step.py 2>&1 | tail -16
echo "--- upload step (dma)"; FX_STAGE=dma timeout 300 python tools/upload_step.py 2>&1 | tail -16
