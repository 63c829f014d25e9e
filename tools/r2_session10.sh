#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/closed_loop_timing.py 2>&1 | grep -E "level|tottime|engine.py|problem.py|reactive_planner.py|trajectories.py|coordinate_system.py" | head -30
