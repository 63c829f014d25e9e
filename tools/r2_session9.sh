#!/bin/bash
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 900 python tools/sweep_tuning.py c2A c2B c3B 2>&1 | grep -E "best|auto"
timeout 300 python tools/probe_phases.py c2A c2B c3B 2>&1 | tail -12
timeout 300 python tools/closed_loop_timing.py 2>&1 | grep -E "level|plan_step_packaged|pack_predictions|_inputs_for_level|plan_finish|make_state|structure_key"
