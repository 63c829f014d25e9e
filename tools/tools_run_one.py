#!/usr/bin/env python3
"""Run one configuration a few times (for rocprofv3 --pmc / --kernel-trace on the GPU box).
usage: tools_run_one.py G WPE VARIANT modeA|modeB n_obst [nt nv nd] [steps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
G, w, var, mode, nobs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
grid = tuple(int(x) for x in sys.argv[6:9]) if len(sys.argv) > 8 else (19, 51, 51)
steps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
b = mode == "modeB"
inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=grid, n_obstacles=nobs, write_bundle=b, write_costmap=b,
                            hull_builder=build_obstacle_hulls)
with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
    eng.set_timing("kernel")
    eng.set_tuning(G, w, var)
    eng.upload(inp)
    for _ in range(steps):
        eng.evaluate(); r = eng.finish()[0]
    print(inp.n_candidates, r["best_index"], eng.last_eval_kernel_ms)
