#!/usr/bin/env python3
"""One workload, one obstacle-stage shape, N resident steps (driver for rocprofv3 passes).
usage: run_split.py <workload c3B|c3Bnc|...> <stage 1|2> <steps per item> [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
W = dict(
    c3B=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
    c3Bnc=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, collision=False),
    c2B=dict(grid=(19, 51, 51)),
)
name, stage, CH = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, **W[name])
with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
    eng.set_obstacle_stage(stage, CH)
    eng.upload(inp)
    for _ in range(n):
        eng.evaluate(); r = eng.finish()[0]
    print(name, stage, CH, r["best_index"], r["n_collisions"], eng.step_info())
