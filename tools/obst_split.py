#!/usr/bin/env python3
"""Where does the obstacle stage spend its time?  1 M candidates x 31 samples x 20 obstacles, select-only:
full stage / prediction cost only (collision off) / no obstacles."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from frenetix_motion_planner_amd.problem import DEFAULT_COST_WEIGHTS


def run(label, steps=8, tuning=(0, 0, 0, 0), **kw):
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, write_bundle=False,
                                write_costmap=False, **kw)
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        eng.set_timing("kernel"); eng.set_tuning(*tuning); eng.upload(inp)
        for _ in range(3): eng.evaluate(); eng.finish()
        ts = []
        for _ in range(steps):
            eng.evaluate(); r = eng.finish()[0]; ts.append(eng.last_eval_kernel_ms)
    print(label, inp.n_candidates, round(float(np.median(ts)) * 1e3, 1), "us", r["n_collisions"], flush=True)


G1M = (19, 230, 229)
for tn in ((0, 0, 0, 0), (1, 2, 2, 256), (1, 3, 2, 256)):
    run(f"full {tn}", tuning=tn, grid=G1M, n_obstacles=20, lead_gap=25.0)
    run(f"pred only {tn}", tuning=tn, grid=G1M, n_obstacles=20, collision=False)
    w = {k: v for k, v in DEFAULT_COST_WEIGHTS.items() if k != "prediction"}
    run(f"collision only {tn}", tuning=tn, grid=G1M, n_obstacles=20, cost_weights=w, lead_gap=25.0)
    run(f"no obstacles {tn}", tuning=tn, grid=G1M)
