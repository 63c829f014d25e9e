#!/usr/bin/env python3
"""Kernel time of the obstacle workloads with automatic tuning: config 3 (Mode B / Mode A), 1 M x 31 x 20 obstacles (Mode A),
config 2 (Mode B / Mode A), 1 M Mode A / B.  usage: c3.py [which ...]   (FXPLAN_SO selects the library build)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

W = dict(
    c3B=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
    c3A=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
    m1o=dict(grid=(19, 230, 229), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
    c2B=dict(grid=(19, 51, 51)),
    c2A=dict(grid=(19, 51, 51), write_bundle=False, write_costmap=False),
    m1A=dict(grid=(19, 230, 229), write_bundle=False, write_costmap=False),
    m1B=dict(grid=(19, 230, 229)),
    c5=dict(grid=(39, 51, 51), horizon=5.0, n_pred=50, n_obstacles=20, write_bundle=False, write_costmap=False),
)
which = sys.argv[1:] or ["c3B", "c3A", "m1o", "c2B", "c2A", "m1A"]
out = {}
for name in which:
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, **W[name])
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        eng.set_timing("kernel"); eng.upload(inp)
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:   # warm clocks: short runs after idle are timed at a lower clock
            eng.evaluate(); eng.finish()
        ts = []
        for _ in range(60 if inp.n_candidates < 200000 else 20):
            eng.evaluate(); r = eng.finish()[0]; ts.append(eng.last_eval_kernel_ms)
    out[name] = round(float(np.median(ts)) * 1e3, 1)
    print(name, out[name], "us  winner", r["best_index"], "coll", r["n_collisions"], flush=True)
print(json.dumps(out))
