#!/usr/bin/env python3
"""Does a vector-unit-bound kernel hide under the store-bound walk when the two run on two streams?  Context 1: config 2 (the walk with the
bundle, no obstacles); context 2: the same grid select-only with the obstacle stage fused (no stores).  Wall time alone and together."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls


def wall(fn, n=300):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return float(np.median(ts)) * 1e6


kw = dict(ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_pred=30, lead_gap=25.0, hull_builder=build_obstacle_hulls)
a = synthetic.make_inputs(n_obstacles=0, **kw)
for label, bkw, stage in (("select-only + 20 obstacles fused (VALU-bound)", dict(n_obstacles=20, write_bundle=False, write_costmap=False), 1),
                          ("select-only, no obstacles", dict(n_obstacles=0, write_bundle=False, write_costmap=False), 0),
                          ("config 3 whole step", dict(n_obstacles=20), 0)):
    b = synthetic.make_inputs(**dict(kw, **bkw))
    with FrenetEngine(max_candidates=a.n_candidates + 64, max_steps=30) as e1, FrenetEngine(max_candidates=b.n_candidates + 64, max_steps=30) as e2:
        if stage:
            e2.set_obstacle_stage(stage)
        e1.upload(a); e2.upload(b)
        w1 = wall(lambda: (e1.evaluate(), e1.finish()))
        w2 = wall(lambda: (e2.evaluate(), e2.finish()))
        both = wall(lambda: (e1.evaluate(), e2.evaluate(), e1.finish(), e2.finish()))
        print(f"walk with bundle alone {w1:.1f} us | {label} alone {w2:.1f} us | together {both:.1f} us (sum {w1 + w2:.1f})", flush=True)
