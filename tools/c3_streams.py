#!/usr/bin/env python3
"""Config 3 (50 388 candidates, bundle, 20 obstacles) as ONE context against the same grid split by time samples over n engine
contexts on n streams (walk(B) overlapping obstacle(A)): wall time until every part's result has arrived.
usage: c3_streams.py [parts ...]   (default 2 3 4)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls


def inputs(nt):
    return synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(nt, 51, 51), n_obstacles=20, n_pred=30, lead_gap=25.0,
                                 hull_builder=build_obstacle_hulls)


def wall(fn, n=200):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return float(np.median(ts)) * 1e6


whole = inputs(19)
with FrenetEngine(max_candidates=whole.n_candidates + 64, max_steps=30) as e:
    e.upload(whole)
    print(f"one context: C={whole.n_candidates} wall {wall(lambda: (e.evaluate(), e.finish())):.1f} us  info {e.step_info()}", flush=True)
for parts in [int(a) for a in sys.argv[1:]] or [2, 3, 4]:
    nts = [19 // parts + (1 if i < 19 % parts else 0) for i in range(parts)]
    ins = [inputs(nt) for nt in nts]
    engs = [FrenetEngine(max_candidates=i.n_candidates + 64, max_steps=30) for i in ins]
    for e, i in zip(engs, ins):
        e.upload(i)
    alone = [wall(lambda e=e: (e.evaluate(), e.finish()), 50) for e in engs]
    def both():
        for e in engs: e.evaluate()
        for e in engs: e.finish()
    print(f"{parts} parts {nts}: alone {[round(a, 1) for a in alone]} us, together {wall(both):.1f} us", flush=True)
    for e in engs: e.close()
