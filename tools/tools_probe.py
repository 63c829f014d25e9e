#!/usr/bin/env python3
"""Probe: fixed vs per-step cost of the evaluation kernel (GPU box)."""
import json, sys
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

VARIANT = int(os.environ.get('FXV', '2'))
def t(inp, G, w, steps=40):
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_ref_knots=2048) as eng:
        eng.set_timing("kernel")
        eng.set_tuning(G, w, VARIANT); eng.upload(inp)
        for _ in range(5): eng.evaluate(); eng.finish()
        ts = []
        for _ in range(steps):
            eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
        return round(float(np.median(ts)) * 1e3, 1)

for label, kw in (("modeA", dict(write_bundle=False, write_costmap=False)), ("modeB", dict())):
    for hz in (1.5, 3.0, 6.0, 12.0):
        inp = synthetic.make_inputs(ref_kind="arc", n_knots=1200, v0=10.0, grid=(19, 51, 51), horizon=hz, **kw)
        print(label, "S", inp.n_samples, {f"G{G}": t(inp, G, 2) for G in (1, 2, 4)})
for C in ((2, 5, 5), (5, 11, 11), (10, 25, 25), (19, 51, 51), (19, 102, 102), (19, 230, 229)):
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=C, write_bundle=False, write_costmap=False)
    print("modeA C", inp.n_candidates, {f"G{G}": t(inp, G, 2, 20) for G in (1, 2, 4)})
