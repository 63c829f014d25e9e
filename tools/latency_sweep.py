#!/usr/bin/env python3
"""Plan-step latency over grid sizes (auto-tuning vs forced lane counts), inputs resident, no obstacles."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine

def lat(eng, n=200):
    for _ in range(20): eng.evaluate(); eng.finish()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); eng.evaluate(); eng.finish(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e6, 1)

for grid in ((3, 5, 4), (5, 9, 8), (7, 13, 12), (9, 17, 16), (13, 25, 24), (17, 33, 32), (19, 51, 51)):
    for label, kw in (("B", {}), ("A", dict(write_bundle=False, write_costmap=False))):
        inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=grid, **kw)
        out = {}
        with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
            for G, blk, mp in ((0, 0, 0), (1, 256, 0), (2, 256, 2), (4, 256, 2), (8, 256, 1), (4, 128, 1), (8, 64, 1), (8, 128, 1)):
                try:
                    eng.set_tuning(G, 2 if G else 0, 2 if G else 0, blk, mp); eng.upload(inp)
                    out[f"G{G}b{blk}m{mp}"] = lat(eng)
                except ValueError as e:
                    out[f"G{G}b{blk}m{mp}"] = None
        print(label, inp.n_candidates, json.dumps(out), flush=True)
