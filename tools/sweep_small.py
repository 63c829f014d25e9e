#!/usr/bin/env python3
"""Planner-sized grids (630 ... 11 220 candidates): kernel and step time over lanes per candidate (4 ... 32) and workgroup size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

W = dict(l2=dict(level=2, n_obstacles=5), l2_no=dict(level=2), l3=dict(level=3, n_obstacles=5), g4k=dict(grid=(7, 24, 24), n_obstacles=5),
         l4=dict(level=4, n_obstacles=5), l2_h5=dict(level=2, n_obstacles=5, horizon=5.0, n_pred=50),
         l2_k20=dict(level=2, n_obstacles=20), l3_k20=dict(level=3, n_obstacles=20),
         l2m=dict(level=2, n_obstacles=5, as_matrix=True), l3m=dict(level=3, n_obstacles=5, as_matrix=True))   # ..m: C x 13 matrix -> generic kernel
for name in sys.argv[1:] or list(W):
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, **W[name])
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        eng.set_timing("kernel"); eng.upload(inp)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            eng.evaluate(); eng.finish()
        rows = []
        for G, blk, mp in [(0, 0, 0)] + [(G, b, 1) for G in (4, 8, 16, 32) for b in (64, 128, 256)] + [(2, 128, 2), (4, 256, 2)]:
            try:
                eng.set_tuning(G, 2, 2 if G and not W[name].get('as_matrix') else 0, blk, mp)
                eng.upload(inp)
                ts, tw = [], []
                for _ in range(40):
                    a = time.perf_counter(); eng.evaluate(); r = eng.finish()[0]; tw.append(time.perf_counter() - a); ts.append(eng.last_eval_kernel_ms)
                info = eng.step_info()
                rows.append((round(float(np.median(ts)) * 1e3, 1), round(float(np.median(tw)) * 1e6, 1), G, blk, info["lanes_per_candidate"], info["block"], "wave split" if info["wave_split"] else "lane split"))
            except Exception as e:
                rows.append((None, None, G, blk, repr(e)[:60]))
        print(name, inp.n_candidates, "candidates (kernel us, step us, G, block, G used, block used):")
        for r in rows:
            print("   ", r, flush=True)
