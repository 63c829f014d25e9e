#!/usr/bin/env python3
"""Kernel time of one workload over the work decompositions (lanes per candidate G, waves per SIMD, workgroup size, horizon
split mapping).  usage: sweep_tuning.py [c3B|c3A|c2B|c2A|...]"""
import itertools, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

W = dict(
    c3B=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
    c3A=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
    c2B=dict(grid=(19, 51, 51)),
    c2A=dict(grid=(19, 51, 51), write_bundle=False, write_costmap=False),
    m1o=dict(grid=(19, 230, 229), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
)
for name in sys.argv[1:] or ["c3B"]:
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, **W[name])
    rows = []
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        eng.set_timing("kernel"); eng.upload(inp)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            eng.evaluate(); eng.finish()
        base = None
        combos = [(0, 0, 0, 0)] + list(itertools.product((1, 2, 4, 8), (2, 3), (0, 128, 64), (1, 2)))
        if inp.n_candidates > 300000:   # large grids: one lane per candidate, the occupancy target and the workgroup size
            combos = [(0, 0, 0, 0)] + [(1, w, b, 0) for w in (2, 3, 4) for b in (0, 128)] + [(2, w, 0, 2) for w in (2, 3)]
        for G, wpe, blk, mp in combos:
            try:
                eng.set_tuning(G, wpe, 2 if G else 0, blk, mp)
                eng.upload(inp)
                ts = []
                for _ in range(30 if inp.n_candidates < 300000 else 8):
                    eng.evaluate(); r = eng.finish()[0]; ts.append(eng.last_eval_kernel_ms)
                t = round(float(np.median(ts)) * 1e3, 1)
            except Exception as e:
                t = None
            if base is None:
                base = r["best_index"]
            assert t is None or r["best_index"] == base
            rows.append((t, G, wpe, blk, mp))
            print(name, "G", G, "wpe", wpe, "block", blk, "map", mp, t, flush=True)
    rows = sorted((r for r in rows if r[0] is not None))
    print(name, "best:", rows[:6], "auto:", [r for r in rows if r[1] == 0])
