#!/usr/bin/env python3
"""Obstacle stage fused into the walk vs as its own kernel (fx_set_obstacle_stage): kernel times (walk, obstacle kernel), synchronous
step wall time, and agreement of the results.  usage: c3_split.py [workload ...] ; FX_SPLIT_STEPS="2,3,5" (steps per work item)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

W = dict(
    c3B=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
    c3Bnc=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, collision=False),
    m1oB=dict(grid=(19, 230, 229), n_obstacles=20, lead_gap=25.0),
    c5B=dict(grid=(39, 51, 51), horizon=5.0, n_pred=50, n_obstacles=20),
    c4=dict(grid=(19, 23, 23), n_obstacles=5),
    l3=dict(grid=(10, 17, 17), n_obstacles=5),
)
steps = [int(v) for v in os.environ.get("FX_SPLIT_STEPS", "2,3,5").split(",")]
which = sys.argv[1:] or ["c3B"]
out = {}
for name in which:
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, **W[name])
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        eng.set_timing("kernel")
        base = None
        for stage, CH in [(1, 0)] + [(2, ch) for ch in steps]:
            try:
                eng.set_obstacle_stage(stage, CH)
                eng.upload(inp)
            except ValueError as e:
                print(name, stage, CH, "not applicable:", e, flush=True)
                continue
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                eng.evaluate(); eng.finish()
            n = 60 if inp.n_candidates < 200000 else 15
            for _ in range(n):
                eng.evaluate(); r = eng.finish()[0]
            ev, st = eng.kernel_times(n)
            ob = eng.obstacle_kernel_times(n)
            eng.set_timing("off")
            tw = []
            for _ in range(n):
                a = time.perf_counter(); eng.evaluate(); eng.finish(); tw.append(time.perf_counter() - a)
            eng.set_timing("kernel")
            cost, flags = eng.costs()
            key = f"{name}:{'fused' if stage == 1 else f'split{CH}'}"
            out[key] = dict(walk_us=round(float(np.median(ev)) * 1e3, 1), obst_us=round(float(np.median(ob)) * 1e3, 1),
                            device_step_us=round(float(np.median(st)) * 1e3, 1), wall_p50_us=round(float(np.median(tw)) * 1e6, 1),
                            winner=r["best_index"], coll=r["n_collisions"], info=eng.step_info()["obstacle_kernel"])
            if base is None:
                base = (cost, flags, r)
            else:
                c = (flags & 16) != 0
                rel = np.abs(cost[c] - base[0][c]) / np.maximum(np.abs(base[0][c]), 1e-12)
                out[key].update(flags_equal=bool(np.array_equal(flags, base[1])), cost_rel=float(rel.max()) if c.any() else 0.0,
                                same_result=bool(r["best_index"] == base[2]["best_index"] and r["n_collisions"] == base[2]["n_collisions"]))
            print(key, out[key], flush=True)
print(json.dumps(out))
