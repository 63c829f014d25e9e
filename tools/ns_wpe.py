#!/usr/bin/env python3
"""The north star as written (1 005 100 candidates x 31 x 20 obstacles, bundle + collision stage) at two and three waves per SIMD;
select-only and bundle-only legs beside it.   usage: ns_wpe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls


def leg(label, wpe, **kw):
    base = dict(ref_kind="arc", v0=10.0, grid=(19, 230, 229), n_obstacles=20, n_pred=30, lead_gap=25.0, hull_builder=build_obstacle_hulls)
    base.update(kw)
    inp = synthetic.make_inputs(**base)
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=30) as e:
        e.set_timing("kernel")
        e.set_tuning(0, wpe, 0, 0, 0)
        e.upload(inp)
        for _ in range(3):
            e.evaluate(); res = e.finish()[0]
        ts, ws = [], []
        for _ in range(12):
            t0 = time.perf_counter(); e.evaluate(); e.finish(); ws.append(time.perf_counter() - t0)
            ts.append(e.last_eval_kernel_ms)
        info = e.step_info()
        print(f"{label:34s} waves/SIMD asked {wpe or 'auto'} -> {info['waves_per_simd']}: evaluation kernel {np.median(ts) * 1e3:7.1f} us (min {min(ts) * 1e3:.1f}), "
              f"step wall {np.median(ws) * 1e6:7.1f} us; winner {res['best_index']} collisions {res['n_collisions']}", flush=True)


for wpe in (0, 2, 3):
    leg("bundle + obstacles (north star)", wpe)
for wpe in (0, 3, 4):
    leg("select only, obstacles", wpe, write_bundle=False, write_costmap=False)
for wpe in (0, 3):
    leg("bundle, no obstacles", wpe, n_obstacles=0)
