#!/usr/bin/env python3
"""The north star as written with the obstacle stage as its own kernel behind the walk (forced), beside the fused kernel: what a
chunked walk || obstacle-kernel pipeline could reach.   usage: ns_split.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls


def leg(label, stage, ch=0, grid=(19, 230, 229), wpe=0):
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=grid, n_obstacles=20, n_pred=30, lead_gap=25.0, hull_builder=build_obstacle_hulls)
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=30) as e:
        e.set_timing("kernel")
        e.set_tuning(0, wpe, 0, 0, 0)
        e.set_obstacle_stage(stage, ch)
        e.upload(inp)
        for _ in range(3):
            e.evaluate(); res = e.finish()[0]
        ts, ws, os_ = [], [], []
        for _ in range(10):
            t0 = time.perf_counter(); e.evaluate(); e.finish(); ws.append(time.perf_counter() - t0)
            ts.append(e.last_eval_kernel_ms)
            os_.append(getattr(e, "last_obstacle_kernel_ms", float("nan")))
        info = e.step_info()
        print(f"{label:28s} {inp.n_candidates:8d} cand: walk {np.median(ts) * 1e3:7.1f} us, obstacle kernel {np.median(os_) * 1e3:7.1f} us, step wall "
              f"{np.median(ws) * 1e6:7.1f} us; G {info.get('lanes_per_candidate')} wpe {info.get('waves_per_simd')} split {info.get('obstacle_kernel')}; "
              f"winner {res['best_index']} collisions {res['n_collisions']}", flush=True)


for grid in ((19, 230, 229), (19, 115, 115), (19, 81, 81)):
    leg("fused (auto)", 0, grid=grid)
    for ch in (0, 5, 8):
        leg(f"split forced, CH {ch or 'auto'}", 2, ch, grid=grid)
