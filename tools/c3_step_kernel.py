#!/usr/bin/env python3
"""Config 3 (and friends) as three launches against the whole step in ONE launch (csrc/fx_step_kernel.h): results compared, wall
time and device time of both.   usage: c3_step_kernel.py [c3 c3pkg c4 small ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

CASES = dict(
    c3=dict(grid=(19, 51, 51), n_obstacles=20, n_pred=30, lead_gap=25.0),
    c3dbg=dict(grid=(19, 51, 51), n_obstacles=20, n_pred=30, lead_gap=25.0, draw_traj_set=True, kinematic_debug=True),
    c4agent=dict(grid=(19, 23, 23), n_obstacles=5),
    mid=dict(grid=(19, 31, 41), n_obstacles=12, ref_kind="scurve", kappa=0.02),
    h5=dict(grid=(39, 21, 31), n_obstacles=20, horizon=5.0, n_pred=50, n_knots=700),
    quarter=dict(grid=(19, 25, 26), n_obstacles=20, n_pred=30, lead_gap=25.0),
    half=dict(grid=(19, 36, 36), n_obstacles=20, n_pred=30, lead_gap=25.0),
    few_obst=dict(grid=(19, 51, 51), n_obstacles=5, n_pred=30, lead_gap=25.0),
)


def wall(fn, n=300):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return float(np.median(ts)) * 1e6


def run(name, mode, ch=0, package=False):
    kw = dict(ref_kind="arc", v0=10.0, hull_builder=build_obstacle_hulls)
    kw.update(CASES[name])
    inp = synthetic.make_inputs(**kw)
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_ref_knots=1024) as e:
        e.set_step_kernel(mode, ch)
        e.set_obstacle_stage(2)
        e.set_package(package)
        if not os.environ.get("FX_NO_TIMING"):
            e.set_timing("kernel")
        e.upload(inp)
        for _ in range(3):
            e.evaluate(); res = e.finish()[0]
        cost, flags = e.costs()
        cm = e.costmap()
        info = e.step_info()
        w = wall(lambda: (e.evaluate(), e.finish()))
        e.evaluate(); res2 = e.finish()[0]
        assert res2["best_index"] == res["best_index"] and res2["n_collisions"] == res["n_collisions"], (res, res2)
        kms = e.last_kernel_ms
    return dict(res=res, cost=cost, flags=flags, cm=cm, info=info, wall=w, kernel_us=kms * 1e3, C=inp.n_candidates)


for name in sys.argv[1:] or ["c3"]:
    a = run(name, 1)
    print(f"{name}: C={a['C']} three launches: wall {a['wall']:.1f} us, device {a['kernel_us']:.1f} us, winner {a['res']['best_index']} "
          f"collisions {a['res']['n_collisions']} feasible {a['res']['n_feasible']}", flush=True)
    for ch in (3, 5, 8, 0):
        b = run(name, 2, ch)
        i = b["info"]
        same_flags = np.array_equal(a["flags"], b["flags"])
        dc = np.abs(a["cost"] - b["cost"]) / np.maximum(np.abs(a["cost"]), 1e-300)
        keys = ("best_index", "best_cost", "n_collisions", "n_feasible", "n_returned")
        same_res = all(a["res"][k] == b["res"][k] for k in keys if k != "best_cost") and list(a["res"]["reason_hist"]) == list(b["res"]["reason_hist"])
        print(f"   one launch (steps/item {ch or 'auto'} -> {i['obstacle_steps_per_item']}, waves {i['obstacle_items']}, step_kernel={i['step_kernel']}): "
              f"wall {b['wall']:.1f} us, device {b['kernel_us']:.1f} us; flags equal {same_flags}, result equal {same_res}, "
              f"cost max rel diff {np.nanmax(dc):.2e} (bitwise {np.array_equal(a['cost'], b['cost'])}), best cost diff {abs(a['res']['best_cost'] - b['res']['best_cost']):.2e}, "
              f"costmap max diff {np.abs(a['cm'] - b['cm']).max():.2e}", flush=True)
