import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
for K, grid in ((20, (19, 51, 51)), (64, (19, 51, 51)), (100, (19, 51, 51)), (256, (19, 51, 51)), (100, (5, 9, 14))):
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, grid=grid, n_obstacles=K, lead_gap=25.0)
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_obstacles=256) as eng:
        eng.set_timing("kernel"); eng.upload(inp)
        for _ in range(20): eng.evaluate(); eng.finish()
        ts = []
        for _ in range(30): eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
        print(K, grid, inp.n_candidates, "kernel us", round(float(np.median(ts)) * 1e3, 1), eng.step_info())
