#!/usr/bin/env python3
"""The reference adapter's plan step (800-row C x 13 sampling matrix through frenetix_compat.TrajectoryHandler) with and without
the product-grid recognition: evaluation kernel and step wall time; results must be identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from frenetix_motion_planner_amd.frenetix_compat import product_grid_of
from frenetix_motion_planner_amd.problem import PlanInputs
import copy

base = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=5.6, level=2, n_obstacles=5, draw_traj_set=True,
                             kinematic_debug=True)
t = np.union1d(base.t_samp, [3.0]); v = np.union1d(base.v_samp, [base.x0_lon[1]]); d = np.union1d(base.d_samp, [base.x0_lat[0]])
as_m = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=5.6, level=2, n_obstacles=5, draw_traj_set=True,
                             kinematic_debug=True, as_matrix=True)
out = {}
with FrenetEngine(max_candidates=4096) as eng:
    eng.set_timing("kernel")
    for name in ("matrix", "ranges"):
        inp = copy.copy(as_m)
        if name == "ranges":
            g = product_grid_of(as_m.sampling_matrix)
            assert g is not None
            inp.sampling_matrix = None
            inp.t_samp, inp.v_samp, inp.d_samp = g
            inp.__post_init__()
        eng.upload(inp)
        for _ in range(50):
            eng.evaluate(); eng.finish()
        tw = []
        for _ in range(300):
            a = time.perf_counter(); eng.evaluate(); r = eng.finish()[0]; tw.append(time.perf_counter() - a)
        ev, st = eng.kernel_times(200)
        cost, flags = eng.costs()
        out[name] = (r, cost, flags)
        print(f"{name}: {inp.n_candidates} candidates  evaluation kernel {np.median(ev) * 1e3:.1f} us  step p50 {np.median(tw) * 1e6:.1f} us  "
              f"winner {r['best_index']}  info {eng.step_info()['grid_kernel']}", flush=True)
a, b = out["matrix"], out["ranges"]
print("identical:", a[0]["best_index"] == b[0]["best_index"] and a[0]["best_cost"] == b[0]["best_cost"] and np.array_equal(a[1], b[1], equal_nan=True)
      and np.array_equal(a[2], b[2]))
