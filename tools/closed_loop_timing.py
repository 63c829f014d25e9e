#!/usr/bin/env python3
"""Host + device time of a planner's closed-loop step (update_externals + plan) on the HIP engine: BASELINE config 1's shape
(level 2: 630 candidates x 31 samples, 5 predicted obstacles)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState

ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
cs = CoordinateSystem(ref)
s0 = float(cs.ref_pos[40] + 0.1)
x0 = ReactivePlannerState(0, np.asarray(cs.convert_to_cartesian_coords(s0, 0.2)), float(cs.ref_theta[40]), 10.0, 0.0, 0.0, 0.0)
preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
for lvl in (2, 4):
    p = ReactivePlannerHip(PlannerConfig(sampling_min=lvl, sampling_max=lvl + 1))
    p.update_externals(reference_path=ref, x_0=x0, desired_velocity=12.0, predictions=preds)
    for _ in range(20):
        p.plan()
    n = 300
    tp, tu = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        p.update_externals(x_0=x0, predictions=preds)
        t1 = time.perf_counter()
        pair = p.plan()
        _ = pair[0][1], pair[2][1]
        t2 = time.perf_counter()
        tu.append(t1 - t0); tp.append(t2 - t1)
    print(f"level {lvl}: {p.last_step.n_candidates} candidates  update_externals p50 {np.median(tu)*1e6:.1f} us  plan() p50 {np.median(tp)*1e6:.1f} us "
          f"p95 {np.percentile(tp, 95)*1e6:.1f} us", flush=True)
    if lvl == 2 and '--profile' in sys.argv:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(200):
            p.update_externals(x_0=x0, predictions=preds); p.plan()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(45)
    p.close()
