#!/usr/bin/env python3
"""Wall-clock segments of ReactivePlannerHip.plan() (perf_counter wrappers, no profiler): where a closed-loop step spends its time."""
import os, sys, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic, problem, engine as engine_mod
from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
import frenetix_motion_planner_amd.reactive_planner as rp

acc = collections.defaultdict(list)
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name].append(time.perf_counter() - t); return r
    setattr(obj, name, g)

ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
cs = CoordinateSystem(ref)
s0 = float(cs.ref_pos[40] + 0.1)
x0 = ReactivePlannerState(0, np.asarray(cs.convert_to_cartesian_coords(s0, 0.2)), float(cs.ref_theta[40]), 10.0, 0.0, 0.0, 0.0)
preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
p = ReactivePlannerHip(PlannerConfig(sampling_min=2, sampling_max=3))
p.update_externals(reference_path=ref, x_0=x0, desired_velocity=12.0, predictions=preds)
for _ in range(30):
    p.plan()
wrap(rp, "pack_predictions")
for n in ("_inputs_for_level", "_get_optimal_trajectory", "_consume_result", "plan_finish", "_compute_trajectory_pair", "update_externals",
          "_compute_initial_states"):
    wrap(p, n)
wrap(p.engine, "plan_step_packaged")
wrap(p.engine, "_state_update_of")
wrap(rp, "PlanInputs")
tot = []
for _ in range(400):
    p.update_externals(x_0=x0, predictions=preds)
    t = time.perf_counter(); pair = p.plan(); _ = pair[0][1], pair[2][1]; tot.append(time.perf_counter() - t)
print(f"plan() p50 {np.median(tot)*1e6:.1f} us (with the wrappers)")
for k, v in acc.items():
    print(f"   {k:28s} p50 {np.median(v)*1e6:7.1f} us  x{len(v)//400}")
p.close()
