#!/usr/bin/env python3
"""End-to-end plan-step times on the GPU box: host buffers in -> winner on the host.
  * engine.plan_step(inputs): pack + H2D upload of the step's inputs + launch + result poll (the PCIe-inclusive step),
  * engine.evaluate()+finish(): inputs resident (what bench.py times),
  * ReactivePlannerHip.plan(): the planner front-end on planner-sized grids (sampling levels 2..4),
  * reading one TrajectorySample (14 x S doubles) and the whole bundle back."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import VehicleParams, synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState


def p50(f, n=200, warm=20):
    for _ in range(warm): f()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts) * 1e6)


out = {}
for label, kw in (("config2_modeB", dict(grid=(19, 51, 51))),
                  ("config2_modeA", dict(grid=(19, 51, 51), write_bundle=False, write_costmap=False)),
                  ("config3_modeB", dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0))):
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, hull_builder=build_obstacle_hulls, **kw)
    with FrenetEngine(max_candidates=inp.n_candidates + 64) as eng:
        eng.upload(inp)
        resident = p50(lambda: (eng.evaluate(), eng.finish()))
        upload = p50(lambda: eng.plan_step(inp))
        rec = {"resident_us": resident, "with_upload_us": upload, "candidates": inp.n_candidates}
        if inp.write_bundle:
            rec["read_one_sample_us"] = p50(lambda: eng.sample(1234), n=100)
            t = time.perf_counter(); eng.bundle(); rec["read_whole_bundle_ms"] = (time.perf_counter() - t) * 1e3
        out[label] = rec

ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
cs = synthetic.CoordinateSystem(ref)
s0 = float(cs.ref_pos[40] + 0.1)
x0 = ReactivePlannerState(time_step=0, position=cs.convert_to_cartesian_coords(s0, 0.2), orientation=float(cs.ref_theta[40]), velocity=10.0)
preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
for lvl in (2, 3, 4):
    rp = ReactivePlannerHip(PlannerConfig(sampling_min=lvl, sampling_max=lvl + 1), VehicleParams())
    rp.update_externals(reference_path=ref, x_0=x0, desired_velocity=12.0, predictions=preds)
    t = p50(lambda: rp.plan(), n=60, warm=5)
    out[f"planner_level{lvl}"] = {"plan_us": t, "candidates": int(rp.last_step.n_candidates)}
    rp.close()
print(json.dumps(out, indent=1))
