import cProfile, pstats, os, sys, io
sys.path.insert(0, os.getcwd())
import numpy as np
from frenetix_motion_planner_amd import VehicleParams, synthetic
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
cs = synthetic.CoordinateSystem(ref)
s0 = float(cs.ref_pos[40] + 0.1)
x0 = ReactivePlannerState(time_step=0, position=cs.convert_to_cartesian_coords(s0, 0.2), orientation=float(cs.ref_theta[40]), velocity=10.0)
preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
rp = ReactivePlannerHip(PlannerConfig(sampling_min=2, sampling_max=3), VehicleParams())
rp.update_externals(reference_path=ref, x_0=x0, desired_velocity=12.0, predictions=preds)
for _ in range(10): rp.plan()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): rp.plan()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
