import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
kw = {'ref_kind': 'scurve', 'kappa': 0.011584531232118365, 'n_knots': 431, 'spacing': 0.5416929461901154, 'v0': 0.0, 'a0': -0.8412594994273541, 'd0': 1.04, 'dd0': 0.21527436530497918, 'ddd0': 0.24163754029566292, 'horizon': 3.0, 'v_des': 3.4616339425603626, 'n_obstacles': 4, 'n_pred': 30, 'draw_traj_set': True, 'kinematic_debug': True, 'seed': 744382204, 'lead_gap': 14.410096316730538, 'level': 1}
inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
bad = 0
ref = None
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    with FrenetEngine(max_candidates=max(inp.n_candidates, 64), max_steps=inp.N, max_pred_steps=64) as e:
        e.set_tuning(32, 3, 1, 256, 0); e.set_store_mode(0); e.set_obstacle_stage(stage, 5)
        res = e.plan_step(inp)
        key = (res["n_returned"], res["n_feasible"], res["best_index"], res["n_collisions"], tuple(res["reason_hist"]))
        if ref is None: ref = key
        if key != ref:
            bad += 1
            print("MISMATCH it", it, key, "ref", ref, e.step_info(), flush=True)
        # several steps on the same engine too
        for _ in range(3):
            e.evaluate(); r2 = e.finish()[0]
            k2 = (r2["n_returned"], r2["n_feasible"], r2["best_index"], r2["n_collisions"], tuple(r2["reason_hist"]))
            if k2 != ref:
                bad += 1; print("MISMATCH resident it", it, k2, flush=True)
print("stage", stage, "done, bad =", bad, "ref", ref)
