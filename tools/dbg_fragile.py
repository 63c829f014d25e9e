import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from oracle import oracle
from tests.test_hip_parity import _random_case
from tests.admissible import plane_error, state_tolerance, cost_tolerance
case, g = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([20241008, case]); kw = _random_case(rng); print(kw)
inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
ref_inp = synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw)
out = oracle.plan_step(ref_inp)
with FrenetEngine(max_candidates=max(inp.n_candidates, 64), max_steps=inp.N, max_pred_steps=max(64, inp.N + 2)) as e:
    res = e.plan_step(inp); cost, flags = e.costs(); got = e.bundle()[g]
outs = oracle.admissible_outcomes(ref_inp, g, out["frag_sites"][g])
print("gpu flags", hex(flags[g]), "cost", cost[g], "margin", out["margin"][g])
for o in outs:
    err = np.abs(got - o["planes"]) / (1.0 + np.abs(o["planes"]).max(axis=1, keepdims=True))
    print(hex(o["flags"]), "cost", o["cost"], "rel", abs(cost[g]-o["cost"])/max(abs(o["cost"]),1e-12), "ctol", cost_tolerance(o["planes"]), "plane err", err.max(), np.unravel_index(err.argmax(), err.shape), "tol", state_tolerance(o["planes"], 1e-7))
    print("   per plane", np.round(err.max(axis=1), 12))
print("s_dot", out["planes"][g][10][-8:], "sdot gpu", got[10][-8:])
