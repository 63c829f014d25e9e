import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from oracle import oracle
from tests.test_hip_parity import _random_case
case, g = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([20241008, case]); kw = _random_case(rng); print(kw)
inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
with FrenetEngine(max_candidates=max(inp.n_candidates, 64), max_steps=inp.N, max_pred_steps=max(64, inp.N + 2)) as e:
    if len(sys.argv) > 3: e.set_tuning(*[int(x) for x in sys.argv[3:8]])
    res = e.plan_step(inp); cost, flags = e.costs(); got = e.bundle()[g]
ref = out["planes"][g]
err = np.abs(got - ref) / (1.0 + np.abs(ref).max(axis=1, keepdims=True))
print("flags", hex(flags[g]), hex(out["flags"][g]), "margin", out["margin"][g])
print("per plane err", err.max(axis=1)); i = int(err.max(axis=0).argmax()); print("worst step", i)
names = ["x","y","th","v","a","kap","kapdot","s","d","thcl","sd","sdd","dd","ddd"]
for j in (i - 1, i):
    print("--- step", j)
    for p in range(14): print(names[p], repr(ref[p][j]), repr(got[p][j]), "peak", np.abs(ref[p]).max())
print("sec at step", 1/np.cos(ref[9][i]), "max sec", np.abs(1/np.cos(ref[9])).max())
