#!/usr/bin/env python3
"""One soak case whose candidates escaped the plane assertion (tolerance >= 1): what their conditioning is and how far the device
is from the oracle on every plane.  usage: [FX_SOAK_MANY=1] python tools/dbg_escaped.py <case>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from oracle import oracle
from tests.test_hip_parity import _random_case, STATE_TOL
from tests.admissible import KINEMATIC_PLANES, conditioning_many, kinematic_conditioning_many

case = int(sys.argv[1])
rng = np.random.default_rng([20241008, case])
kw = _random_case(rng)
if os.environ.get("FX_SOAK_MANY"):
    kw["n_obstacles"] = int(rng.integers(9, 49)) if os.environ["FX_SOAK_MANY"] != "2" else int(rng.integers(49, 257))
    if "grid" in kw:
        kw["grid"] = (min(kw["grid"][0], 4), kw["grid"][1], kw["grid"][2])
print(kw)
inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
with FrenetEngine(max_candidates=max(inp.n_candidates, 64), max_steps=inp.N, max_pred_steps=max(64, inp.N + 2), max_obstacles=256) as e:
    res = e.plan_step(inp)
    got = e.bundle(0)
refp = out["planes"]
cond, ck = conditioning_many(refp), kinematic_conditioning_many(refp)
tol = np.repeat((STATE_TOL + 2e-14 * cond)[:, None], 14, axis=1)
tol[:, KINEMATIC_PLANES] = (STATE_TOL + 2e-14 * ck)[:, None]
stored = out["returned"] & (out["costed"] | inp.draw_traj_set)
esc = np.nonzero(stored & (tol >= 1.0).any(axis=1))[0]
err = (np.abs(got - refp) / (1.0 + np.abs(refp).max(axis=2, keepdims=True))).max(axis=2)
np.set_printoptions(precision=3, linewidth=200)
print("escaped candidates", esc, "of", int(stored.sum()), "low_vel", inp.low_vel_mode)
for g in esc:
    th = refp[g, 9]
    print(f"cand {g}: cond {cond[g]:.3g} cond_kin {ck[g]:.3g}  max|theta_cl| {np.abs(th).max():.17g}  pi/2 - that {np.pi/2 - np.abs(th).max():.3g}")
    print("   rel err per plane     ", err[g])
    print("   ref peak per plane    ", np.abs(refp[g]).max(axis=1))
    print("   device peak per plane ", np.abs(got[g]).max(axis=1))
    i = int(np.argmax(np.abs(th)))
    print(f"   at step {i}: ref v {refp[g,3,i]:.6g} dev v {got[g,3,i]:.6g}  ref kappa {refp[g,5,i]:.6g} dev {got[g,5,i]:.6g}  d' ref {refp[g,12,i]:.6g} dev {got[g,12,i]:.6g} s' ref {refp[g,10,i]:.6g} dev {got[g,10,i]:.6g}")
