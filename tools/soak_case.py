#!/usr/bin/env python3
"""One case of tools/soak_parity.py, with what a failure needs to be understood: the candidates whose planes carry no digit
(tolerance >= 1: count, their (t, v) pairs, conditioning) and, for cost / cost-map disagreements, the offending candidates with
their conditioning and terms.  usage: [FX_SOAK_*=..] python tools/soak_case.py <case>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import synthetic, _abi
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from oracle import oracle
from tests.test_hip_parity import _random_case, compare, FRAGILE, STATE_TOL
from tests.admissible import conditioning_many, kinematic_conditioning_many

case = int(sys.argv[1])
rng = np.random.default_rng([20241008, case])
kw = _random_case(rng)
if os.environ.get("FX_SOAK_MANY"):
    kw["n_obstacles"] = int(rng.integers(9, 49)) if os.environ["FX_SOAK_MANY"] != "2" else int(rng.integers(49, 257))
    if "grid" in kw:
        kw["grid"] = (min(kw["grid"][0], 4), kw["grid"][1], kw["grid"][2])
if os.environ.get("FX_SOAK_COSTS"):
    from frenetix_motion_planner_amd._abi import COST_NAMES
    w = {n: float(rng.uniform(0.1, 5.0)) for n in COST_NAMES if n != "lane_center_offset" and rng.uniform() < 0.5}
    if rng.uniform() < 0.5:   # (drawn behind the ten terms of the earlier soaks: their cases keep their draws)
        w["lane_center_offset"] = float(rng.uniform(0.1, 5.0))
        if rng.uniform() < 0.7:
            kw["lanelets"] = (float(rng.uniform(2.5, 4.5)), int(rng.integers(10, 120)))
    kw["cost_weights"] = w or {"lateral_jerk": 1.0}
if os.environ.get("FX_SOAK_MATRIX") and "stop_point_s" not in kw:
    kw["as_matrix"] = True
if os.environ.get("FX_SOAK_PROJ"):
    kw["pseudo_normal"] = bool(rng.integers(0, 2))
    kw["vertex_tangent"] = "bisector" if rng.integers(0, 2) else "chord"
    if kw.get("ref_kind", "arc") != "scurve":
        kw["knot_jitter"] = 0.3
print(kw)
inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
with FrenetEngine(max_candidates=max(inp.n_candidates, 64), max_steps=inp.N, max_pred_steps=max(64, inp.N + 2), max_obstacles=256) as e:
    if os.environ.get("FX_SOAK_TUNING"):   # the forced decomposition soak_parity.py draws for this case
        tn = (int(rng.choice([0, 1, 2, 4, 8, 16, 32])), int(rng.choice([0, 2, 3, 4])), int(rng.choice([0, 1, 2])),
              int(rng.choice([0, 64, 128, 256])), int(rng.choice([0, 1, 2])))
        sm, ob = int(rng.integers(0, 3)), (int(rng.choice([0, 1, 2])), int(rng.choice([0, 2, 3, 5])))
        print("tuning", tn, "store mode", sm, "obstacle stage", ob)
        e.set_tuning(*tn); e.set_store_mode(sm); e.set_obstacle_stage(*ob)
        try:
            res = e.plan_step(inp)
        except ValueError:
            print("   (does not apply: automatic)")
            e.set_tuning(0, 0, 0, 0, 0); e.set_obstacle_stage(0)
            res = e.plan_step(inp)
    else:
        res = e.plan_step(inp)
    print("launch:", e.step_info())
    cost, flags = e.costs()
    planes_dev = e.bundle() if inp.write_bundle else None
    cm = e.costmap() if inp.write_costmap and len(inp.cost_names) else None
    try:
        compare(e, inp, out, res)
        print("compare: ok")
    except AssertionError as ex:
        print("compare FAILED:", ex)
cond, ck = conditioning_many(out["planes"]), kinematic_conditioning_many(out["planes"])
robust = out["margin"] >= FRAGILE
stored = out["returned"] & robust & (out["costed"] | inp.draw_traj_set)
tol = STATE_TOL + 2e-14 * ck
esc = stored & (tol >= 1.0)
nD = len(inp.d_samp) if inp.sampling_matrix is None else 1
print(f"{inp.n_candidates} candidates, {int(stored.sum())} stored, {int(esc.sum())} with a kinematic tolerance >= 1; their (t, v) pairs: "
      f"{sorted(set((np.nonzero(esc)[0] // max(nD, 1)).tolist()))}; low_vel_mode {inp.low_vel_mode}, v0 {kw.get('v0')}")
for g in np.nonzero(esc)[0][:8]:
    print(f"   candidate {g}: conditioning {ck[g]:.3e}, peak |theta_cl| {np.abs(out['planes'][g, 9]).max():.6f}, peak |v| {np.abs(out['planes'][g, 3]).max():.3e}")
if cm is not None:
    c = out["costed"] & robust & ((flags & _abi.FX_FLAG_COSTED) != 0)
    rel = np.abs(cm - out["costmap"]) / np.maximum(np.abs(out["costmap"]), 1e-9)
    rel[~c] = 0
    g, j = np.unravel_index(np.argmax(rel), rel.shape)
    print(f"largest cost-map disagreement: candidate {g} term {inp.cost_names[j]}: device {cm[g, j]!r} oracle {out['costmap'][g, j]!r} rel {rel[g, j]:.3e}; "
          f"kinematic conditioning {ck[g]:.3e}; cost device {cost[g]!r} oracle {out['cost'][g]!r}")
    print("   the candidate's terms (device | oracle):", {n: (float(cm[g, k]), float(out["costmap"][g, k])) for k, n in enumerate(inp.cost_names)})

ref = out["result"]
print(f"winner device {res['best_index']} (cost {res['best_cost']!r}) oracle {ref['best_index']} (cost {ref['best_cost']!r}); collisions {res['n_collisions']} / {ref['n_collisions']}")
if res["best_index"] != ref["best_index"] and min(res["best_index"], ref["best_index"]) >= 0:
    a, b = res["best_index"], ref["best_index"]
    print(f"   oracle costs of the two: {out['cost'][a]!r} {out['cost'][b]!r} (gap {abs(out['cost'][a] - out['cost'][b]):.3e}); device costs {cost[a]!r} {cost[b]!r}; "
          f"margins {out['margin'][a]:.3e} {out['margin'][b]:.3e}; collision flags oracle {out['collision'][a]} {out['collision'][b]}")
if planes_dev is not None:
    err = (np.abs(planes_dev - out["planes"]) / (1.0 + np.abs(out["planes"]).max(axis=2, keepdims=True))).max(axis=2)
    err[~stored] = 0
    g, pl = np.unravel_index(np.argmax(np.where((ck < 1e3)[:, None], err, 0)), err.shape)
    i = int(np.argmax(np.abs(planes_dev[g, pl] - out["planes"][g, pl])))
    print(f"largest plane error among well-conditioned candidates: candidate {g} plane {pl}: {err[g, pl]:.3e} at step {i}: device {planes_dev[g, pl, i]!r} oracle {out['planes'][g, pl, i]!r}; "
          f"conditioning {ck[g]:.3e}; theta_cl there {out['planes'][g, 9, i]!r}, v {out['planes'][g, 3, i]!r}, d {out['planes'][g, 8, i]!r}, kappa row {out['planes'][g, 5, max(i-1,0):i+2]}")
