#!/usr/bin/env python3
"""Parity soak of BATCHED launches on the GPU box: every case is 2 ... 5 random agents (tests/test_hip_parity.py::_random_case:
different references, grids, horizons, obstacle counts, flag sets) evaluated in ONE launch, each agent against the oracle.
usage: python tools/soak_batch.py [first_case] [n_cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from oracle import oracle
from tests.test_hip_parity import _random_case, compare, FRAGILE

first = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
stats = dict(agents=0, cands=0)
for case in range(first, first + n):
    rng = np.random.default_rng([20241008, case])
    kws = [_random_case(rng) for _ in range(int(rng.integers(2, 6)))]
    for kw in kws:   # one step interval and one bundle mode per launch are not required; keep the generator's freedom
        kw.pop("stop_point_s", None) if rng.uniform() < 0.5 else None
        if os.environ.get("FX_SOAK_MANY") and rng.uniform() < 0.5:   # some agents in crowded scenes, a few beyond 64 obstacles
            kw["n_obstacles"] = int(rng.integers(9, 49)) if rng.uniform() < 0.7 else int(rng.integers(65, 200))
    try:
        inps = [synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw) for kw in kws]
        outs = [oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw)) for kw in kws]
        cap = sum(max(i.n_candidates, 64) + 64 for i in inps)
        with FrenetEngine(max_candidates=cap, max_steps=max(i.N for i in inps), max_pred_steps=max(64, max(i.N for i in inps) + 2),
                          max_obstacles=256, max_agents=len(inps)) as e:
            res = e.plan_batch(inps)
            for a, (inp, out) in enumerate(zip(inps, outs)):
                compare(e, inp, out, res[a], agent=a)
                if bool(np.all(out["margin"] >= FRAGILE)):
                    ga, gb = res[a]["best_index"], out["result"]["best_index"]
                    if ga != gb:
                        # the rule of soak_parity.py: compare() has held the two winners' costs to 1e-9; what may still differ is a TIE --
                        # two candidates whose costs agree to the last ulp or two (here: the lateral grid's -1.2000000000000002 beside the
                        # appended current offset -1.2, case 2170242) sort by index on one side and by that ulp on the other.  Counted,
                        # and bounded at the end of the run.
                        assert ga >= 0 and gb >= 0 and abs(out["cost"][ga] - out["cost"][gb]) <= 8 * np.spacing(abs(out["cost"][gb])), (a, ga, gb)
                        stats["tie_decided"] = stats.get("tie_decided", 0) + 1
                    else:
                        assert res[a]["n_collisions"] == out["result"]["n_collisions"], a
            res2 = e.plan_batch(inps)   # the in-place update path with unchanged inputs
            for a in range(len(inps)):
                assert res2[a]["best_index"] == res[a]["best_index"] and res2[a]["best_cost"] == res[a]["best_cost"], a
        stats["agents"] += len(inps); stats["cands"] += sum(i.n_candidates for i in inps)
    except Exception as ex:
        bad += 1
        print("CASE", case, "FAILED:", repr(ex)[:300], [(k.get("grid"), k.get("level"), k["horizon"], k["n_obstacles"]) for k in kws], flush=True)
if stats.get("tie_decided", 0) > max(1, 2e-4 * stats["agents"]):
    print("batch soak: too many winners decided by a last-ulp tie", flush=True)
    bad += 1
print(f"batch soak: {n} cases from {first}: {bad} failures; {stats}", flush=True)
