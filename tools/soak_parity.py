#!/usr/bin/env python3
"""Parity soak on the GPU box: N seeded random scenarios (tests/test_hip_parity.py::_random_case) against the oracle.
usage: python tools/soak_parity.py [first_case] [n_cases]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
from oracle import oracle
from tests.test_hip_parity import _random_case, compare, FRAGILE, PARITY_STATS

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
stats = dict(cands=0, winners=0, collided=0, fragile=0)
for case in range(first, first + n):
    rng = np.random.default_rng([20241008, case])
    kw = _random_case(rng)
    if os.environ.get("FX_SOAK_MANY"):   # crowded scenes: 9 ... 48 obstacles (the generator stops at 8); =2: 49 ... 256 (multi-word masks)
        kw["n_obstacles"] = int(rng.integers(9, 49)) if os.environ["FX_SOAK_MANY"] != "2" else int(rng.integers(49, 257))
        if "grid" in kw:
            kw["grid"] = (min(kw["grid"][0], 4), kw["grid"][1], kw["grid"][2])
    if os.environ.get("FX_SOAK_COSTS"):  # random cost functions over all eleven terms (windowed costs -> generic kernel)
        from frenetix_motion_planner_amd._abi import COST_NAMES
        w = {n: float(rng.uniform(0.1, 5.0)) for n in COST_NAMES if n != "lane_center_offset" and rng.uniform() < 0.5}
        if rng.uniform() < 0.5:   # (drawn behind the ten terms of the earlier soaks: their cases keep their draws)
            w["lane_center_offset"] = float(rng.uniform(0.1, 5.0))
            if rng.uniform() < 0.7:
                kw["lanelets"] = (float(rng.uniform(2.5, 4.5)), int(rng.integers(10, 120)))
        kw["cost_weights"] = w or {"lateral_jerk": 1.0}
    if os.environ.get("FX_SOAK_MATRIX") and "stop_point_s" not in kw:   # the adapter's C x 13 sampling matrix
        kw["as_matrix"] = True
    if os.environ.get("FX_SOAK_PROJ"):   # the other readings of the projection on jittered knots
        kw["pseudo_normal"] = bool(rng.integers(0, 2))
        kw["vertex_tangent"] = "bisector" if rng.integers(0, 2) else "chord"
        if kw.get("ref_kind", "arc") != "scurve":
            kw["knot_jitter"] = 0.3
    try:
        inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
        out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
        with FrenetEngine(max_candidates=max(inp.n_candidates, 64), max_steps=inp.N, max_pred_steps=max(64, inp.N + 2), max_obstacles=256) as e:
            tail = bool(os.environ.get("FX_SOAK_TAIL"))   # the step's end inside the evaluation kernel whatever the size (fx_tail.h)
            if tail:
                e.set_fused_selection(2)
                e.set_obstacle_stage(1)
            step = (lambda i: e.plan_step_packaged(i, yaw_rate0=0.0)) if tail else (lambda i: (e.plan_step(i), None))
            pkg = None
            stepk = bool(os.environ.get("FX_SOAK_STEPK"))   # the whole step in ONE launch where it applies (fx_step_kernel.h): two lanes per
            if stepk:                                        # candidate on two waves, obstacle stage deferred, 3 / 5 / 8 steps per item
                e.set_step_kernel(2, int(rng.choice([0, 3, 5, 8])))
                e.set_tuning(2, 2, 2, 256, 2)
                e.set_obstacle_stage(2, 0)
                e.set_package(bool(rng.integers(0, 2)))
                try:
                    res = e.plan_step(inp)
                except ValueError:   # (no obstacles / a windowed cost / a road boundary: the forced stage does not apply)
                    e.set_tuning(0, 0, 0, 0, 0); e.set_obstacle_stage(0); e._resident_key = None
                    res = e.plan_step(inp)
                stats["step_kernel_steps"] = stats.get("step_kernel_steps", 0) + e.step_info()["step_kernel"]
                compare(e, inp, out, res)
                res = e.plan_step(inp)      # a second step on the same context (barrier words, tickets, counters left clean)
            elif os.environ.get("FX_SOAK_TUNING"):  # also walk through the work decompositions / kernel variants
                tn = (int(rng.choice([0, 1, 2, 4, 8, 16, 32])), int(rng.choice([0, 2, 3, 4])), int(rng.choice([0, 1, 2])),
                      int(rng.choice([0, 64, 128, 256])), int(rng.choice([0, 1, 2])))
                e.set_tuning(*tn)
                e.set_store_mode(int(rng.integers(0, 3)))
                e.set_obstacle_stage(1 if tail else int(rng.choice([0, 1, 2])), int(rng.choice([0, 2, 3, 5])))   # fused / its own kernel
                try:
                    res, pkg = step(inp)
                except ValueError:  # a forced variant that does not apply to this case
                    e.set_tuning(0, 0, 0, 0, 0)
                    e.set_obstacle_stage(1 if tail else 0)
                    e._resident_key = None
                    res, pkg = step(inp)
            else:
                res, pkg = step(inp)
            compare(e, inp, out, res)
            if tail:
                info = e.step_info()
                stats["tail_steps"] = stats.get("tail_steps", 0) + (1 if info["tail"] else 0)
                assert (pkg is None) == (res["best_index"] < 0)
                if pkg is not None:   # the package the tail gathered = the classic read-back of the same candidate
                    cand = e.candidate(res["best_index"])
                    assert pkg.index == res["best_index"] and pkg.cost == res["best_cost"] == cand["cost"] and pkg.flags == cand["flags"]
                    if cand["planes"] is not None:
                        assert np.array_equal(pkg.planes, cand["planes"]) and np.array_equal(pkg.lon, cand["lon"]) and pkg.tau_lat == cand["tau_lat"]
                    stats["packages"] = stats.get("packages", 0) + 1
            if os.environ.get("FX_SOAK_TOPK"):   # the k best collision-free candidates against NumPy on the engine's own costs / flags
                from frenetix_motion_planner_amd import _abi
                k = int(rng.integers(1, 65))
                tc, ti = e.topk(k)
                cost, flags = e.costs()
                ok = ((flags & _abi.FX_FLAG_SELECTABLE) != 0) & ((flags & (_abi.FX_FLAG_COLLISION | _abi.FX_FLAG_BOUNDARY)) == 0) & ~np.isnan(cost)
                ids = np.nonzero(ok)[0]
                order = ids[np.lexsort((ids, cost[ids]))][:k]
                assert list(ti[0][:len(order)]) == list(order + inp.shard_begin) and np.all(ti[0][len(order):] == -1), ("topk", k)
                assert np.array_equal(tc[0][:len(order)], cost[order])
                stats["topk"] = stats.get("topk", 0) + 1
            robust = bool(np.all(out["margin"] >= FRAGILE))
            if robust and res["best_index"] != out["result"]["best_index"]:
                # compare() has asserted that the two winners' costs lie within 1e-9 relative of each other; what may still differ
                # is a TIE: two candidates whose costs agree to the last ulp or two (mirror-image lateral offsets) sort by index
                # on one side and by that ulp on the other.  Counted, and bounded at the end of the run.
                a, b = res["best_index"], out["result"]["best_index"]
                assert a >= 0 and b >= 0 and abs(out["cost"][a] - out["cost"][b]) <= 8 * np.spacing(abs(out["cost"][b])), (a, b)
                stats["tie_decided"] = stats.get("tie_decided", 0) + 1
            elif robust:
                assert res["n_collisions"] == out["result"]["n_collisions"]
        stats["cands"] += inp.n_candidates
        stats["winners"] += out["result"]["best_index"] >= 0
        stats["collided"] += int(out["collision"].sum())
        stats["fragile"] += int((out["margin"] < FRAGILE).sum())
    except Exception as ex:
        bad += 1
        print("CASE", case, "FAILED:", repr(ex)[:300], kw, flush=True)
print(f"soak: {n} cases from {first}: {bad} failures; {stats}; how they were checked: {PARITY_STATS}", flush=True)
ck = max(PARITY_STATS["checked"], 1)
print(f"   fixed 1e-9: {PARITY_STATS['fixed'] / ck:.4%}  conditioning-scaled: {PARITY_STATS['scaled'] / ck:.4%}  magnitude only (tolerance >= 1): "
      f"{PARITY_STATS['escaped'] / ck:.5%}  fragile (admissible outcomes): {stats['fragile'] / max(stats['cands'], 1):.3%} of all candidates, "
      f"{PARITY_STATS['fragile_same'] / max(PARITY_STATS['fragile'], 1):.3%} of them with the reference's own flag word", flush=True)
if stats.get("tie_decided", 0) > max(1, 2e-4 * n):
    print("soak: too many winners decided by a last-ulp tie", flush=True)
    bad += 1
if PARITY_STATS["escaped"] > max(3, 5e-4 * ck) or PARITY_STATS["fixed"] < 0.97 * ck:
    print("soak: TOO MANY candidates went through the scaled / unasserted doors", flush=True)
    bad += 1
sys.exit(1 if bad else 0)
