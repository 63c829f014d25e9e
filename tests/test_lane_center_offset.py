"""lane_center_offset (partial_cost_functions.py:91-117) -- the one cost term that reads the lanelet network: per trajectory point
the first lanelet that contains it and the distance to that lanelet's centre line, 5 where no lanelet does, averaged over the
points.  The reference asks commonroad-io (`find_lanelet_by_position`) and shapely (`project` / `interpolate`), neither of which is
in the reference tree: include/fxplan.h (FxProblem.n_lane) is the normative definition, parity unpinned.  Here:
the oracle against a NumPy restatement of that definition on the ZAM_Tjunction lanelets and on synthetic lanes, and the HIP path
against the oracle."""
import os

import numpy as np
import pytest

from frenetix_motion_planner_amd import commonroad_xml as crx
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.problem import pack_lanelets

HERE = os.path.dirname(os.path.abspath(__file__))


def restated_cost(lanelets, xs, ys):
    """the definition, point by point, with NumPy (what a reader of partial_cost_functions.py:106-117 would write down)"""
    total = 0.0
    for x, y in zip(xs, ys):
        d = 5.0
        for ll in lanelets:
            poly = np.vstack([ll.left_vertices, ll.right_vertices[::-1]])
            if not (poly[:, 0].min() <= x <= poly[:, 0].max() and poly[:, 1].min() <= y <= poly[:, 1].max()):
                continue
            xi, yi = poly[:, 0], poly[:, 1]
            xj, yj = np.roll(xi, 1), np.roll(yi, 1)
            cross = (yi > y) != (yj > y)
            with np.errstate(divide="ignore", invalid="ignore"):
                hit = cross & (x < (xj - xi) * (y - yi) / (yj - yi) + xi)
            if not (np.count_nonzero(hit) & 1):
                continue
            c = 0.5 * (np.asarray(ll.left_vertices) + np.asarray(ll.right_vertices))
            a, b = c[:-1], c[1:] - c[:-1]
            t = np.clip(((x - a[:, 0]) * b[:, 0] + (y - a[:, 1]) * b[:, 1]) / (b[:, 0] ** 2 + b[:, 1] ** 2), 0.0, 1.0)
            d = float(np.sqrt(np.min((x - (a[:, 0] + t * b[:, 0])) ** 2 + (y - (a[:, 1] + t * b[:, 1])) ** 2)))
            break
        total += d
    return total / len(xs)


def _lane_case(**kw):
    args = dict(ref_kind="scurve", v0=9.0, grid=(3, 5, 9), lanelets=(3.5, 60), draw_traj_set=True, kinematic_debug=True,
                cost_weights=dict(lane_center_offset=2.0, lateral_jerk=0.2))
    args.update(kw)
    return synthetic.make_inputs(**args)


def test_packing_keeps_network_order_and_outlines():
    sc = crx.read_scenario_json(os.path.join(HERE, "golden", "ZAM_Tjunction-1_42_T-1.scenario.json"))
    pk = pack_lanelets(sc)
    ids = list(sc.lanelets)
    assert pk["n"] == len(ids) and pk["poly_off"][0] == 0 and pk["ctr_off"][0] == 0
    for k, lid in enumerate(ids):
        ll = sc.lanelets[lid]
        poly = pk["poly"][pk["poly_off"][k]:pk["poly_off"][k + 1]]
        assert np.array_equal(poly, np.vstack([ll.left_vertices, ll.right_vertices[::-1]]))
        assert np.array_equal(pk["ctr"][pk["ctr_off"][k]:pk["ctr_off"][k + 1]], ll.center_vertices)
        assert pk["bbox"][k, 0] == poly[:, 0].min() and pk["bbox"][k, 3] == poly[:, 1].max()


def test_oracle_equals_the_restated_definition_on_synthetic_lanes():
    from oracle import oracle
    inp = _lane_case()
    out = oracle.plan_step(inp)
    col = inp.cost_names.index("lane_center_offset")
    lanes = synthetic.lanes_along(inp.coordinate_system, 3.5, 60)
    ids = np.nonzero(out["costed"])[0]
    assert len(ids) > 20
    seen = set()
    for g in ids[:: max(1, len(ids) // 40)]:
        want = restated_cost(lanes, out["planes"][g, 0], out["planes"][g, 1])
        assert abs(out["costmap"][g, col] - want) <= 1e-12 * max(1.0, abs(want)), (g, out["costmap"][g, col], want)
        seen.add(round(want, 3))
    assert len(seen) > 5   # inside the own lane, in the neighbour, off the road: not one value
    # weighted into the total as every other term (cost_function.py:78-91)
    w = inp.cost_weights
    tot = sum(w[n] * out["costmap"][ids, k] for k, n in enumerate(inp.cost_names))
    assert np.allclose(tot, out["cost"][ids], rtol=1e-12, atol=1e-12)


def test_oracle_on_the_tjunction_lanelets_and_without_lanelets():
    from oracle import oracle
    sc = crx.read_scenario_json(os.path.join(HERE, "golden", "ZAM_Tjunction-1_42_T-1.scenario.json"))
    from tests.fixtures import golden_names, inputs_from_fixture, load_golden
    name = [n for n in golden_names() if "Tjunction" in n or "tjunction" in n.lower()]
    fx = load_golden(name[0] if name else golden_names()[0])
    inp = inputs_from_fixture(fx, oracle.build_obstacle_hulls)
    inp.cost_weights = dict(lane_center_offset=1.0)
    inp.__dict__.pop("_skey", None)
    inp.lanelets = sc
    inp.__post_init__()
    out = oracle.plan_step(inp)
    ids = np.nonzero(out["costed"])[0]
    lanes = list(sc.lanelets.values())
    for g in ids[:: max(1, len(ids) // 25)]:
        want = restated_cost(lanes, out["planes"][g, 0], out["planes"][g, 1])
        assert abs(out["costmap"][g, 0] - want) <= 1e-12 * max(1.0, abs(want))
    # no lanelets at all: every point is "not on a lanelet" (partial_cost_functions.py:112-115) -> 5
    bare = _lane_case(lanelets=None)
    o2 = oracle.plan_step(bare)
    col = bare.cost_names.index("lane_center_offset")
    assert np.all(o2["costmap"][o2["costed"], col] == 5.0)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [0, 1, 4])
@pytest.mark.parametrize("name", ["lane_center", "lane_center_nolanes"])
def test_hip_vs_oracle(name, lanes):
    from frenetix_motion_planner_amd.engine import FrenetEngine
    from oracle import oracle
    from tests.test_hip_parity import CASES, compare, hip_hulls
    kw = CASES[name]
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
    with FrenetEngine(max_candidates=1024, max_steps=inp.N) as e:
        e.set_tuning(lanes, 0, 0)
        res = e.plan_step(inp)
        info = e.step_info()
        assert not info["grid_kernel"] and info["lanes_per_candidate"] == 1   # a running sum over the horizon: one lane, generic kernel
        compare(e, inp, out, res)
        assert res["best_index"] == out["result"]["best_index"]
        col = inp.cost_names.index("lane_center_offset")
        cm = e.costmap()
        m = out["costed"]
        # the same arithmetic in the same order on (x, y) that agree to the planes' own tolerance: 1e-9 (north star: costs 1e-9 relative)
        assert np.allclose(cm[m, col], out["costmap"][m, col], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_planner_with_lanelets_closed_loop():
    """ReactivePlannerHip.set_lanelets: the term takes part in the selection, and the closed loop's state update keeps it"""
    from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
    from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = CoordinateSystem(ref)
    s0 = float(cs.ref_pos[40] + 0.1)
    x0 = ReactivePlannerState(0, np.asarray(cs.convert_to_cartesian_coords(s0, 0.2)), float(cs.ref_theta[40]), 10.0, 0.0, 0.0, 0.0)
    cfg = PlannerConfig(sampling_min=2, sampling_max=3)
    cfg.cost_weights = dict(cfg.cost_weights, lane_center_offset=3.0)
    p = ReactivePlannerHip(cfg)
    try:
        p.set_lanelets(synthetic.lanes_along(cs, 3.5, 60))
        p.update_externals(reference_path=ref, x_0=x0, desired_velocity=12.0, predictions={})
        first = p.plan()
        assert first is not None and "lane_center_offset" in p.last_step.inputs.cost_names
        k = p.last_step.inputs.cost_names.index("lane_center_offset")
        best = p.optimal_trajectory
        assert 0.0 <= best.costMap["lane_center_offset"][0] < 5.0
        p.update_externals(x_0=x0, predictions={})
        again = p.plan()
        assert again[0][1].position[0] == first[0][1].position[0]
        assert k == p.last_step.inputs.cost_names.index("lane_center_offset")
    finally:
        p.close()


@pytest.mark.gpu
def test_batched_agents_with_their_own_lanelets():
    """every agent of a batched launch brings its own lanelets (they share the grown-on-demand staging block of the road
    boundary): the batch's results are those of the single launches, also after the in-place state update"""
    from frenetix_motion_planner_amd.engine import FrenetEngine
    from tests.test_hip_parity import CASES, hip_hulls
    kws = [dict(CASES["lane_center"], v0=8.0 + a, seed=a, lanelets=(3.0 + 0.5 * a, 40 + 20 * a)) for a in range(3)]
    kws[2] = dict(kws[2], lanelets=None)   # one agent without lanelets: 5 m everywhere
    agents = [synthetic.make_inputs(hull_builder=hip_hulls(), **kw) for kw in kws]
    with FrenetEngine(max_candidates=4096, max_steps=agents[0].N, max_agents=3) as batch:
        got = batch.plan_batch(agents)
        got2 = batch.plan_batch(agents)            # resident: the in-place update path
        maps = [batch.costmap(a) for a in range(3)]
    for a, inp in enumerate(agents):
        with FrenetEngine(max_candidates=1024, max_steps=inp.N) as one:
            ref = one.plan_step(inp)
            cm = one.costmap()
        for k in ("best_index", "best_cost", "n_feasible", "n_collisions"):
            assert got[a][k] == ref[k] == got2[a][k], (a, k)
        assert np.array_equal(maps[a], cm)
    col = agents[2].cost_names.index("lane_center_offset")
    assert np.all(maps[2][:, col][maps[2][:, col] != 0] == 5.0)
