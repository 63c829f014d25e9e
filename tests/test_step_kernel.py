"""The whole plan step in ONE launch (csrc/fx_step_kernel.h, fx_set_step_kernel; opt-in) against the three launches it replaces --
walk, obstacle kernel, selection -- and against the oracle: same flag words, counters, winner, collision count; costs bit for bit
at three steps per obstacle item (the phases are the kernels' bodies), to 1e-12 relative at five or eight (the prediction sum is
then grouped by other chunks of the horizon).  Batches, the winner package, the debug flag set, a list that ends inside a tile,
an agent without a single costed candidate."""
import numpy as np
import pytest

from frenetix_motion_planner_amd import _abi, synthetic

pytestmark = pytest.mark.gpu

CASES = {
    "prod_20obs": dict(ref_kind="arc", v0=10.0, grid=(19, 31, 41), n_obstacles=20, lead_gap=25.0),
    "debug_12obs_scurve": dict(ref_kind="scurve", kappa=0.02, v0=9.0, grid=(19, 33, 31), n_obstacles=12, draw_traj_set=True, kinematic_debug=True),
    "horizon5": dict(ref_kind="arc", n_knots=700, v0=11.0, grid=(20, 25, 31), horizon=5.0, n_pred=50, n_obstacles=7),
    "slow_lowvel": dict(ref_kind="arc", v0=1.2, v_des=3.0, grid=(19, 31, 31), n_obstacles=5, seed=3),
}


def _hulls():
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls
    return build_obstacle_hulls


def _run(inps, mode, ch=0, package=False):
    from frenetix_motion_planner_amd.engine import FrenetEngine
    n = sum(i.n_candidates for i in inps)
    with FrenetEngine(max_candidates=n + 64 * len(inps), max_steps=max(i.N for i in inps), max_ref_knots=1024, max_agents=len(inps)) as e:
        e.set_step_kernel(mode, ch)
        e.set_obstacle_stage(2)          # the split step (own obstacle kernel) whatever the grid's wave count
        e.set_package(package)
        res = e.plan_batch(inps)
        info = e.step_info()
        out = []
        for a in range(len(inps)):
            cost, flags = e.costs(a)
            out.append(dict(res=res[a], cost=cost, flags=flags, cm=e.costmap(a)))
        pk = [e.package(a, 0.25) for a in range(len(inps))] if package else None
        res2 = e.plan_batch(inps)      # a second step on the same context: counters, tickets and barrier words are left clean
        for a in range(len(inps)):
            assert res2[a]["best_index"] == res[a]["best_index"] and res2[a]["n_collisions"] == res[a]["n_collisions"]
            assert list(res2[a]["reason_hist"]) == list(res[a]["reason_hist"]) and res2[a]["n_feasible"] == res[a]["n_feasible"]
    return out, info, pk


def _same(a, b, exact):
    keys = ("best_index", "n_collisions", "n_feasible", "n_returned", "n_candidates")
    assert all(a["res"][k] == b["res"][k] for k in keys), (a["res"], b["res"])
    assert list(a["res"]["reason_hist"]) == list(b["res"]["reason_hist"])
    assert np.array_equal(a["flags"], b["flags"])
    if exact:
        assert np.array_equal(a["cost"], b["cost"]) and a["res"]["best_cost"] == b["res"]["best_cost"]
        assert np.array_equal(a["cm"], b["cm"])
    else:
        rel = np.abs(a["cost"] - b["cost"]) / np.maximum(np.abs(a["cost"]), 1e-300)
        assert np.nanmax(rel) < 1e-12 and abs(a["res"]["best_cost"] - b["res"]["best_cost"]) <= 1e-12 * abs(a["res"]["best_cost"])


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("ch", [3, 5, 8, 0])
def test_one_launch_equals_three(name, ch):
    inp = synthetic.make_inputs(hull_builder=_hulls(), **CASES[name])
    three, info3, _ = _run([inp], 1)
    assert info3["obstacle_kernel"] == 1 and info3["step_kernel"] == 0
    one, info1, _ = _run([inp], 2, ch)
    assert info1["step_kernel"] == 1 and (ch == 0 or info1["obstacle_steps_per_item"] == ch)
    _same(three[0], one[0], exact=info1["obstacle_steps_per_item"] == 3 == info3["obstacle_steps_per_item"])


def test_one_launch_against_the_oracle():
    from oracle import oracle
    kw = CASES["prod_20obs"]
    inp = synthetic.make_inputs(hull_builder=_hulls(), **kw)
    one, info, _ = _run([inp], 2, 3)
    assert info["step_kernel"] == 1
    ref = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw), want_planes=False)
    robust = ref["margin"] >= 1e-9
    assert np.array_equal(one[0]["flags"][robust], ref["flags"][robust])
    c = ref["costed"] & robust
    assert (np.abs(one[0]["cost"][c] - ref["cost"][c]) / np.maximum(np.abs(ref["cost"][c]), 1e-12)).max() < 1e-9
    assert one[0]["res"]["best_index"] == ref["result"]["best_index"] and one[0]["res"]["n_collisions"] == ref["result"]["n_collisions"]
    assert ref["collision"].sum() > 50


def test_batch_of_agents_and_the_winner_package():
    hb = _hulls()
    inps = [synthetic.make_inputs(hull_builder=hb, ref_kind="arc", v0=8.0 + a, d0=0.1 * a, grid=(19, 23, 23), n_obstacles=5 + a, seed=a + 1)
            for a in range(3)]
    three, info3, pk3 = _run(inps, 1, package=True)
    one, info1, pk1 = _run(inps, 2, 3, package=True)
    assert info1["step_kernel"] == 1 and info1["agents"] == 3 and info3["step_kernel"] == 0
    for a in range(3):
        _same(three[a], one[a], exact=True)
        assert (pk3[a] is None) == (pk1[a] is None)
        if pk1[a] is not None:
            assert pk1[a].index == pk3[a].index and np.array_equal(pk1[a].block, pk3[a].block) and pk1[a].cost == pk3[a].cost


def test_nothing_costed_and_a_ragged_last_tile():
    """an agent whose candidates are all infeasible (the list stays empty: no tile, no partial, no winner) and one whose list ends
    inside a tile"""
    hb = _hulls()
    none = synthetic.make_inputs(hull_builder=hb, ref_kind="arc", v0=10.0, a0=-40.0, grid=(19, 31, 31), n_obstacles=3)   # |a| above a_max at step 0: everything pre-filtered
    three, _, _ = _run([none], 1)
    one, info, _ = _run([none], 2, 3)
    assert info["step_kernel"] == 1
    _same(three[0], one[0], exact=True)
    assert one[0]["res"]["n_feasible"] == 0 and one[0]["res"]["best_index"] == -1
    ragged = synthetic.make_inputs(hull_builder=hb, ref_kind="arc", v0=10.0, grid=(19, 31, 33), n_obstacles=4)
    three, _, _ = _run([ragged], 1)
    one, info, _ = _run([ragged], 2, 3)
    assert info["step_kernel"] == 1 and one[0]["res"]["n_feasible"] % 64 != 0
    _same(three[0], one[0], exact=True)


def test_not_applicable_keeps_the_three_launches():
    """no bundle / fused obstacle stage / planner-sized grids: the switch is accepted and the step runs as before"""
    from frenetix_motion_planner_amd.engine import FrenetEngine
    inp = synthetic.make_inputs(hull_builder=_hulls(), ref_kind="arc", v0=10.0, level=2, n_obstacles=5)
    with FrenetEngine(max_candidates=4096) as e:
        e.set_step_kernel(2)
        res = e.plan_step(inp)
        assert e.step_info()["step_kernel"] == 0 and res["best_index"] >= 0
        with pytest.raises(Exception):
            e.set_step_kernel(3)
        with pytest.raises(Exception):
            e.set_step_kernel(2, 4)
