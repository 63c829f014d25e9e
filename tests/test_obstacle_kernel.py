"""The obstacle stage as its own (candidate x step)-parallel kernel (csrc/fx_obstacle_kernel.h, fx_set_obstacle_stage) against the
CPU oracle and against the stage fused into the walk: prediction cost collision_probability.py:264-299, collision walk
planner.py:329-392 / collision_check.py:110-200."""
import numpy as np
import pytest

from frenetix_motion_planner_amd import _abi, synthetic
from tests.fixtures import golden_names, inputs_from_fixture, load_golden
from tests.test_hip_parity import CASES, RESULT_KEYS, compare, hip_hulls

pytestmark = pytest.mark.gpu

STEPS_PER_ITEM = [2, 3, 5]
OBST_CASES = sorted(n for n, kw in CASES.items() if kw.get("n_obstacles") and not kw.get("cost_weights"))


@pytest.fixture(scope="module")
def eng():
    from frenetix_motion_planner_amd.engine import FrenetEngine
    e = FrenetEngine(max_candidates=120_000, max_steps=60, max_ref_knots=1024, max_obstacles=64, max_pred_steps=64, max_agents=8)
    yield e
    e.set_obstacle_stage(0)
    e.close()


@pytest.mark.parametrize("steps", STEPS_PER_ITEM)
@pytest.mark.parametrize("name", OBST_CASES)
def test_obstacle_kernel_vs_oracle(eng, name, steps):
    from oracle import oracle
    kw = CASES[name]
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    ref_inp = synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw)
    out = oracle.plan_step(ref_inp)
    eng.set_obstacle_stage(2, steps)
    try:
        res = eng.plan_step(inp)
        assert eng.step_info()["obstacle_kernel"] == 1
        compare(eng, inp, out, res, ref_inp=ref_inp)
    finally:
        eng.set_obstacle_stage(0)


@pytest.mark.parametrize("wg", ["0", "1"])
@pytest.mark.parametrize("steps", STEPS_PER_ITEM)
@pytest.mark.parametrize("name", ["dense_debug_obs", "dense_prod_obs", "dense_horizon5", "stop_dense_obs"])
def test_obstacle_kernel_items_and_workgroups(eng, monkeypatch, name, steps, wg):
    """the chunks of a tile as single-wave items meeting in the selection kernel, and as the waves of one workgroup meeting in
    LDS (the launch picks by tile count; FX_OBST_WG forces either): the same result, and step_info names the one that ran"""
    from oracle import oracle
    kw = CASES[name]
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
    monkeypatch.setenv("FX_OBST_WG", wg)
    eng.set_obstacle_stage(2, steps)
    try:
        res = eng.plan_step(inp)
        info = eng.step_info()
        assert info["obstacle_kernel"] == 1
        chunks = -(-inp.N // steps)   # steps 1 .. N (step 0 is the current state)
        assert info["obstacle_workgroup_waves"] == (chunks if wg == "1" and chunks <= 16 else 0)
        compare(eng, inp, out, res)
    finally:
        eng.set_obstacle_stage(0)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 32])
@pytest.mark.parametrize("name", ["dense_debug_obs", "dense_prod_obs", "dense_horizon5", "stop_dense_obs"])
def test_obstacle_kernel_behind_every_walk_decomposition(eng, name, lanes, variant):
    """the walk in front of the obstacle kernel split over 1 ... 32 lanes per candidate, generic and grid kernel"""
    from oracle import oracle
    kw = CASES[name]
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
    eng.set_tuning(lanes, 0, variant)
    eng.set_obstacle_stage(2)
    try:
        try:
            res = eng.plan_step(inp)
        except ValueError as e:
            if "grid kernel" in str(e):
                pytest.skip("grid kernel not applicable to this case (LDS budget)")
            raise
        assert eng.step_info()["obstacle_kernel"] == 1
        compare(eng, inp, out, res)
    finally:
        eng.set_tuning(0, 0, 0)
        eng.set_obstacle_stage(0)


@pytest.mark.parametrize("name", [n for n in golden_names() if "obs" in n])
def test_obstacle_kernel_on_golden_cases(eng, name):
    from oracle import oracle
    fx = load_golden(name)
    inp = inputs_from_fixture(fx, hip_hulls())
    ref_inp = inputs_from_fixture(fx, oracle.build_obstacle_hulls)
    out = oracle.plan_step(ref_inp)
    eng.set_obstacle_stage(2)
    try:
        try:
            res = eng.plan_step(inp)
        except ValueError as e:
            if "obstacle kernel forced" in str(e):
                pytest.skip("windowed cost terms keep the horizon in one lane of the generic kernel")
            raise
        assert eng.step_info()["obstacle_kernel"] == 1
        compare(eng, inp, out, res, ref_inp=ref_inp)
    finally:
        eng.set_obstacle_stage(0)


@pytest.mark.parametrize("collision", [True, False])
@pytest.mark.parametrize("name", OBST_CASES)
def test_obstacle_kernel_equals_fused_stage(eng, name, collision):
    """Same decisions bit for bit (flag words, counters, winner, collision count), every plane and every other cost term
    identical; the prediction cost summed in another order: 1e-12."""
    kw = dict(CASES[name], collision=collision)
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    outs = []
    try:
        for stage in (1, 2):
            eng.set_obstacle_stage(stage)
            res = eng.plan_step(inp)
            assert eng.step_info()["obstacle_kernel"] == stage - 1
            outs.append((res, *eng.costs(), eng.bundle(), eng.costmap()))
            # resident re-evaluation: the obstacle kernel reads cost[] the walk rewrites every step
            eng.evaluate()
            again = eng.finish()[0]
            for k in RESULT_KEYS:
                assert again[k] == res[k], k
    finally:
        eng.set_obstacle_stage(0)
    a, b = outs
    assert np.array_equal(a[2], b[2]), "flag words"
    assert np.array_equal(a[3], b[3]), "planes"
    ip = inp.cost_names.index("prediction")
    others = [n for n in range(len(inp.cost_names)) if n != ip]
    assert np.array_equal(a[4][:, others], b[4][:, others])
    assert np.allclose(a[4][:, ip], b[4][:, ip], rtol=1e-12, atol=0)
    assert np.allclose(a[1], b[1], rtol=1e-12, atol=0)
    for k in ("best_index", "n_returned", "n_feasible", "n_collisions", "reason_hist"):
        assert a[0][k] == b[0][k], k


def test_obstacle_kernel_without_a_prediction_term(eng):
    """collision stage only: the cost function has no prediction term, the walk's cost stands"""
    from oracle import oracle
    kw = dict(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=12, lead_gap=20.0,
              cost_weights=dict(lateral_jerk=0.2, longitudinal_jerk=0.2, velocity_offset=1.0, distance_to_reference_path=5.0))
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
    assert out["collision"].sum() > 0
    eng.set_obstacle_stage(2)
    try:
        res = eng.plan_step(inp)
        assert eng.step_info()["obstacle_kernel"] == 1
        compare(eng, inp, out, res)
    finally:
        eng.set_obstacle_stage(0)


def test_obstacle_kernel_prediction_term_last(eng):
    """no term behind the prediction term (velocity_offset off): cost_tail stays unused"""
    from oracle import oracle
    kw = dict(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=7,
              cost_weights=dict(lateral_jerk=0.2, longitudinal_jerk=0.2, prediction=0.3, distance_to_reference_path=5.0))
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
    eng.set_obstacle_stage(2)
    try:
        res = eng.plan_step(inp)
        compare(eng, inp, out, res)
    finally:
        eng.set_obstacle_stage(0)


def test_obstacle_kernel_in_a_batch_of_mixed_agents(eng):
    """agents with and without obstacles in one launch: only those with obstacles are deferred"""
    from oracle import oracle
    kws = [dict(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=20, lead_gap=22.0),
           dict(ref_kind="scurve", kappa=0.02, v0=8.0, grid=(7, 15, 17), seed=4),
           dict(ref_kind="arc", v0=12.0, grid=(5, 9, 33), n_obstacles=3, seed=9, draw_traj_set=True, kinematic_debug=True),
           dict(ref_kind="arc", v0=9.0, grid=(3, 7, 13), n_obstacles=1, seed=2, collision=False)]
    inps = [synthetic.make_inputs(hull_builder=hip_hulls(), **kw) for kw in kws]
    outs = [oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw)) for kw in kws]
    eng.set_obstacle_stage(2, 5)
    try:
        eng.upload(inps)
        eng.evaluate()
        res = eng.finish()
        assert eng.step_info()["obstacle_kernel"] == 1
        for a, (inp, out) in enumerate(zip(inps, outs)):
            compare(eng, inp, out, res[a], agent=a)
    finally:
        eng.set_obstacle_stage(0)


def test_obstacle_kernel_refuses_what_it_cannot_run(eng):
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), ref_kind="arc", v0=10.0, grid=(5, 9, 11), n_obstacles=4, write_bundle=False,
                                write_costmap=False)
    eng.set_obstacle_stage(2)
    try:
        with pytest.raises(ValueError, match="obstacle kernel forced"):
            eng.plan_step(inp)
        eng.set_obstacle_stage(0)
        res = eng.plan_step(inp)   # automatic: fused
        assert eng.step_info()["obstacle_kernel"] == 0 and res["n_candidates"] == inp.n_candidates
    finally:
        eng.set_obstacle_stage(0)


def test_config3_full_size_on_the_obstacle_kernel(eng):
    """BASELINE config 3 (50 388 x 31, 20 obstacles, lead vehicle) -- the automatic choice is the obstacle kernel; every
    candidate against the oracle"""
    from oracle import oracle
    kw = dict(ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0)
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    out = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw))
    res = eng.plan_step(inp)
    assert eng.step_info()["obstacle_kernel"] == 1
    compare(eng, inp, out, res)
    assert res["n_collisions"] > 0


def test_obstacle_kernel_list_of_costed_candidates(eng):
    """The obstacle kernel visits the walk's list of COSTED candidates, not the grid (production flag set: infeasible candidates
    have neither a prediction cost nor a collision check): results equal the oracle and the fused stage whatever the list's order;
    a step whose list is EMPTY (nothing feasible: non-finite ego state) ends with no winner and leaves the list empty for the
    next, ordinary step; repeated steps are bit-identical."""
    from oracle import oracle
    kw = dict(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=12, lead_gap=20.0)
    inp = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
    ref_inp = synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw)
    out = oracle.plan_step(ref_inp)
    assert 0 < out["costed"].sum() < inp.n_candidates   # a real mix of costed and infeasible candidates
    eng.set_obstacle_stage(2, 3)
    try:
        res = eng.plan_step(inp)
        assert eng.step_info()["obstacle_kernel"] == 1
        compare(eng, inp, out, res, ref_inp=ref_inp)
        c0, f0 = eng.costs()
        for _ in range(3):   # the list is rebuilt in another order every step: nothing that is computed from it may move
            eng.evaluate(); r = eng.finish()[0]
            c, f = eng.costs()
            assert np.array_equal(c, c0) and np.array_equal(f, f0) and r["best_index"] == res["best_index"] and r["n_collisions"] == res["n_collisions"]
        # nothing feasible at all: an empty list
        bad = synthetic.make_inputs(hull_builder=hip_hulls(), **kw)
        bad.x0_lon = np.array([bad.x0_lon[0], float("nan"), 0.0])
        rb = eng.plan_step(bad)
        assert rb["best_index"] == -1 and rb["n_feasible"] == 0 and rb["n_collisions"] == 0
        # ... and the next ordinary step is unaffected
        res2 = eng.plan_step(inp)
        compare(eng, inp, out, res2, ref_inp=ref_inp)
        assert {k: res2[k] for k in RESULT_KEYS} == {k: res[k] for k in RESULT_KEYS}
    finally:
        eng.set_obstacle_stage(0)
