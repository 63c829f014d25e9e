"""On-disk log formats against goldens written by the reference's own loggers
(tests/golden/gen_logging_golden.py -> logging_golden.json): same SQLite schema and rows, same CSV text."""
import importlib.util
import json
import os

import numpy as np
import pytest

from frenetix_motion_planner_amd import logging_formats as lf

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("gen_logging_golden", os.path.join(HERE, "golden", "gen_logging_golden.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)  # authored inputs + the driver + dump(); its main() (the reference import) is not run


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "logging_golden.json")))


def _norm_sql(s):
    return None if s is None else " ".join(s.replace("\n", " ").split()).replace("( ", "(").replace(" )", ")").replace(", ", ",")


@pytest.mark.parametrize("case", ["save_all", "optimal_only"])
def test_same_files_as_the_reference_logger(golden, case, tmp_path):
    save_all = case == "save_all"
    assert golden["trajectories"] == json.loads(json.dumps(gen.authored_trajectories()))  # inputs are reproducible
    cfg_plan = gen.Obj(debug=gen.Obj(save_unweighted_costs=False, save_all_traj=save_all),
                       cost=gen.Obj(external_cost_weights=dict(gen.EXTERNAL)), planning=gen.Obj(dt=0.1, planning_horizon=3.0))
    cfg_sim = gen.Obj(simulation=gen.Obj(ego_agent_id=60000), vehicle=gen.Obj(length=4.508, width=1.61))
    logger = lf.DataLoggingCosts(str(tmp_path), cfg_plan, cfg_sim, save_all_traj=save_all, cost_params=dict(gen.COST_WEIGHTS),
                                 external_cost_weights=dict(gen.EXTERNAL))
    gen.drive(logger, gen.as_objects(golden["trajectories"]), golden["hist"], save_all)
    logger.close()
    got = gen.dump(str(tmp_path))
    want = golden["cases"][case]
    for f, text in want["files"].items():
        assert got["files"][f] == text, f
    assert sorted(got["tables"]) == sorted(want["tables"])
    for name, rows in want["tables"].items():
        assert got["tables"][name] == rows, name
    assert [(t, n, _norm_sql(s)) for t, n, s in got["schema"]] == [(t, n, _norm_sql(s)) for t, n, s in want["schema"]]


def test_header_only_and_unweighted(tmp_path):
    h = lf.DataLoggingCosts(str(tmp_path / "x"), header_only=True)
    assert h.header is None and not (tmp_path / "x").exists()
    lg = lf.DataLoggingCosts(str(tmp_path / "y"), save_all_traj=True, cost_params={"b": 1.0, "a": 2.0},
                             save_unweighted_costs=True)
    assert lg.get_headers().endswith("costs_cumulative_weighted;a_cost;b_cost")
    t = gen.as_objects(gen.authored_trajectories(1))[0]
    t.costMap = {"a": (3.0, 6.0)}
    line = lg.log_costs_of_single_trajectory(t, "", ["a"])
    assert line == f';"{t.cost}";"3.0";"0"'
    lg.close()


@pytest.mark.gpu
def test_bulk_path_equals_per_trajectory_path(tmp_path):
    """A real plan step: logging `all_traj` plane-wise (9 plane copies) writes exactly what the per-trajectory path
    writes from TrajectorySample views."""
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    from frenetix_motion_planner_amd.reactive_planner import _LazySortedList
    from frenetix_motion_planner_amd.trajectories import PlanStepResult
    inp = synthetic.make_inputs(ref_kind="scurve", v0=9.0, level=1, n_obstacles=3, draw_traj_set=True, kinematic_debug=True,
                                hull_builder=build_obstacle_hulls)
    with FrenetEngine(max_candidates=inp.n_candidates) as eng:
        res = eng.plan_step(inp)
        step = PlanStepResult(eng, inp, res)
        lazy = _LazySortedList(step)
        assert len(lazy) == res["n_candidates"] > 50
        outs = []
        for k, trajs in enumerate((lazy, list(lazy))):
            d = tmp_path / str(k)
            lg = lf.DataLoggingCosts(str(d), save_all_traj=True, cost_params=dict(inp.cost_weights))
            lg.log_all_trajectories(trajs, 5)
            lg.log(step.best, 5, [0] * 11, 50.0, 0.01, [type("S", (), {"position": np.array([1.0, 2.0])})()], desired_velocity=12.0)
            lg.close()
            outs.append(gen.dump(str(d)))
        assert outs[0]["files"]["trajectories.csv"].count("\n") == len(lazy)
        assert outs[0] == outs[1]
