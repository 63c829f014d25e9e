"""Golden vectors of the on-disk log formats: runs the REFERENCE's own loggers
(frenetix_motion_planner/utility/logging_helpers.py: SqlLogger, DataLoggingCosts -- imported from /root/reference
through the stub harness of ref_harness.py) on a small set of authored trajectory objects and stores what they wrote:
the schema and every row of trajectories.db, and the text of logs.csv / trajectories.csv / predictions.csv /
collision.csv.  tests/test_logging_formats.py feeds the same trajectory objects to this package's loggers and compares.

    python tests/golden/gen_logging_golden.py      ->  tests/golden/logging_golden.json
"""
import json
import os
import sqlite3
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

COST_WEIGHTS = {"distance_to_reference_path": 5.0, "lateral_jerk": 0.2, "longitudinal_jerk": 0.2, "prediction": 0.2,
                "velocity_offset": 1.0}
EXTERNAL = {"responsibility": 0.0}


class Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def authored_trajectories(n=7, S=31, seed=20241008):
    """Plain-number description of n trajectories (JSON-able); special values exercise the float formats."""
    rng = np.random.default_rng(seed)
    out = []
    for g in range(n):
        arr = {k: (rng.standard_normal(S) * sc).tolist() for k, sc in
               (("x", 100.0), ("y", 10.0), ("theta", 1.0), ("kappa", 0.01), ("theta_cl", 0.1), ("v", 10.0), ("a", 2.0),
                ("s", 200.0), ("d", 1.0))}
        arr["x"][0], arr["x"][1], arr["x"][2], arr["x"][3] = 0.0, -0.0, 1e-7, 123456.789
        arr["v"][0], arr["v"][1], arr["a"][0] = 12.0, 0.1, -1.5e-12
        arr["s"][0], arr["d"][0] = 57.30000000000001, -0.25
        names = [n_ for n_ in COST_WEIGHTS if not (g == 3 and n_ == "prediction")]
        raw = {n_: float(abs(rng.standard_normal()) * 3) for n_ in names}
        out.append(dict(
            uniqueId=int(g * 3 + 1), feasible=bool(g % 3 != 1), dt=0.1, horizon=float(np.round(1.1 + 0.3 * g, 2)),
            cost=float(sum(COST_WEIGHTS[n_] * raw[n_] for n_ in names)), planes=arr,
            costMap={n_: [raw[n_], COST_WEIGHTS[n_] * raw[n_]] for n_ in names},
            feasabilityMap={"Yaw rate Constraint": float(g % 3 == 1), "Acceleration Constraint": 0.0,
                            "Curvature Constraint": float(g == 4), "Curvature Rate Constraint": 0.0},
            sampling_parameters=[0.0, float(np.round(1.1 + 0.3 * g, 2)), 57.3, 9.5, 0.25, 3.0 + g, 0.0, -0.25, 0.1, 0.0,
                                 -3.0 + 0.75 * g, 0.0, 0.0],
            ego_risk=None if g % 2 else 0.01 * g, obst_risk=None if g % 2 else 0.002 * g,
            coll_detected=None if g == 0 else bool(g == 5), boundary_harm=None if g < 2 else (0.0 if g != 6 else 0.0123)))
    return out


def as_objects(desc):
    """Objects with the frenetix TrajectorySample attribute surface (numpy arrays, as both back-ends hold them)."""
    objs = []
    for d in desc:
        p = {k: np.array(v, dtype=np.float64) for k, v in d["planes"].items()}
        objs.append(Obj(uniqueId=d["uniqueId"], feasible=d["feasible"], dt=d["dt"], cost=d["cost"],
                        cartesian=Obj(x=p["x"], y=p["y"], theta=p["theta"], kappa=p["kappa"], v=p["v"], a=p["a"]),
                        curvilinear=Obj(s=p["s"], d=p["d"], theta=p["theta_cl"]),
                        costMap={k: tuple(v) for k, v in d["costMap"].items()}, feasabilityMap=dict(d["feasabilityMap"]),
                        sampling_parameters=np.array(d["sampling_parameters"]), _ego_risk=d["ego_risk"],
                        _obst_risk=d["obst_risk"], _coll_detected=d["coll_detected"], boundary_harm=d["boundary_harm"],
                        harm_occ_module=None))
    return objs


def predictions():
    return {7: dict(pos_list=np.array([[1.0, 2.0], [1.5, 2.25]]), orientation_list=np.array([0.1, 0.2]),
                    cov_list=np.array([[[0.1, 0.0], [0.0, 0.1]]] * 2), shape=dict(length=4.5, width=1.9))}


def drive(logger, trajs, hist, save_all):
    """The calls planner.py:637-649 makes, on either implementation."""
    ego = Obj(initial_state=Obj(position=np.array([-10.071488, 0.40359501])))
    logger.sql_logger.write_reference_path(np.array([[0.0, 0.0], [1.0, 0.5], [2.0, 1.25]]))
    logger.log(trajs[0], time_step=0, infeasible_kinematics=hist, percentage_kinematics=61.42857142857143, planning_time=0.0123,
               ego_vehicle=ego, desired_velocity=12.0, replanning_counter=0)
    logger.log_predicition(predictions())
    if save_all:
        logger.log_all_trajectories(trajs, 0)
    logger.log(trajs[0], time_step=1, infeasible_kinematics=hist, percentage_kinematics=61.42857142857143, planning_time=0.0,
               ego_vehicle=ego, desired_velocity=12.0, replanning_counter=1)
    logger.log(trajs[2], time_step=3, infeasible_kinematics=hist, percentage_kinematics=None, planning_time=0.5,
               ego_vehicle=ego, desired_velocity=11.5, replanning_counter=0)
    logger.log(None, time_step=4, infeasible_kinematics=hist, percentage_kinematics=0.0, planning_time=0.25,
               ego_vehicle=ego, desired_velocity=11.5)
    if save_all:
        logger.log_all_trajectories(trajs[1:4], 3)
    logger.log_collision(True, 4.508, 1.61, 0.75, center=[1.0, 2.0], last_center=[0.5, 1.5], r_x=2.25, r_y=0.8, orientation=0.3)


def dump(path_logs):
    """Everything the loggers wrote, as JSON-able data."""
    con = sqlite3.connect(os.path.join(path_logs, "trajectories.db"))
    schema = [list(r) for r in con.execute("SELECT type, name, sql FROM sqlite_master ORDER BY name")]
    tables = {}
    for (name,) in con.execute("SELECT name FROM sqlite_master WHERE type = 'table' ORDER BY name"):
        rows = [list(r) for r in con.execute(f"SELECT * FROM {name} ORDER BY 1, 2")]
        tables[name] = [[v.hex() if isinstance(v, bytes) else v for v in r] for r in rows]
    con.close()
    files = {}
    for f in ("logs.csv", "trajectories.csv", "predictions.csv", "collision.csv"):
        p = os.path.join(path_logs, f)
        files[f] = open(p).read() if os.path.exists(p) else None
    return dict(schema=schema, tables=tables, files=files)


def main():
    import ref_harness
    ref_harness.install()
    import frenetix_motion_planner.utility.logging_helpers as lh

    desc = authored_trajectories()
    hist = [3, 0, 1, 0, 0, 2, 1, 0, 0, 0, 0]
    out = dict(trajectories=desc, hist=hist, cases={})
    for save_all in (True, False):
        with tempfile.TemporaryDirectory() as tmp:
            cfg_plan = Obj(debug=Obj(save_unweighted_costs=False, save_all_traj=save_all),
                           cost=Obj(external_cost_weights=dict(EXTERNAL)), planning=Obj(dt=0.1, planning_horizon=3.0))
            cfg_sim = Obj(simulation=Obj(ego_agent_id=60000), vehicle=Obj(length=4.508, width=1.61))
            logger = lh.DataLoggingCosts(path_logs=tmp, config_plan=cfg_plan, config_sim=cfg_sim, scenario=None,
                                         planning_problem=Obj(), save_all_traj=save_all, cost_params=dict(COST_WEIGHTS))
            drive(logger, as_objects(desc), hist, save_all)
            logger.sql_logger.con.commit()
            logger.sql_logger.con.close()
            out["cases"]["save_all" if save_all else "optimal_only"] = dump(tmp)
    path = os.path.join(HERE, "logging_golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
