"""On-disk formats of the planner's logs, fed from the structure-of-arrays TrajectoryBundle.

Same files, tables, columns and text formats as frenetix_motion_planner/utility/logging_helpers.py:

    SqlLogger         trajectories.db (:22-297): tables trajectories / trajectories_meta / sampling_params / meta /
                      infeasability / costs; float arrays as JSON text with '{:.5g}' elements
    DataLoggingCosts  logs.csv, predictions.csv, collision.csv, trajectories.csv (:300-663), ';'-separated, values
                      wrapped by json.dumps(str(.)), trajectory arrays '{:.5e}' in trajectories.csv

A reader of the reference's logs (its visualisation and evaluation scripts) reads these unchanged.  What differs is
where the numbers come from: `log_all_trajectories` recognises the planner's lazy `all_traj` list and pulls the nine
logged planes of the whole step with nine plane copies ([S, C] each, `fx_read_plane_agent`) instead of one gather per
trajectory, then formats rows from those arrays.  Any iterable of objects with the TrajectorySample attributes works
too (that path is what the golden test compares with the reference's own logger).

The `meta` table's `scenario` blob is whatever the caller hands over (the reference stores the CommonRoad protobuf
written by commonroad-io, which is not available here); configs are stored as JSON like the reference does.
"""
import json
import math
import os
import sqlite3
from pathlib import Path
from typing import Iterable, List, Optional

import numpy as np

from . import _abi

_INF_NAMES = ["Yaw_rate", "Acceleration", "Curvature", "Curvature_Rate"]  # logging_helpers.py:161-166
_SQL_PLANES = ("x", "y", "theta", "kappa", "theta_cl", "v", "a", "s", "d")  # column order of `trajectories`
_REASON_BIT = {"Curvature": 5, "Yaw_rate": 6, "Curvature_Rate": 7, "Acceleration": 8}


def _g5(values) -> str:
    return "[" + ",".join("{:.5g}".format(x) for x in values) + "]"


def _e5(values) -> str:
    for x in values:
        assert math.isfinite(x)
    return json.dumps(",".join("{:.5e}".format(x) for x in values))


def _q(x) -> str:
    """json.dumps(str(x), default=default) of logging_helpers.py"""
    return json.dumps(str(x))


def _config_dict(config) -> dict:
    """SqlLogger._convert_dict_config (:31-49): two levels of plain attributes, private ones skipped."""
    if config is None:
        return {}
    if isinstance(config, dict):
        return config
    out = {}
    for name, section in vars(config).items():
        if name.startswith("_"):
            continue
        out[name] = dict(vars(section)) if hasattr(section, "__dict__") else section
    return out


class _BulkStep:
    """The nine logged planes, costs, flags and sampling parameters of a whole plan step, read plane-wise."""

    def __init__(self, step, ids: np.ndarray):
        eng, agent = step.engine, step.agent
        step._check()
        self.ids = ids
        idx = {"x": 0, "y": 1, "theta": 2, "v": 3, "a": 4, "kappa": 5, "s": 7, "d": 8, "theta_cl": 9}
        # [S, C] per plane -> rows of the listed candidates [len(ids), S]
        self.planes = {n: np.ascontiguousarray(eng.plane(p, agent)[:, ids].T) for n, p in idx.items()}
        self.cost = step.cost[ids]
        self.flags = step.flags[ids]
        inp = step.inputs
        self.names = list(inp.cost_names)
        cm = eng.costmap(agent)[ids] if self.names else np.zeros((len(ids), 0))
        self.raw = cm
        self.weights = np.array([inp.cost_weights[n] for n in self.names])
        self.params = np.array([inp.candidate_params(int(g) + inp.shard_begin) for g in ids]).reshape(len(ids), 13)
        self.dt = inp.dt
        self.samples = [step.sample(int(g)) for g in ids]  # cheap views; risk / harm fields live there


class SqlLogger:
    def __init__(self, path_logs, config_plan=None, config_sim=None, scenario_blob: bytes = b""):
        self.path_logs = Path(path_logs)
        self.path_logs.mkdir(parents=True, exist_ok=True)
        self._db_path = self.path_logs / "trajectories.db"
        self._db_path.unlink(missing_ok=True)
        self.con = sqlite3.connect(self._db_path, isolation_level="EXCLUSIVE")
        self.con.executescript("PRAGMA journal_mode = OFF; PRAGMA locking_mode = EXCLUSIVE; PRAGMA temp_store = MEMORY;")
        arrays = ", ".join(f"{c} TEXT NOT NULL" for c in ("x", "y", "theta", "kappa", "curvilinear_theta", "v", "a",
                                                           "trajectory_long", "trajectory_lat"))
        self.con.execute(f"CREATE TABLE trajectories(time_step INT NOT NULL, id INT NOT NULL, {arrays}, "
                         "PRIMARY KEY(time_step, id)) STRICT")
        self.con.execute("CREATE TABLE trajectories_meta(time_step INT NOT NULL, id INT NOT NULL, dt REAL NOT NULL, "
                         "s_position REAL NOT NULL, d_position REAL NOT NULL, ego_risk REAL, obst_risk REAL, "
                         "collision_detected INT, boundary_harm REAL, horizon REAL NOT NULL, PRIMARY KEY(time_step, id)) STRICT")
        params = ", ".join(f"{c} REAL NOT NULL" for c in ("t0", "t1", "s0", "ss0", "sss0", "ss1", "sss1", "d0", "dd0", "ddd0",
                                                          "d1", "dd1", "ddd1"))
        self.con.execute(f"CREATE TABLE sampling_params(time_step INT NOT NULL, id INT NOT NULL, {params}, "
                         "PRIMARY KEY(time_step, id)) STRICT")
        self.con.execute("CREATE TABLE meta(key TEXT PRIMARY KEY, value ANY) STRICT")
        self.con.execute("INSERT INTO meta VALUES(?, ?)", ("scenario", scenario_blob))
        self.con.commit()
        self.con.execute("INSERT INTO meta VALUES(?, json(?))", ("config_plan", json.dumps(_config_dict(config_plan), skipkeys=True)))
        self.con.execute("INSERT INTO meta VALUES(?, json(?))", ("config_sim", json.dumps(_config_dict(config_sim), skipkeys=True)))
        self.con.commit()
        self.cost_names: List[str] = []
        self.set_inf_names(list(_INF_NAMES))

    def write_reference_path(self, reference_path) -> None:
        rp = {"x": np.asarray(reference_path)[:, 0].tolist(), "y": np.asarray(reference_path)[:, 1].tolist()}
        self.con.execute("INSERT INTO meta VALUES(?, json(?))", ("reference_path", json.dumps(rp)))

    def set_inf_names(self, inf_names_list: List[str]) -> None:
        self.inf_names = inf_names_list
        cols = "".join(f"inf_{n.lower()} INT NOT NULL, " for n in inf_names_list)
        self.con.execute(f"CREATE TABLE infeasability(time_step INT NOT NULL, id INT NOT NULL, feasible INT NOT NULL, {cols}"
                         "PRIMARY KEY(time_step, id)) STRICT")

    def set_cost_names(self, cost_names_list: List[str]) -> None:
        self.cost_names = list(cost_names_list)
        cols = "".join(f"{n} REAL NOT NULL, " for n in self.cost_names)
        self.con.execute(f"CREATE TABLE costs(time_step INT NOT NULL, id INT NOT NULL, costs_cumulative_weighted REAL NOT NULL, "
                         f"{cols}PRIMARY KEY(time_step, id)) STRICT")

    # -- rows (:198-258) --
    @staticmethod
    def _trajectories_row(time_step: int, t):
        c, k = t.cartesian, t.curvilinear
        return (time_step, str(t.uniqueId), _g5(c.x), _g5(c.y), _g5(c.theta), _g5(c.kappa), _g5(k.theta), _g5(c.v), _g5(c.a),
                _g5(k.s), _g5(k.d))

    @staticmethod
    def _trajectories_meta_row(time_step: int, t):
        return (time_step, t.uniqueId, t.dt, t.curvilinear.s[0], t.curvilinear.d[0], t._ego_risk, t._obst_risk, t._coll_detected,
                t.boundary_harm, t.sampling_parameters[1])

    @staticmethod
    def _sampling_params_row(time_step: int, t):
        return [time_step, t.uniqueId] + list(t.sampling_parameters)

    def _costs_row(self, time_step: int, t):
        row = [time_step, t.uniqueId, t.cost]
        for n in self.cost_names:
            row.append(t.costMap[n][1] if n in t.costMap else 0.0)
        return row

    def _infeasability_row(self, time_step: int, t):
        row = [time_step, t.uniqueId, t.feasible]
        for n in self.inf_names:
            row.append(t.feasabilityMap[n.replace("_", " ") + " Constraint"])
        return row

    def _insert(self, traj, meta, samp, cost, inf):
        self.con.executemany("INSERT INTO trajectories VALUES(?, ?, json(?), json(?), json(?), json(?), json(?), json(?), "
                             "json(?), json(?), json(?))", traj)
        self.con.executemany(f"INSERT INTO trajectories_meta VALUES(?, ?, {','.join(8 * '?')})", meta)
        self.con.executemany(f"INSERT INTO sampling_params VALUES(?, ?, {','.join(13 * '?')})", samp)
        self.con.executemany(f"INSERT INTO costs VALUES(?, ?, ?, {','.join(len(self.cost_names) * '?')})", cost)
        self.con.executemany(f"INSERT INTO infeasability VALUES(?, ?, ?, {','.join(len(self.inf_names) * '?')})", inf)
        self.con.commit()

    def log_all_trajectories(self, all_trajectories, time_step: int):
        bulk = _bulk_of(all_trajectories)
        if bulk is not None:
            return self._log_bulk(bulk, time_step)
        rows = [[], [], [], [], []]
        for t in all_trajectories:
            rows[0].append(self._trajectories_row(time_step, t))
            rows[1].append(self._trajectories_meta_row(time_step, t))
            rows[2].append(self._sampling_params_row(time_step, t))
            rows[3].append(self._costs_row(time_step, t))
            rows[4].append(self._infeasability_row(time_step, t))
        self._insert(*rows)

    def _log_bulk(self, b: _BulkStep, time_step: int):
        P = b.planes
        traj, meta, samp, cost, inf = [], [], [], [], []
        col = {n: k for k, n in enumerate(b.names)}
        for j, g in enumerate(b.ids):
            g = int(g)
            t = b.samples[j]
            traj.append((time_step, str(g)) + tuple(_g5(P[n][j]) for n in _SQL_PLANES))
            meta.append((time_step, g, b.dt, float(P["s"][j, 0]), float(P["d"][j, 0]), t._ego_risk, t._obst_risk, t._coll_detected,
                         t.boundary_harm, float(b.params[j, 1])))
            samp.append([time_step, g] + [float(x) for x in b.params[j]])
            cost.append([time_step, g, float(b.cost[j])] +
                        [float(b.weights[col[n]] * b.raw[j, col[n]]) if n in col else 0.0 for n in self.cost_names])
            reasons = (int(b.flags[j]) >> _abi.FX_REASON_SHIFT) & 0x7FF
            inf.append([time_step, g, bool(int(b.flags[j]) & _abi.FX_FLAG_FEASIBLE)] +
                       [float((reasons >> _REASON_BIT[n]) & 1) for n in self.inf_names])
        self._insert(traj, meta, samp, cost, inf)

    def close(self):
        self.con.commit()
        self.con.close()


def _bulk_of(all_trajectories) -> Optional[_BulkStep]:
    step = getattr(all_trajectories, "_step", None)
    if step is None or not hasattr(all_trajectories, "_order") or step._stale:
        return None
    return _BulkStep(step, np.asarray(all_trajectories._order(), dtype=np.int64))


class DataLoggingCosts:
    LOG_HEAD = ("trajectory_number;calculation_time_s;x_position_vehicle_m;y_position_vehicle_m;optimal_trajectory;"
                "percentage_feasible_traj;infeasible_sum;inf_kin_acceleration;inf_kin_negative_s_velocity;inf_kin_max_s_idx;"
                "inf_kin_negative_v_velocity;inf_kin_max_curvature;inf_kin_yaw_rate;inf_kin_max_curvature_rate;"
                "inf_kin_vehicle_acc;inf_cartesian_transform;inf_precision_error;x_positions_m;y_positions_m;"
                "theta_orientations_rad;kappa_rad;curvilinear_orientations_rad;velocities_mps;desired_velocity_mps;"
                "accelerations_mps2;trajectory_long;trajectory_lat;s_position_m;d_position_m;ego_risk;obst_risk;"
                "accepted_occ_harm;costs_cumulative_weighted;")
    TRAJ_HEAD = ("time_step;trajectory_number;unique_id;feasible;horizon;dt;x_positions_m;y_positions_m;"
                 "theta_orientations_rad;kappa_rad;curvilinear_orientations_rad;velocities_mps;accelerations_mps2;"
                 "s_position_m;d_position_m;ego_risk;obst_risk;costs_cumulative_weighted;")

    def __init__(self, path_logs: str, config_plan=None, config_sim=None, scenario_blob: bytes = b"", header_only: bool = False,
                 save_all_traj: bool = False, cost_params: Optional[dict] = None, external_cost_weights: Optional[dict] = None,
                 save_unweighted_costs: bool = False):
        self.save_all_traj = save_all_traj
        self.header = self.trajectories_header = self.prediction_header = self.collision_header = None
        self.save_unweighted_costs = save_unweighted_costs
        self.path_logs = str(path_logs)
        self._cost_list_length = None
        self.cost_names_list = None
        self.trajectories_file_name = "trajectories.csv"
        if header_only:
            return
        self.trajectory_number = 0
        self._trajectories_log_path = None
        os.makedirs(self.path_logs, exist_ok=True)
        self._log_path = os.path.join(self.path_logs, "logs.csv")
        self._prediction_log_path = os.path.join(self.path_logs, "predictions.csv")
        self._collision_log_path = os.path.join(self.path_logs, "collision.csv")
        self.sql_logger = SqlLogger(Path(self.path_logs), config_plan, config_sim, scenario_blob)
        self.set_logging_header(cost_params, external_cost_weights)

    def set_logging_header(self, cost_function_names=None, external_cost_weights=None):
        cost_names = ""
        if cost_function_names:  # :344-349
            self.cost_names_list = sorted(set(cost_function_names.keys()) | set((external_cost_weights or {}).keys()))
            self._cost_list_length = len(self.cost_names_list)
            for n in self.cost_names_list:
                cost_names += n.replace(" ", "_") + "_cost;"
        self.sql_logger.set_cost_names(self.cost_names_list or [])
        self.header = self.LOG_HEAD + cost_names.strip(";")
        self.trajectories_header = (self.TRAJ_HEAD + cost_names + "inf_kin_yaw_rate;inf_kin_acceleration;inf_kin_max_curvature;"
                                    "inf_kin_max_curvature_rate;").strip(";")
        self.prediction_header = "trajectory_number;prediction"
        with open(self._log_path, "w+") as fh:
            fh.write(self.header)
        with open(self._prediction_log_path, "w+") as fh:
            fh.write(self.prediction_header)
        if self.save_all_traj:
            self._trajectories_log_path = os.path.join(self.path_logs, self.trajectories_file_name)
            with open(self._trajectories_log_path, "w+") as fh:
                fh.write(self.trajectories_header)

    def get_headers(self):
        return self.header

    # -- logs.csv (:425-515) --
    def log(self, trajectory, time_step: int, infeasible_kinematics, percentage_kinematics, planning_time: float,
            ego_vehicle, collision: bool = False, desired_velocity: float = None, replanning_counter: int = 0):
        line = "\n" + str(time_step)
        if trajectory is not None:
            c, k = trajectory.cartesian, trajectory.curvilinear
            r = replanning_counter
            pos = _ego_position(ego_vehicle)
            line += ";" + _q(planning_time) + ";" + _q(pos[0]) + ";" + _q(pos[1]) + ";True"
            line += (";" + _q(percentage_kinematics)) if percentage_kinematics is not None else ";"
            for kin in infeasible_kinematics:
                line += ";" + _q(kin)
            for arr in (c.x[r:], c.y[r:], c.theta[r:], c.kappa[r:], k.theta[r:], c.v[r:]):
                line += ";" + _q(",".join(map(str, arr)))
            line += ";" + _q(desired_velocity)
            line += ";" + _q(",".join(map(str, c.a[r:])))
            line += ";" + _q(",".join(map(str, k.s))) + ";" + _q(",".join(map(str, k.d)))
            line += ";" + _q(k.s[r]) + ";" + _q(k.d[r])
            if trajectory._ego_risk is not None and trajectory._obst_risk is not None:
                line += ";" + _q(trajectory._ego_risk) + ";" + _q(trajectory._obst_risk)
            else:
                line += ";;"
            harm = getattr(trajectory, "harm_occ_module", None)
            line += (";" + _q(harm)) if harm is not None else ";"
            line = self.log_costs_of_single_trajectory(trajectory, line, list(trajectory.costMap.keys()))
        else:
            line += ";" + _q(planning_time) + ";False"
            for kin in infeasible_kinematics:
                line += ";" + _q(kin)
            line += ";None" * 8
            for _ in range(self._cost_list_length or 0):
                line += ";None"
        with open(self._log_path, "a") as fh:
            fh.write(line)

    def log_predicition(self, prediction):
        line = "\n" + str(self.trajectory_number) + ";" + json.dumps(prediction, default=_json_default)
        with open(self._prediction_log_path, "a") as fh:
            fh.write(line)

    def log_collision(self, collision_with_obj, ego_length, ego_width, progress, center=None, last_center=None, r_x=None, r_y=None,
                      orientation=None):
        self.collision_header = "ego_length;ego_width;progress;center_x;center_y;last_center_x;last_center_y;r_x;r_y;orientation"
        with open(self._collision_log_path, "w+") as fh:
            fh.write(self.collision_header)
        line = "\n" + str(ego_length) + ";" + str(ego_width) + ";" + str(progress)
        if collision_with_obj:
            line += "".join(";" + str(v) for v in (center[0], center[1], last_center[0], last_center[1], r_x, r_y, orientation))
        else:
            line += ";None" * 7
        with open(self._collision_log_path, "a") as fh:
            fh.write(line)

    # -- trajectories.csv (:566-640) --
    def log_all_trajectories(self, all_trajectories, time_step: int):
        bulk = _bulk_of(all_trajectories)
        if bulk is not None:
            lines = [self._trajectory_line_bulk(bulk, j, time_step) for j in range(len(bulk.ids))]
            with open(self._trajectories_log_path, "a") as fh:
                fh.write("".join(lines))
            self.sql_logger._log_bulk(bulk, time_step)
            return
        for i, t in enumerate(all_trajectories):
            self.log_trajectory(t, i, time_step, t.feasible)
        self.sql_logger.log_all_trajectories(all_trajectories, time_step)

    def _trajectory_line(self, t, number: int, time_step, feasible) -> str:
        line = "\n" + str(time_step) + ";" + str(number) + ";" + str(t.uniqueId) + ";" + str(feasible)
        line += ";" + str(round(t.horizon, 3) if hasattr(t, "horizon") else round(t.sampling_parameters[1], 3))
        line += ";" + str(t.dt)
        c, k = t.cartesian, t.curvilinear
        for arr in (c.x, c.y, c.theta, c.kappa, k.theta, c.v, c.a):
            line += ";" + _e5(arr)
        line += ";" + _q(k.s[0]) + ";" + _q(k.d[0])
        if t._ego_risk is not None and t._obst_risk is not None:
            line += ";" + _q(t._ego_risk) + ";" + _q(t._obst_risk)
        else:
            line += ";;"
        line = self.log_costs_of_single_trajectory(t, line, list(t.costMap.keys()))
        for key in ("Yaw rate Constraint", "Acceleration Constraint", "Curvature Constraint", "Curvature Rate Constraint"):
            line += ";" + str(t.feasabilityMap[key])
        return line

    def _trajectory_line_bulk(self, b: _BulkStep, j: int, time_step) -> str:
        t = b.samples[j]
        P = b.planes
        line = "\n" + str(time_step) + ";" + str(j) + ";" + str(int(b.ids[j])) + ";" + str(t.feasible)
        line += ";" + str(round(t.horizon, 3)) + ";" + str(b.dt)
        for n in ("x", "y", "theta", "kappa", "theta_cl", "v", "a"):
            line += ";" + _e5(P[n][j])
        line += ";" + _q(P["s"][j, 0]) + ";" + _q(P["d"][j, 0])
        if t._ego_risk is not None and t._obst_risk is not None:
            line += ";" + _q(t._ego_risk) + ";" + _q(t._obst_risk)
        else:
            line += ";;"
        line += ";" + _q(float(b.cost[j]))
        col = {n: k for k, n in enumerate(b.names)}
        for n in self.cost_names_list or []:
            if n in col:
                raw = float(b.raw[j, col[n]])
                line += ";" + _q(raw if self.save_unweighted_costs else float(b.weights[col[n]] * raw))
            else:
                line += ";" + _q(0)
        reasons = (int(b.flags[j]) >> _abi.FX_REASON_SHIFT) & 0x7FF
        for n in ("Yaw_rate", "Acceleration", "Curvature", "Curvature_Rate"):
            line += ";" + str(float((reasons >> _REASON_BIT[n]) & 1))
        return line

    def log_trajectory(self, trajectory, trajectory_number: int, time_step, feasible: bool):
        with open(self._trajectories_log_path, "a") as fh:
            fh.write(self._trajectory_line(trajectory, trajectory_number, time_step, feasible))

    def log_costs_of_single_trajectory(self, trajectory, line: str, cost_list_names) -> str:
        line += ";" + _q(trajectory.cost)
        for n in self.cost_names_list or []:
            if n in cost_list_names:
                line += ";" + _q(trajectory.costMap[n][0 if self.save_unweighted_costs else 1])
            else:
                line += ";" + _q(0)
        return line

    def close(self):
        self.sql_logger.close()


def _ego_position(ego_vehicle):
    """`ego_vehicle.initial_state.position` (a CommonRoad DynamicObstacle in the reference); here also a list of planner
    states (ReactivePlannerHip.ego_vehicle_history entries) or a plain position."""
    if ego_vehicle is None:
        return (None, None)
    if hasattr(ego_vehicle, "initial_state"):
        return ego_vehicle.initial_state.position
    if isinstance(ego_vehicle, (list, tuple)) and len(ego_vehicle) and hasattr(ego_vehicle[0], "position"):
        return ego_vehicle[0].position
    return ego_vehicle


def _json_default(obj):
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, np.integer):
        return int(obj)
    raise TypeError("Not serializable (type: " + str(type(obj)) + ")")
