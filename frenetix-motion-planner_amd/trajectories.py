"""Result types of a plan step -- the attribute surface of the reference's trajectory objects, backed by the
engine's structure-of-arrays TrajectoryBundle in HBM.

Reference types mirrored (read/written by planner.py:371-437, reactive_planner_cpp.py:355-357,456,470-482,
logging_helpers.py, visualization.py):
  frenetix_motion_planner/trajectories.py:56-197   CartesianSample   x, y, theta, v, a, kappa, kappa_dot
  frenetix_motion_planner/trajectories.py:200-334  CurviLinearSample s, d, theta, s_dot, s_ddot, d_dot, d_ddot
  frenetix_motion_planner/trajectories.py:337-477  TrajectorySample  + the frenetix.TrajectorySample extras
  frenetix_motion_planner/trajectories.py:480-602  TrajectoryBundle

A TrajectorySample here is a *view*: the flag word and total cost come from the arrays the engine always
reads back; the 14 planes, the per-name cost map and the polynomial coefficients of a candidate are gathered
from device memory on first access and then cached, so the object stays valid after the next plan step
overwrites the device buffers (the reference keeps `optimal_trajectory` alive across steps,
reactive_planner_cpp.py:430,437 / frenet_interface.py:277).
"""
from typing import List, Optional

import numpy as np

from . import _abi

_FEAS_KEYS = {  # frenetix feasabilityMap keys (reactive_planner_cpp.py:470-482) -> reason bit
    "Curvature Constraint": 5, "Yaw rate Constraint": 6, "Curvature Rate Constraint": 7, "Acceleration Constraint": 8,
}


class _Sample:
    def __init__(self, current_time_step: int):
        self.current_time_step = current_time_step

    def length(self) -> int:
        return self.current_time_step


class CartesianSample(_Sample):
    def __init__(self, x, y, theta, v, a, kappa, kappa_dot, current_time_step):
        super().__init__(current_time_step)
        self.x, self.y, self.theta, self.v, self.a, self.kappa, self.kappa_dot = x, y, theta, v, a, kappa, kappa_dot

    def length(self) -> int:
        return len(self.x)


class CurviLinearSample(_Sample):
    def __init__(self, s, d, theta, dd=None, ddd=None, ss=None, sss=None, current_time_step=0):
        super().__init__(current_time_step)
        self.s, self.d, self.theta = s, d, theta
        self.d_dot, self.d_ddot, self.s_dot, self.s_ddot = dd, ddd, ss, sss

    def length(self) -> int:
        return len(self.s)


class PolynomialView:
    """coeffs / delta_tau of polynomial_trajectory.py:17-272 (read-only)."""

    def __init__(self, coeffs: np.ndarray, delta_tau: float):
        self.coeffs = coeffs
        self.delta_tau = delta_tau
        self.tau_0 = 0

    def squared_jerk_integral(self, t):
        c = self.coeffs
        t2 = t * t
        t3 = t2 * t
        t4 = t3 * t
        t5 = t4 * t
        return (36 * c[3] * c[3] * t + 144 * c[3] * c[4] * t2 + 240 * c[3] * c[5] * t3 + 192 * c[4] * c[4] * t3 +
                720 * c[4] * c[5] * t4 + 720 * c[5] * c[5] * t5)


class TrajectorySample:
    # what a fresh sample has not got yet lives on the class: the reference's C++ adapter asks for EVERY evaluated trajectory as
    # an object each plan step (reactive_planner_cpp.py:353-358), so a sample is a handful of instance attributes
    _planes = _costmap = _coeffs = _pkg = _sp = None
    _materialised = False
    # writable from Python (planner.py:325-326,381-382)
    _ego_risk = _obst_risk = _boundary_harm = None
    harm_occ_module = None

    @classmethod
    def bulk(cls, step: "PlanStepResult", ids) -> list:
        """the samples of `ids` (indices within the shard) from the step's cost / flag arrays, without per-sample array look-ups"""
        ids = [int(g) for g in ids]
        base = step.inputs.shard_begin
        fl, co = step.flags[ids].tolist(), step.cost[ids].tolist()
        FEAS, VALID, COLL, SEL = _abi.FX_FLAG_FEASIBLE, _abi.FX_FLAG_VALID, _abi.FX_FLAG_COLLISION, _abi.FX_FLAG_SELECTABLE
        pkg_index = step.package.index if step.package is not None else None
        out = []
        new = object.__new__
        for g, f, c in zip(ids, fl, co):
            if pkg_index is not None and g + base == pkg_index:   # the packaged winner takes the long way (it carries its arrays)
                out.append(cls(step, g))
                continue
            t = new(cls)
            t._step, t.uniqueId, t.global_id, t._flags, t._cost = step, g, g + base, f, c
            t.feasible, t.valid = bool(f & FEAS), bool(f & VALID)
            t._coll_detected = bool(f & COLL) if (f & SEL) else None
            out.append(t)
        return out

    @property
    def dt(self) -> float:
        return self._step.inputs.dt

    @property
    def horizon(self) -> float:
        return self._step.inputs.N * self._step.inputs.dt

    def __init__(self, step: "PlanStepResult", index: int):
        self._step = step
        self.uniqueId = int(index)           # index within the evaluated shard (== creation order when nothing is sharded)
        self.global_id = int(index) + step.inputs.shard_begin   # creation order in the whole grid (reactive_planner.py:172)
        pkg = step.package
        if pkg is not None and pkg.index == int(index) + step.inputs.shard_begin:
            # the winner: the library has already delivered everything (fx_read_package), nothing is fetched
            flags, self._cost = pkg.flags, pkg.cost
            self._planes = pkg.planes
            self._pkg = pkg   # (coefficients and the cost map are built from the package when they are asked for)
        elif step.have_arrays:
            flags = int(step.flags[index])
            self._cost = float(step.cost[index])
        else:
            # the step's cost / flag arrays have not been read back: fetch this candidate whole (one synchronisation)
            rec = step.fetch_candidate(index)
            flags = int(rec["flags"])
            self._cost = float(rec["cost"])
            if rec["planes"] is not None:
                self._planes = rec["planes"]
                self._coeffs = (rec["lon"], rec["lat"], rec["traj_len"], rec["tau_lat"])
            if rec["raw_costs"] is not None:
                names, w = step.inputs.cost_names, step.inputs.cost_weights
                self._costmap = {n: (float(rec["raw_costs"][k]), float(w[n] * rec["raw_costs"][k])) for k, n in enumerate(names)}
        self._flags = flags
        self.feasible = bool(flags & _abi.FX_FLAG_FEASIBLE)
        self.valid = bool(flags & _abi.FX_FLAG_VALID)
        self._coll_detected = bool(flags & _abi.FX_FLAG_COLLISION) if (flags & _abi.FX_FLAG_SELECTABLE) else None

    # ---- cheap attributes ----
    @property
    def cost(self) -> float:
        return self._cost

    @cost.setter
    def cost(self, value):   # (an occlusion module's calc_costs adds to it before the list is sorted, trajectories.py:557-560)
        self._cost = float(value)

    @property
    def leaves_road(self) -> Optional[bool]:
        """True/False for walked candidates when the step ran the road-boundary stage, else None"""
        bound = self._step.inputs._bound   # (FX_MODE_ROAD_BOUNDARY <=> a packed boundary with pieces: PlanInputs.mode)
        if bound is None or bound["n"] <= 0 or not (self._flags & _abi.FX_FLAG_SELECTABLE):
            return None
        return bool(self._flags & _abi.FX_FLAG_BOUNDARY)

    @property
    def boundary_harm(self):
        """planner.py:369-381: logistic regression of the velocity at the first step outside the road, 0 inside."""
        if self._boundary_harm is not None or self.leaves_road is None:
            return self._boundary_harm
        if not self.leaves_road:
            return 0
        i = int(self._step.boundary_steps[self.uniqueId])
        v = float(self.cartesian.v[i])
        c = self._step.harm_coeff
        return float(1.0 / (1.0 + np.exp(-c[0] - c[1] * v)))

    @boundary_harm.setter
    def boundary_harm(self, value):
        self._boundary_harm = value

    @property
    def reasons(self) -> int:
        """bit r set <=> infeasibility reason r (reactive_planner.py:352-545)"""
        return (self._flags >> _abi.FX_REASON_SHIFT) & 0x7FF

    @property
    def feasabilityMap(self) -> dict:
        r = self.reasons
        return {k: float((r >> b) & 1) for k, b in _FEAS_KEYS.items()}

    @property
    def sampling_parameters(self) -> np.ndarray:
        if self._sp is None:
            self._sp = self._step.inputs.candidate_params(self.uniqueId + self._step.inputs.shard_begin)
        return self._sp

    # ---- lazily gathered from the device bundle ----
    def _need_planes(self):
        if self._planes is None:
            self._planes = self._step.fetch_sample(self.uniqueId)
        return self._planes

    def materialise(self):
        """Pull everything this sample can ever need off the device (call before the next plan step)."""
        if self._materialised:
            return self
        self._materialised = True
        if self._pkg is not None:   # the library's package is host memory already: nothing on the device this sample still needs
            if self.leaves_road:
                self._boundary_harm = self.boundary_harm
            return self
        self._need_planes()
        _ = self.costMap
        _ = self.trajectory_long
        if self.leaves_road:
            self._boundary_harm = self.boundary_harm
        return self

    @property
    def cartesian(self) -> CartesianSample:
        p = self._need_planes()
        return CartesianSample(p[0], p[1], p[2], p[3], p[4], p[5], p[6], current_time_step=self.actual_traj_length)

    @property
    def curvilinear(self) -> CurviLinearSample:
        p = self._need_planes()
        return CurviLinearSample(p[7], p[8], p[9], ss=p[10], sss=p[11], dd=p[12], ddd=p[13],
                                 current_time_step=self.actual_traj_length)

    @property
    def costMap(self) -> dict:
        if self._costmap is None and self._pkg is not None:
            raw = self._pkg.raw_cost_list()
            if raw is not None:
                names, w = self._step.inputs.cost_names, self._step.inputs.cost_weights
                self._costmap = {n: (raw[k], float(w[n] * raw[k])) for k, n in enumerate(names)}
        if self._costmap is None:
            raw = self._step.fetch_costmap_row(self.uniqueId)
            names = self._step.inputs.cost_names
            w = self._step.inputs.cost_weights
            self._costmap = {n: (float(raw[k]), float(w[n] * raw[k])) for k, n in enumerate(names)}
        return self._costmap

    def _need_coeffs(self):
        if self._coeffs is None:
            pkg = self._pkg
            self._coeffs = ((pkg.lon, pkg.lat, pkg.traj_len, pkg.tau_lat) if pkg is not None
                            else self._step.fetch_coeffs(self.uniqueId))
        return self._coeffs

    @property
    def trajectory_long(self) -> PolynomialView:
        lon = self._need_coeffs()[0]
        return PolynomialView(lon, float(self.sampling_parameters[1] - self.sampling_parameters[0]))

    @property
    def trajectory_lat(self) -> PolynomialView:
        # delta_tau is what the device built the quintic over: t at speed, the arc length s_lon_goal in LOW_VEL_MODE
        # (reactive_planner.py:161-171, stop-point bundle :650-659)
        co = self._need_coeffs()
        return PolynomialView(co[1], float(co[3]))

    @property
    def actual_traj_length(self) -> int:
        return int(self._need_coeffs()[2])

    def length(self) -> int:
        return self._step.inputs.n_samples

    def __repr__(self):
        return (f"TrajectorySample(uniqueId={self.uniqueId}, cost={self._cost:.6g}, feasible={self.feasible}, "
                f"valid={self.valid})")


class StandstillSample:
    """_compute_standstill_trajectory (reactive_planner.py:579-626): plain arrays of length N (not N+1)."""

    def __init__(self, horizon, dt, cartesian, curvilinear, trajectory_long, trajectory_lat, cost_names):
        self.horizon, self.dt = horizon, dt
        self.cartesian, self.curvilinear = cartesian, curvilinear
        self.trajectory_long, self.trajectory_lat = trajectory_long, trajectory_lat
        self.uniqueId = 0
        self.feasible = None
        self.valid = None
        self.cost = 0
        self.costMap = {n: (0, 0) for n in cost_names}
        self.feasabilityMap = {k: 0.0 for k in _FEAS_KEYS}
        self._ego_risk = self._obst_risk = self.boundary_harm = self._coll_detected = None
        self.actual_traj_length = None
        self.harm_occ_module = None


class PlanStepResult:
    """Everything one evaluated plan step produced; hands out TrajectorySample views."""

    def __init__(self, engine, inputs, result: dict, agent: int = 0):
        self.engine, self.inputs, self.result, self.agent = engine, inputs, result, agent
        self.package = None               # engine.WinnerPackage of the step's winner when the library packaged it
        self._cost = self._flags = None   # [C] arrays, read back on first use (all_traj, masks, sorted lists)
        self._stale = False
        self._samples = {}
        self.harm_coeff = (-4.591, 0.185)  # log_reg.ignore_angle const / speed (configurations/harm_parameters.json)
        self._bsteps = None

    @property
    def have_arrays(self) -> bool:
        return self._cost is not None or not hasattr(self.engine, "candidate")

    def _load_arrays(self):
        if self._cost is None:
            self._check()
            self._cost, self._flags = self.engine.costs(self.agent)

    @property
    def cost(self) -> np.ndarray:
        self._load_arrays()
        return self._cost

    @property
    def flags(self) -> np.ndarray:
        self._load_arrays()
        return self._flags

    def fetch_candidate(self, index) -> dict:
        self._check()
        return self.engine.candidate(int(index), self.agent)

    @property
    def boundary_steps(self) -> np.ndarray:
        if self._bsteps is None:
            self._check()
            self._bsteps = self.engine.boundary_steps(self.agent)
        return self._bsteps

    def invalidate(self):
        """The engine is about to run another step: device buffers will be overwritten."""
        self._stale = True

    def _check(self):
        if self._stale:
            raise RuntimeError("this plan step's device data has been overwritten; materialise() samples you keep")

    def fetch_sample(self, index):
        self._check()
        return self.engine.sample(index, self.agent)

    def fetch_costmap_row(self, index):
        self._check()
        if not hasattr(self, "_cm"):
            self._cm = self.engine.costmap(self.agent)
        return self._cm[index]

    def fetch_coeffs(self, index):
        self._check()
        return self.engine.coeffs(index, self.agent)

    def sample(self, index: int) -> TrajectorySample:
        if index not in self._samples:
            self._samples[index] = TrajectorySample(self, index)
        return self._samples[index]

    # ---- views the planner needs ----
    @property
    def n_candidates(self) -> int:
        return int(self.result["n_candidates"])

    def mask(self, bit) -> np.ndarray:
        return (self.flags & bit) != 0

    def sorted_ids(self, pool_bit=_abi.FX_FLAG_COSTED) -> np.ndarray:
        """ids of the pool in stable cost order (TrajectoryBundle.sort, trajectories.py:524-561)."""
        ids = np.nonzero(self.mask(pool_bit))[0]
        return ids[np.argsort(self.cost[ids], kind="stable")]

    def samples(self, ids) -> List[TrajectorySample]:
        """the samples of `ids` (indices within the shard), in that order"""
        todo = [int(g) for g in ids if int(g) not in self._samples]
        if len(todo) > 8:   # many new samples at once (the adapter's sorted list): built from the arrays in one go
            for t in TrajectorySample.bulk(self, todo):
                self._samples[t.uniqueId] = t
        return [self.sample(int(g)) for g in ids]

    def sorted_trajectories(self, pool_bit=_abi.FX_FLAG_COSTED, limit: Optional[int] = None) -> List[TrajectorySample]:
        ids = self.sorted_ids(pool_bit)
        if limit is not None:
            ids = ids[:limit]
        return self.samples(ids)

    @property
    def best(self) -> Optional[TrajectorySample]:
        g = self.result["best_index"] - self.inputs.shard_begin
        return self.sample(int(g)) if self.result["best_index"] >= 0 else None
