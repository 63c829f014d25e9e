"""`frenetix`-shaped calling convention over the HIP engine.

The reference's C++ back-end adapter (frenetix_motion_planner/reactive_planner_cpp.py) drives the un-vendored
`frenetix` wheel through exactly the names below; installing this module as `sys.modules["frenetix"]`
(see `install()`) lets that adapter run unchanged with `use_cpp=True`:

    handler = frenetix.TrajectoryHandler(dt=)                                   :49
    handler.add_feasability_function(ff.Check*Constraint(...))                   :96-112
    handler.add_cost_function(cf.Calculate*Cost(name, weight, ...))              :114-178
    handler.add_function(frenetix.trajectory_functions.FillCoordinates(...))     :143-149
    handler.generate_trajectories(sampling_matrix[C,13], low_vel_mode)           :256
    handler.reset_Trajectories()                                                 :330
    handler.evaluate_all_current_functions(True) / ..._concurrent(True)          :345-349
    for trajectory in handler.get_sorted_trajectories(): ...                     :353
    frenetix.CoordinateSystemWrapper(reference_path)                             :192
    frenetix.compute_initial_state(coordinate_system=, x_0=, wheelbase=, low_velocity_mode=)   :212-218
    frenetix.TrajectorySample.compute_standstill_trajectory(cs, planner_state, dt, horizon)   :226

Semantics: SURVEY.md 3.5 -- the *evaluation* follows the reference's Python path (validity, pre-filter, five
kinematic checks with the hard-coded 0.4 curvature-rate limit, NumPy cost definitions, inverse-Mahalanobis
`prediction` cost); the functor objects only carry parameters (vehicle limits, weights, v_des, predictions).
`generate_stopping_trajectories` builds the Python back-end's stop-point set (reactive_planner.py:628-671; the
upstream C++ sampler is not in the reference tree) and raises ValueError for a stop point behind the ego, which
the adapter handles by falling back to regular sampling (:336-341).
"""
import logging
import sys
import types
from typing import Dict, List, Optional

import numpy as np

from . import _abi
from .coordinate_system import CoordinateSystem
from .problem import PlanInputs, VehicleParams, pack_predictions
from .trajectories import PlanStepResult, TrajectorySample as _ViewSample

_logger = logging.getLogger("Message_logger")


# ---------------------------------------------------------------- data types
class PoseWithCovariance:
    def __init__(self, position, orientation, covariance):
        self.position = np.asarray(position, dtype=np.float64)
        self.orientation = np.asarray(orientation, dtype=np.float64)  # quaternion (x, y, z, w)
        self.covariance = np.asarray(covariance, dtype=np.float64)

    @property
    def yaw(self) -> float:
        q = self.orientation
        return float(2.0 * np.arctan2(q[2], q[3]))   # (NumPy's arctan2: math.atan2 differs in the last bit for some inputs)


class PredictedObject:
    def __init__(self, object_id: int, predicted_path: List[PoseWithCovariance], length: float, width: float):
        self.object_id, self.predictedPath, self.length, self.width = int(object_id), list(predicted_path), length, width


class CartesianPlannerState:
    def __init__(self, position, orientation, velocity, acceleration, steering_angle):
        self.pos = np.asarray(position, dtype=np.float64)
        self.orientation, self.velocity, self.acceleration, self.steering_angle = orientation, velocity, acceleration, steering_angle


class CurvilinearPlannerState:
    def __init__(self, x0_lon, x0_lat):
        self.x0_lon, self.x0_lat = np.asarray(x0_lon, dtype=np.float64), np.asarray(x0_lat, dtype=np.float64)


class PlannerState:
    def __init__(self, x_0: CartesianPlannerState, x_cl: CurvilinearPlannerState, wheelbase: float):
        self.x_0, self.x_cl, self.wheelbase = x_0, x_cl, wheelbase


class SamplingConfiguration:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class CoordinateSystemWrapper(CoordinateSystem):
    def __init__(self, reference_path):
        super().__init__(np.asarray(reference_path, dtype=np.float64))


# ---------------------------------------------------------------- functors (parameter carriers)
class _Functor:
    pass


class FillCoordinates(_Functor):
    def __init__(self, lowVelocityMode, initialOrientation, coordinateSystem, horizon):
        self.lowVelocityMode, self.initialOrientation = bool(lowVelocityMode), float(initialOrientation)
        self.coordinateSystem, self.horizon = coordinateSystem, horizon


class _Feas(_Functor):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class CheckYawRateConstraint(_Feas):
    def __init__(self, deltaMax, wheelbase, wholeTrajectory):
        super().__init__(deltaMax=deltaMax, wheelbase=wheelbase, wholeTrajectory=wholeTrajectory)


class CheckAccelerationConstraint(_Feas):
    def __init__(self, switchingVelocity, maxAcceleration, wholeTrajectory):
        super().__init__(switchingVelocity=switchingVelocity, maxAcceleration=maxAcceleration, wholeTrajectory=wholeTrajectory)


class CheckCurvatureConstraint(_Feas):
    def __init__(self, deltaMax, wheelbase, wholeTrajectory):
        super().__init__(deltaMax=deltaMax, wheelbase=wheelbase, wholeTrajectory=wholeTrajectory)


class CheckCurvatureRateConstraint(_Feas):
    def __init__(self, wheelbase, velocityDeltaMax, wholeTrajectory):
        super().__init__(wheelbase=wheelbase, velocityDeltaMax=velocityDeltaMax, wholeTrajectory=wholeTrajectory)


class _Cost(_Functor):
    cost_name = None

    def __init__(self, name, weight, *args, **kw):
        self.name, self.weight = str(name), float(weight)


def _cost_class(cls_name, cost_name):
    return type(cls_name, (_Cost,), {"cost_name": cost_name})


CalculateAccelerationCost = _cost_class("CalculateAccelerationCost", "acceleration")
CalculateJerkCost = _cost_class("CalculateJerkCost", "jerk")
CalculateLateralJerkCost = _cost_class("CalculateLateralJerkCost", "lateral_jerk")
CalculateLongitudinalJerkCost = _cost_class("CalculateLongitudinalJerkCost", "longitudinal_jerk")
CalculateOrientationOffsetCost = _cost_class("CalculateOrientationOffsetCost", "orientation_offset")
CalculateDistanceToReferencePathCost = _cost_class("CalculateDistanceToReferencePathCost", "distance_to_reference_path")


class CalculateLaneCenterOffsetCost(_Cost):
    """reactive_planner_cpp.py:135-137 constructs it with (name, weight) only.  partial_cost_functions.py:91-117 reads the
    scenario's lanelet network per trajectory point; hand it over as `lanelets=` (a Scenario, or lanelets with left_vertices /
    right_vertices) -- without it every point counts as off every lanelet (5 m, :112-115).  Parity unpinned."""
    cost_name = "lane_center_offset"

    def __init__(self, name, weight, lanelets=None):
        super().__init__(name, weight)
        self.lanelets = lanelets


class CalculateCollisionProbabilityFast(_Cost):
    cost_name = "prediction"

    def __init__(self, name, weight, predictions: Dict[int, PredictedObject], length, width, wb_rear_axle):
        super().__init__(name, weight)
        self.predictions, self.length, self.width, self.wb_rear_axle = predictions, length, width, wb_rear_axle


class CalculateDistanceToObstacleCost(_Cost):
    cost_name = "distance_to_obstacles"

    def __init__(self, name, weight, obstacle_positions):
        super().__init__(name, weight)
        self.obstacle_positions = np.asarray(obstacle_positions, dtype=np.float64).reshape(-1, 2)


class CalculateVelocityOffsetCost(_Cost):
    cost_name = "velocity_offset"

    def __init__(self, name, weight, desired_velocity, dT=0.1, t_min=1.1, limit_to_t_min=False, norm_order=2):
        super().__init__(name, weight)
        self.desired_velocity = float(desired_velocity)


# ---------------------------------------------------------------- handler
def product_grid_of(m: np.ndarray):
    """(t1 values, end-velocity values, d1 values) when the C x 13 sampling matrix is exactly their Cartesian product in row-major
    order with every other column constant and at the values the range form assumes (t0 = 0, end accelerations and lateral end
    rates 0) -- what generate_sampling_matrix (sampling_matrix.py:85-121) produces; None otherwise."""
    C = len(m)
    if C == 0:
        return None
    d_col, v_col, t_col = m[:, 10], m[:, 5], m[:, 1]
    ch = np.nonzero((v_col[1:] != v_col[:-1]) | (t_col[1:] != t_col[:-1]))[0]
    nD = int(ch[0]) + 1 if len(ch) else C
    if C % nD:
        return None
    tv = t_col[::nD]
    ch = np.nonzero(tv[1:] != tv[:-1])[0]
    nV = int(ch[0]) + 1 if len(ch) else len(tv)
    if (C // nD) % nV:
        return None
    nT = C // (nD * nV)
    t, v, d = t_col[::nD * nV].copy(), v_col[:nD * nV:nD].copy(), d_col[:nD].copy()
    if len(set(t.tolist())) != nT or len(set(v.tolist())) != nV or len(set(d.tolist())) != nD:
        return None
    want = np.empty((nT, nV, nD, 3))
    want[..., 0], want[..., 1], want[..., 2] = t[:, None, None], v[None, :, None], d[None, None, :]
    if not np.array_equal(want.reshape(C, 3), m[:, [1, 5, 10]]):
        return None
    const = m[:, [0, 2, 3, 4, 6, 7, 8, 9, 11, 12]]
    if not (const == const[0]).all() or m[0, 0] != 0.0 or m[0, 6] != 0.0 or m[0, 11] != 0.0 or m[0, 12] != 0.0:
        return None
    return t, v, d


class TrajectoryHandler:
    detect_grid = True   # evaluate a sampling matrix that is a Cartesian product as ranges (grid kernel; identical results)

    def __init__(self, dt: float, engine=None, device: int = 0):
        self.dt = float(dt)
        self._engine = engine
        self._device = device
        self._feas: Dict[type, _Feas] = {}
        self._costs: Dict[str, _Cost] = {}
        self._fill: Optional[FillCoordinates] = None
        self._matrix = None
        self._stop = None
        self._low_vel = False
        self._step: Optional[PlanStepResult] = None
        self.draw_traj_set = True      # get_sorted_trajectories() returns feasible and infeasible ones (:353-358)
        self.kinematic_debug = True

    @property
    def engine(self):
        if self._engine is None:
            from .engine import FrenetEngine
            self._engine = FrenetEngine(max_candidates=65536, max_steps=127, max_ref_knots=4096, max_obstacles=256,
                                        max_pred_steps=130, device=self._device)
        return self._engine

    # registration: later registrations of the same functor type / cost name replace earlier ones, which is what
    # re-registering "changing functions" every plan() amounts to (reactive_planner_cpp.py:143-178,303)
    def add_feasability_function(self, f):
        self._feas[type(f)] = f

    def add_cost_function(self, f):
        self._costs[f.cost_name] = f

    def add_function(self, f):
        if isinstance(f, FillCoordinates):
            self._fill = f

    def reset_Trajectories(self):
        if self._step is not None:
            self._step.invalidate()
        self._matrix, self._step, self._stop = None, None, None

    def generate_trajectories(self, sampling_matrix, low_vel_mode: bool):
        m = np.ascontiguousarray(sampling_matrix, dtype=np.float64)
        if m.ndim != 2 or m.shape[1] != 13:
            raise ValueError("sampling matrix must be C x 13")
        self._matrix, self._low_vel, self._stop = m, bool(low_vel_mode), None

    def generate_stopping_trajectories(self, planner_state, sampling_config, stop_point_s, v_stop=0.0,
                                       low_vel_mode: bool = False):
        """Stop-point candidates (reactive_planner_cpp.py:258-290).  The upstream C++ sampler behind this call is not
        in the reference tree; the set built here is the Python back-end's (reactive_planner.py:628-671): end times
        TimeSampling(t_min, min(t_max, horizon)), end positions LongitudinalPositionSampling((s0 + s_stop) / 2,
        s_stop), lateral end states LateralPositionSampling(d0 -+ d_delta) u {d0}, longitudinal quintic to (s, 0, 0).
        Raises ValueError for a stop point behind the ego, as upstream does (:263-264) -- the adapter then falls back
        to regular sampling (:336-341)."""
        from .sampling import LateralPositionSampling, LongitudinalPositionSampling, TimeSampling
        if planner_state is None or sampling_config is None or getattr(planner_state, "x_cl", None) is None:
            raise ValueError("generate_stopping_trajectories: planner state / sampling configuration missing")
        x0_lon = np.asarray(planner_state.x_cl.x0_lon, dtype=np.float64)
        x0_lat = np.asarray(planner_state.x_cl.x0_lat, dtype=np.float64)
        if stop_point_s is None or stop_point_s < x0_lon[0]:
            raise ValueError("stop point behind current longitudinal position")
        if self._fill is None:
            raise RuntimeError("FillCoordinates must be registered (add_function) before sampling")
        cfg = sampling_config
        horizon = float(self._fill.horizon) if float(self._fill.horizon) < 1000 else float(self._fill.horizon) * self.dt
        level = max(0, min(int(getattr(cfg, "sampling_level", 2)), int(round(1.0 / self.dt)) - 1))  # TimeSampling step >= dt
        t_min = max(float(getattr(cfg, "t_min", 1.1)), self.dt)
        t_max = min(float(getattr(cfg, "t_max", horizon)), horizon)
        if t_max < t_min:
            raise ValueError("empty end-time range")
        d_delta = float(getattr(cfg, "d_delta", 0.4))
        n = level + 1
        t = TimeSampling(t_min, t_max, n, self.dt).ordered(level)
        sv = LongitudinalPositionSampling((x0_lon[0] + stop_point_s) / 2, float(stop_point_s), n).ordered(level)
        d = LateralPositionSampling(x0_lat[0] - d_delta, x0_lat[0] + d_delta, n).ordered(level, float(x0_lat[0]))
        if len(t) == 0:
            raise ValueError("empty end-time range")
        self._stop = dict(t=t, s=sv, d=d, x0_lon=x0_lon, x0_lat=x0_lat)
        self._matrix, self._low_vel = None, bool(low_vel_mode)

    def _vehicle(self) -> VehicleParams:
        v = VehicleParams()
        for f in self._feas.values():
            if isinstance(f, (CheckYawRateConstraint, CheckCurvatureConstraint)):
                v.delta_max, v.wheelbase = f.deltaMax, f.wheelbase
            elif isinstance(f, CheckAccelerationConstraint):
                v.v_switch, v.a_max = f.switchingVelocity, f.maxAcceleration
            elif isinstance(f, CheckCurvatureRateConstraint):
                v.wheelbase, v.v_delta_max = f.wheelbase, f.velocityDeltaMax
        p = self._costs.get("prediction")
        if p is not None:
            v.length, v.width, v.wb_rear_axle = p.length, p.width, p.wb_rear_axle
        return v

    def _predictions(self):
        p = self._costs.get("prediction")
        if p is None or not p.predictions:
            return None
        out = {}
        for key, obj in p.predictions.items():
            path = obj.predictedPath
            out[key] = dict(pos_list=np.array([q.position[:2] for q in path]),
                            cov_list=np.array([q.covariance[:2, :2] for q in path]),
                            orientation_list=np.array([q.yaw for q in path]),
                            shape=dict(length=obj.length, width=obj.width))
        return out

    def _evaluate(self):
        if self._matrix is None and self._stop is None:
            raise RuntimeError("generate_trajectories() must be called before evaluation")
        if self._fill is None:
            raise RuntimeError("FillCoordinates must be registered (add_function) before evaluation")
        from .engine import build_obstacle_hulls
        weights = {n: f.weight for n, f in self._costs.items() if f.weight != 0}
        vo = self._costs.get("velocity_offset")
        dto = self._costs.get("distance_to_obstacles")
        lco = self._costs.get("lane_center_offset")
        if lco is not None and lco.lanelets is not None and not isinstance(lco.lanelets, dict):
            from .problem import pack_lanelets
            lco.lanelets = pack_lanelets(lco.lanelets)   # packed once per functor
        N = int(round(float(self._fill.horizon) / self.dt)) if float(self._fill.horizon) < 1000 else int(self._fill.horizon)
        preds = self._predictions()
        if self._stop is not None:
            st = self._stop
            sampling = dict(x0_lon=st["x0_lon"], x0_lat=st["x0_lat"], t_samp=st["t"], v_samp=st["s"], d_samp=st["d"],
                            stop_point=True)
        else:
            grid = product_grid_of(self._matrix) if self.detect_grid else None
            if grid is not None:
                # the adapter's matrix is itertools.product over three sets (reactive_planner_cpp.py:228-253): evaluated as
                # ranges -- the grid kernel with its shared longitudinal table, same results bit for bit
                sampling = dict(x0_lon=self._matrix[0, 2:5], x0_lat=self._matrix[0, 7:10], t_samp=grid[0], v_samp=grid[1], d_samp=grid[2])
            else:
                sampling = dict(x0_lon=self._matrix[0, 2:5], x0_lat=self._matrix[0, 7:10], sampling_matrix=self._matrix)
        inputs = PlanInputs(
            N=N, dt=self.dt, low_vel_mode=self._low_vel,
            x0_orientation=self._fill.initialOrientation, v_des=vo.desired_velocity if vo is not None else 0.0,
            vehicle=self._vehicle(), coordinate_system=self._fill.coordinateSystem, **sampling,
            cost_weights=weights, draw_traj_set=self.draw_traj_set, kinematic_debug=self.kinematic_debug,
            obstacles=pack_predictions(preds, N + 1, build_obstacle_hulls),
            dto_pos=dto.obstacle_positions if dto is not None else None, lanelets=lco.lanelets if lco is not None else None)
        if self._step is not None:
            self._step.invalidate()
        if inputs.sampling_matrix is None and hasattr(self.engine, "plan_batch"):
            # same structure as the resident upload (the usual case from the second cycle on): only state, sampling values and
            # predictions are rewritten in place (fx_update_state) instead of a full upload
            res = self.engine.plan_batch([inputs])[0]
        else:
            res = self.engine.plan_step(inputs)
        self._step = PlanStepResult(self.engine, inputs, res)
        return res

    def evaluate_all_current_functions(self, calculate_all_costs: bool = True):
        return self._evaluate()

    def evaluate_all_current_functions_concurrent(self, calculate_all_costs: bool = True):
        return self._evaluate()

    def get_sorted_trajectories(self):
        """All evaluated trajectories in stable cost order (feasible and not; the adapter splits them :353-358)."""
        if self._step is None:
            return []
        return self._step.sorted_trajectories(_abi.FX_FLAG_COSTED)

    @property
    def last_result(self) -> Optional[dict]:
        return self._step.result if self._step is not None else None


# ---------------------------------------------------------------- free functions
class _InitialState:
    def __init__(self, lon, lat):
        self.x0_lon, self.x0_lat = np.asarray(lon), np.asarray(lat)


def compute_initial_state(coordinate_system, x_0: CartesianPlannerState, wheelbase: float, low_velocity_mode: bool):
    from .reactive_planner import ReactivePlannerHip, ReactivePlannerState
    rp = ReactivePlannerHip.__new__(ReactivePlannerHip)
    rp.coordinate_system = coordinate_system
    rp.vehicle_params = VehicleParams(wheelbase=wheelbase)
    rp._LOW_VEL_MODE = bool(low_velocity_mode)
    st = ReactivePlannerState(position=x_0.pos, orientation=x_0.orientation, velocity=x_0.velocity,
                              acceleration=x_0.acceleration, steering_angle=x_0.steering_angle)
    lon, lat = rp._compute_initial_states(st)
    return _InitialState(lon, lat)


class TrajectorySample(_ViewSample):
    @staticmethod
    def compute_standstill_trajectory(coordinate_system, planner_state: PlannerState, dt: float, horizon: float):
        from .reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
        rp = ReactivePlannerHip(PlannerConfig(dt=dt, planning_horizon=horizon), VehicleParams(wheelbase=planner_state.wheelbase),
                                engine=object())
        rp.coordinate_system = coordinate_system
        c = planner_state.x_0
        rp.x_0 = ReactivePlannerState(position=c.pos, orientation=c.orientation, velocity=c.velocity,
                                      acceleration=c.acceleration, steering_angle=c.steering_angle)
        rp.x_cl = (list(planner_state.x_cl.x0_lon), list(planner_state.x_cl.x0_lat))
        return rp._compute_standstill_trajectory()


def setup_logger(logger):
    global _logger
    _logger = logger


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def install(force: bool = False):
    """Register this module tree as `frenetix` so `import frenetix` (reactive_planner_cpp.py:21-24) resolves here."""
    if "frenetix" in sys.modules and not force:
        return sys.modules["frenetix"]
    ff = _module("frenetix.trajectory_functions.feasability_functions", CheckYawRateConstraint=CheckYawRateConstraint,
                 CheckAccelerationConstraint=CheckAccelerationConstraint, CheckCurvatureConstraint=CheckCurvatureConstraint,
                 CheckCurvatureRateConstraint=CheckCurvatureRateConstraint)
    cf = _module("frenetix.trajectory_functions.cost_functions", CalculateAccelerationCost=CalculateAccelerationCost,
                 CalculateJerkCost=CalculateJerkCost, CalculateLateralJerkCost=CalculateLateralJerkCost,
                 CalculateLongitudinalJerkCost=CalculateLongitudinalJerkCost,
                 CalculateOrientationOffsetCost=CalculateOrientationOffsetCost,
                 CalculateLaneCenterOffsetCost=CalculateLaneCenterOffsetCost,
                 CalculateDistanceToReferencePathCost=CalculateDistanceToReferencePathCost,
                 CalculateCollisionProbabilityFast=CalculateCollisionProbabilityFast,
                 CalculateDistanceToObstacleCost=CalculateDistanceToObstacleCost,
                 CalculateVelocityOffsetCost=CalculateVelocityOffsetCost)
    tf = _module("frenetix.trajectory_functions", FillCoordinates=FillCoordinates, feasability_functions=ff, cost_functions=cf)
    inner = _module("frenetix._frenetix", setup_logger=setup_logger)
    top = _module("frenetix", TrajectoryHandler=TrajectoryHandler, CoordinateSystemWrapper=CoordinateSystemWrapper,
                  PoseWithCovariance=PoseWithCovariance, PredictedObject=PredictedObject,
                  CartesianPlannerState=CartesianPlannerState, CurvilinearPlannerState=CurvilinearPlannerState,
                  PlannerState=PlannerState, SamplingConfiguration=SamplingConfiguration,
                  compute_initial_state=compute_initial_state, TrajectorySample=TrajectorySample,
                  trajectory_functions=tf, _frenetix=inner, __version__="0.4.0+fxplan")
    for m in (top, tf, ff, cf, inner):
        sys.modules[m.__name__] = m
    return top
