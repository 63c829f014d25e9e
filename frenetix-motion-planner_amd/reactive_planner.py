"""ReactivePlannerHip -- the planner front-end over the HIP engine; same surface as the reference's
`ReactivePlannerPython` / `ReactivePlannerCpp` (frenetix_motion_planner/reactive_planner.py:32-130,
reactive_planner_cpp.py:32-441, base class planner.py:48-710) for the hot path.

What is mirrored: update_externals() and the setters, the sampling-level escalation loop of plan(), the
Frenet initial state (_compute_initial_states, planner.py:567-635), the velocity sampling range
(set_desired_velocity, :292-310), the cost-ordered walk with the dynamic-obstacle collision check (GPU) and an
optional host-side road-boundary callback (planner.py:362-390), standstill fallback
(reactive_planner.py:114-118,579-626), output packaging (_compute_trajectory_pair, planner.py:394-447 with
shift_orientation :536-542) and the counters `_infeasible_count_kinematics`, `infeasible_kinematics_percentage`,
`infeasible_count_collision`, `all_traj`, `optimal_trajectory`, `trajectory_pair`, `x_cl`.

What is not: CommonRoad containers (scenario, planning problem, DynamicObstacle, Trajectory) -- not installed
here and outside the hot path; states are plain dataclasses with the same field names.  Risk/harm logging,
occlusion module, behaviour planner hooks, SQLite/CSV logging are out of scope.
"""
import logging
import math
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import _abi
from .coordinate_system import CoordinateSystem, interpolate_angle
from .problem import DEFAULT_COST_WEIGHTS, MAX_OBSTACLES, PlanInputs, VehicleParams, pack_predictions
from .sampling import SamplingHandler, dense_cached, v_sampling_bounds

try:   # the CPython helper built next to the package (csrc/fx_host_ext.c); PlanInputs.next_step is the same thing in Python
    from ._fxhost import next_inputs as _NEXT_INPUTS
except ImportError:
    _NEXT_INPUTS = None
try:
    from ._fxhost import next_inputs_level as _NEXT_INPUTS_LEVEL
except ImportError:
    _NEXT_INPUTS_LEVEL = None
from .trajectories import (CartesianSample, CurviLinearSample, PlanStepResult, PolynomialView, StandstillSample,
                           TrajectorySample)


@dataclass
class ReactivePlannerState:
    """frenetix_motion_planner/state.py:16-75: rear-axle KS state + acceleration and yaw rate."""
    time_step: int = 0
    position: np.ndarray = field(default_factory=lambda: np.zeros(2))
    orientation: float = 0.0
    velocity: float = 0.0
    acceleration: float = 0.0
    yaw_rate: float = 0.0
    steering_angle: float = 0.0

    def __deepcopy__(self, memo=None) -> "ReactivePlannerState":
        # the only mutable member is the position array (copy.deepcopy's generic walk costs 100 us per state)
        return ReactivePlannerState(self.time_step, np.array(self.position, dtype=np.float64), self.orientation, self.velocity,
                                    self.acceleration, self.yaw_rate, self.steering_angle)

    def shift_positions_to_center(self, wb_rear_axle: float) -> "ReactivePlannerState":
        o = self.orientation
        return ReactivePlannerState(self.time_step, np.asarray(self.position) + wb_rear_axle * np.array([np.cos(o), np.sin(o)]),
                                    o, self.velocity, self.acceleration, self.yaw_rate, self.steering_angle)


@dataclass
class PlannerConfig:
    """configurations/frenetix_motion_planner/{planning,cost,debug}.yaml defaults the hot path reads."""
    dt: float = 0.1
    planning_horizon: float = 3.0
    low_vel_mode_threshold: float = 2.0
    replanning_frequency: int = 3
    t_min: float = 1.1
    d_min: float = -3.0
    d_max: float = 3.0
    d_ego_pos: bool = False
    sampling_min: int = 2
    sampling_max: int = 3
    emergency_mode: str = "stopping"
    # C++ back-end's stopping-trajectory selection when nothing is collision-free (reactive_planner_cpp.py:404-413).  The Python
    # back-end instead takes, at the LAST sampling level, the feasible trajectory with the lowest ego + obstacle risk
    # (reactive_planner.py:262-269, set_risk_costs): the harm model behind that is out of scope, the hook is
    # ReactivePlannerHip.set_fallback_selector / min_risk_selector below.  With neither configured a plan step whose feasible
    # candidates all collide returns None (plus the standstill trajectory at v <= 0.1, reactive_planner.py:105-109).
    emergency_selection: bool = False
    cost_weights: Dict[str, float] = field(default_factory=lambda: dict(DEFAULT_COST_WEIGHTS))
    draw_traj_set: bool = True       # debug.yaml:8
    kinematic_debug: bool = True     # debug.yaml:20
    save_all_traj: bool = False
    survivors: int = 16              # top-k kept for the host-side road-boundary walk
    # dense grid (n_t, n_v, n_d) in natural order instead of the reference's sampling levels (BASELINE configs 2 - 5): T from
    # t_min in steps of dt, V over the planner's velocity range, D over [d_min, d_max] plus the current d
    dense_grid: Optional[Tuple[int, int, int]] = None


class ReactivePlannerHip:
    def __init__(self, config: Optional[PlannerConfig] = None, vehicle: Optional[VehicleParams] = None, engine=None,
                 msg_logger=None, device: int = 0,
                 road_boundary_check: Optional[Callable[[TrajectorySample], float]] = None):
        self.config = config or PlannerConfig()
        self.vehicle_params = vehicle or VehicleParams()
        self.horizon = self.config.planning_horizon
        self.dT = self.config.dt
        self.N = int(self.config.planning_horizon / self.config.dt)
        assert self.dT > 0 and self.N > 0 and self.horizon > 0  # planner.py:544-548
        self._low_vel_mode_threshold = self.config.low_vel_mode_threshold
        self.msg_logger = msg_logger or logging.getLogger("Message_logger")
        self._device = device
        self._engine = engine
        self.road_boundary_check = road_boundary_check
        self.fallback_selector: Optional[Callable] = None   # last-level selection among colliding candidates (set_fallback_selector)
        self.occlusion_module = None       # planner.py:99 (set_occlusion_module)
        self.use_occ_model = False
        self.road_boundary = None          # segments [n][4]; checked on the GPU (set_road_boundary)
        self._packed_boundary = None
        self.params_harm = {"log_reg": {"ignore_angle": {"const": -4.591, "speed": 0.185}}}  # configurations/harm_parameters.json

        self.x_0: Optional[ReactivePlannerState] = None
        self.x_cl: Optional[Tuple[List, List]] = None
        self.reference_path = None
        self.coordinate_system: Optional[CoordinateSystem] = None
        self.set_new_ref_path = None
        self._LOW_VEL_MODE = False
        self.predictions = None
        self.use_prediction = False
        self.desired_velocity = None
        self.cost_weights = dict(self.config.cost_weights)
        self._weights_nz = self._weights_src = self._weights_sig = None
        self._sampling_min, self._sampling_max = self.config.sampling_min, self.config.sampling_max
        self.sampling_handler = SamplingHandler(dt=self.dT, max_sampling_number=self.config.sampling_max,
                                                t_min=self.config.t_min, horizon=self.horizon,
                                                delta_d_max=self.config.d_max, delta_d_min=self.config.d_min,
                                                d_ego_pos=self.config.d_ego_pos)
        self._draw_traj_set = self.config.draw_traj_set
        self._kinematic_debug = self.config.kinematic_debug
        self.save_all_traj = self.config.save_all_traj

        self._collision_counter = 0
        self._total_count = 0
        self._infeasible_count_kinematics = None
        self.infeasible_kinematics_percentage = None
        self.all_traj = None
        self.optimal_trajectory = None
        self.trajectory_pair = None
        self.ego_vehicle_history = []
        self.last_step: Optional[PlanStepResult] = None
        self.planning_time = None
        self._packed_predictions = None
        self._packed_lanelets = None      # set_lanelets: the lane_center_offset cost's lanelets
        self._prev_inputs = None          # the last PlanInputs built (a closed loop's next step differs in a few fields)
        self._prev_level = self._prev_sh = self._prev_t = None   # the sampling level / handler / time sampling those inputs were built from
        self.logger = None               # logging_formats.DataLoggingCosts (planner.py:150-158)
        self.record_state_list = []
        self.record_input_list = []

    # ------------------------------------------------------------------ engine
    @property
    def engine(self):
        if self._engine is None:
            from .engine import FrenetEngine
            cap = self.max_candidates_per_step()
            self._engine = FrenetEngine(max_candidates=max(cap, 4096), max_steps=self.N, max_ref_knots=4096,
                                        max_obstacles=MAX_OBSTACLES, max_pred_steps=max(64, self.N + 2), device=self._device)
        return self._engine

    def max_candidates_per_step(self) -> int:
        """Candidates of the densest sampling level this planner can reach (or of its dense grid): the capacity its engine
        needs so that the level escalation of plan() never outgrows it."""
        if self.config.dense_grid is not None:
            n_t, n_v, n_d = self.config.dense_grid
            return int(n_t * n_v * (n_d + 1))
        return self.sampling_handler.max_candidates(range(min(self._sampling_min, self._sampling_max - 1), self._sampling_max),
                                                    cpp_style=True)

    @property
    def infeasible_count_collision(self):
        return self._collision_counter

    # ------------------------------------------------------------------ externals (planner.py:172-217)
    def update_externals(self, reference_path: np.ndarray = None, x_0: ReactivePlannerState = None, x_cl=None,
                         cost_weights=None, desired_velocity: float = None, predictions=None, **ignored):
        if reference_path is not None:
            self.reference_path = reference_path
            self.set_reference_and_coordinate_system(reference_path)
        if x_0 is not None:
            self.set_x_0(x_0)
            self.set_x_cl(x_cl)
        if cost_weights is not None:
            self.set_cost_function(cost_weights)
        if desired_velocity is not None:
            self.set_desired_velocity(desired_velocity, x_0.velocity if x_0 is not None else self.x_0.velocity)
        if predictions is not None:
            self.set_predictions(predictions)
        if self.sampling_handler.d_ego_pos:
            self.sampling_handler.set_d_sampling(self.x_cl[1][0])

    def update_step(self, x_0: ReactivePlannerState, x_cl, desired_velocity: float, predictions):
        """update_externals(x_0=, x_cl=, desired_velocity=, predictions=) as a closed-loop step calls it (planner.py:172-217 from
        frenet_interface.py:172-205) with the four setters inlined; anything else -- a new reference path, no x_cl yet, lateral
        sampling around the ego -- goes through update_externals."""
        if self.x_cl is None or self.set_new_ref_path or x_cl is None or self.sampling_handler.d_ego_pos or desired_velocity is None:
            return self.update_externals(x_0=x_0, x_cl=x_cl, desired_velocity=desired_velocity, predictions=predictions)
        self.x_0 = x_0
        v = x_0.velocity
        self._LOW_VEL_MODE = bool(v < self._low_vel_mode_threshold)
        self.x_cl = x_cl
        self.desired_velocity = desired_velocity
        vp = self.vehicle_params
        self.sampling_handler.set_v_sampling(*v_sampling_bounds(v, vp.a_max, self.horizon, vp.v_max, 36))
        if predictions is not None:
            self.use_prediction = True
            self.predictions = predictions
            self._packed_predictions = predictions.packed if hasattr(predictions, "packed") else None

    def set_reference_and_coordinate_system(self, reference_path: np.ndarray):
        self.coordinate_system = CoordinateSystem(reference_path)
        self.set_new_ref_path = True

    def set_x_0(self, x_0: ReactivePlannerState):
        self.x_0 = x_0
        self._LOW_VEL_MODE = bool(x_0.velocity < self._low_vel_mode_threshold)

    def set_x_cl(self, x_cl):
        if self.x_cl is not None and not self.set_new_ref_path and x_cl is not None:
            self.x_cl = x_cl
        else:
            self.x_cl = self._compute_initial_states(self.x_0)
            self.set_new_ref_path = False

    def set_cost_function(self, cost_weights):
        self.cost_weights = dict(cost_weights)
        self._weights_nz = None

    def set_fallback_selector(self, selector: Optional[Callable]):
        """What happens at the LAST sampling level when no feasible candidate is collision-free.  The reference's Python
        back-end computes `set_risk_costs` for every feasible trajectory and returns
        `sorted(feasible_trajectories, key=lambda traj: traj._ego_risk + traj._obst_risk)[0]` (reactive_planner.py:262-269).
        `selector(feasible)` is called with the feasible trajectories in creation order -- a lazy sequence of TrajectorySample
        views over the step's device results (valid and feasible, every one of them colliding) -- and returns the chosen
        sample or None.  `min_risk_selector(risk)` builds the reference's rule from a per-trajectory risk function; the harm
        model that computes the reference's risk is outside this package (out of scope)."""
        self.fallback_selector = selector

    def set_occlusion_module(self, occ_module):
        """planner.py:271-273.  The occlusion module itself is outside this package (it is not in the reference
        tree either); its two call points are: `occ_module.calc_costs(trajectories)` on the step's feasible trajectories before
        they are sorted (trajectories.py:557-560) and `occ_module.trajectory_safety_assessment(trajectory) -> (metric, ok)` for the
        collision-free candidates of the cost order until one passes (planner.py:384-388).  With a module set the selection of a
        step runs on the host over TrajectorySample views of the device's results (costs, collision and road-boundary flags of
        EVERY candidate are there already); None switches back to the device's selection."""
        self.occlusion_module = occ_module
        self.use_occ_model = occ_module is not None

    @staticmethod
    def min_risk_selector(risk: Callable[[TrajectorySample], float]) -> Callable:
        """sorted(feasible, key=risk)[0] -- the first trajectory of the lowest risk, in creation order (Python's sort is stable,
        so this is min() over the sequence)."""
        def select(feasible):
            return min(feasible, key=risk) if len(feasible) else None
        return select

    def set_road_boundary(self, segments):
        """Road boundary as straight segments [n][4] = (ax, ay, bx, by) (planner.py:550-565 builds it once per
        scenario).  The footprint test of planner.py:362-381 then runs inside the evaluation kernel for every
        candidate; trajectories that leave the road carry `boundary_harm` != 0 and are never selected."""
        self.road_boundary = None if segments is None else np.ascontiguousarray(segments, dtype=np.float64).reshape(-1, 4)
        self._packed_boundary = None

    def set_lanelets(self, lanelets):
        """The lanelet network the `lane_center_offset` cost reads (partial_cost_functions.py:91-117 asks the scenario's
        lanelet_network per trajectory point): a Scenario, a dict / list of lanelets (left_vertices, right_vertices), or None.
        Packed once (problem.pack_lanelets); without it every point counts as off every lanelet (5 m, :112-115)."""
        from .problem import pack_lanelets
        self._packed_lanelets = None if lanelets is None else pack_lanelets(lanelets)

    def set_predictions(self, predictions: dict):
        """predictions: the reference's dict (prediction_helpers.py:209-261) -- or problem.PackedPredictions, the same content
        already packed (a batch of agents packs its shared predictions once)."""
        self.use_prediction = True
        self.predictions = predictions
        self._packed_predictions = predictions.packed if hasattr(predictions, "packed") else None

    def set_sampling_parameters(self, t_min: float, horizon: float, delta_d_min: float, delta_d_max: float):
        self.sampling_handler.update_static_params(t_min, horizon, delta_d_min, delta_d_max)

    def set_desired_velocity(self, desired_velocity: float, current_speed: float = None, stopping: bool = False,
                             v_limit: float = 36):
        self.desired_velocity = desired_velocity
        min_v, max_v = v_sampling_bounds(current_speed, self.vehicle_params.a_max, self.horizon, self.vehicle_params.v_max,
                                         v_limit)
        self.sampling_handler.set_v_sampling(min_v, max_v)

    # ------------------------------------------------------------------ initial Frenet state (planner.py:567-635)
    def _compute_initial_states(self, x_0: ReactivePlannerState):
        """x_0 -> x_cl along the current reference; the ego curvature comes from the steering angle, tan(delta) / wheelbase"""
        try:
            return self.coordinate_system.frenet_state(
                x_0.position[0], x_0.position[1], x_0.orientation, x_0.velocity, x_0.acceleration,
                np.tan(x_0.steering_angle) / self.vehicle_params.wheelbase, arc_length_lateral=self._LOW_VEL_MODE)
        except ValueError as ex:
            if "faces against" in str(ex):
                raise
            raise ValueError("Initial state could not be transformed.") from ex

    # ------------------------------------------------------------------ plan (reactive_planner.py:67-130)
    def _inputs_for_level(self, samp_level: int, stop_point_s: Optional[float] = None) -> PlanInputs:
        from .engine import build_obstacle_hulls
        x_lon, x_lat = self.x_cl
        prev = self._prev_inputs
        if self._packed_predictions is None:
            self._packed_predictions = pack_predictions(self.predictions if self.use_prediction else None, self.N + 1,
                                                        build_obstacle_hulls)
        if (_NEXT_INPUTS is not None and prev is not None and stop_point_s is None and self.config.dense_grid is not None
                and self.road_boundary is None and prev.road_boundary is None
                and not prev.stop_point and prev.sampling_matrix is None):
            # the closed loop's usual step on a dense grid, in ONE extension call (`_fxhost.next_inputs`): the velocity set from
            # its bounds, the lateral set (+ the current d), the state arrays and the copy of the last inputs with those replaced
            cw = self.cost_weights
            if (self._weights_src is cw and self._weights_sig == tuple(cw.items()) and prev.cost_weights is self._weights_nz
                    and prev.coordinate_system is self.coordinate_system and prev.lanelets is self._packed_lanelets
                    and prev.vehicle is self.vehicle_params and prev.N == self.N and prev.dt == self.dT
                    and prev.draw_traj_set == self._draw_traj_set and prev.kinematic_debug == self._kinematic_debug
                    and prev.collision == self.use_prediction):
                n_t, n_v, n_d = self.config.dense_grid
                t_c, d_c, _, _ = dense_cached(n_t, n_v, n_d, self.horizon, self.dT, self.config.t_min, self.config.d_min, self.config.d_max)
                vs = self.sampling_handler.v_sampling
                if prev.t_samp is t_c and len(prev.v_samp) == n_v:
                    inp = _NEXT_INPUTS(prev, self._LOW_VEL_MODE, x_lon, x_lat, self.x_0.orientation, self.desired_velocity,
                                       vs.minimum, vs.maximum, d_c, x_lat[0], self._packed_predictions)
                    if inp is not NotImplemented:
                        self._prev_inputs = inp
                        return inp
        if (_NEXT_INPUTS_LEVEL is not None and prev is not None and stop_point_s is None and self.config.dense_grid is None
                and self.road_boundary is None and prev.road_boundary is None and not prev.stop_point
                and prev.sampling_matrix is None and self._prev_level == samp_level):
            # the closed loop's usual step at a sampling level of the reference, in ONE extension call (`_fxhost.next_inputs_level`):
            # set(np.linspace(v_min, v_max, n)) and d_level.union({d}) in the sets' own iteration order (built with the same set
            # operations, in C), the state arrays and the copy of the last inputs with those replaced; the time set of the level is
            # the last step's
            cw = self.cost_weights
            if (self._weights_src is cw and self._weights_sig == tuple(cw.items()) and prev.cost_weights is self._weights_nz
                    and prev.coordinate_system is self.coordinate_system and prev.lanelets is self._packed_lanelets
                    and prev.vehicle is self.vehicle_params and prev.N == self.N and prev.dt == self.dT
                    and prev.draw_traj_set == self._draw_traj_set and prev.kinematic_debug == self._kinematic_debug
                    and prev.collision == self.use_prediction):
                sh = self.sampling_handler
                vs = sh.v_sampling
                if 0 <= samp_level < vs.max_density and self._prev_sh is sh and self._prev_t is sh.t_sampling:
                    inp = _NEXT_INPUTS_LEVEL(prev, self._LOW_VEL_MODE, x_lon, x_lat, self.x_0.orientation, self.desired_velocity,
                                             vs.minimum, vs.maximum, 2 ** (samp_level + 1) + 1, sh.d_sampling.to_range(samp_level),
                                             x_lat[0], self._packed_predictions)
                    if inp is not NotImplemented:
                        self._prev_inputs = inp
                        return inp
        if self.config.dense_grid is not None and stop_point_s is None:
            from .sampling import dense_ranges
            n_t, n_v, n_d = self.config.dense_grid
            vs = self.sampling_handler.v_sampling
            t, v, d = dense_ranges(n_t, n_v, n_d, vs.minimum, vs.maximum, self.horizon, self.dT, x_lat[0], self.config.t_min,
                                   self.config.d_min, self.config.d_max)
        else:
            t, v, d = self.sampling_handler.ordered_ranges(samp_level, x_lat[0])
        if stop_point_s is not None:
            # stop-point sampling: end positions in [(s0 + s_stop) / 2, s_stop] (reactive_planner.py:637-643)
            self.sampling_handler.set_s_sampling((x_lon[0] + stop_point_s) / 2, stop_point_s)
            v = self.sampling_handler.s_sampling.ordered(samp_level)
        if self._packed_predictions is None:
            self._packed_predictions = pack_predictions(self.predictions if self.use_prediction else None, self.N + 1,
                                                        build_obstacle_hulls)
        boundary = None
        if self.road_boundary is not None and len(self.road_boundary):
            from .problem import pack_road_boundary
            d_reach = 1.5 * max(float(np.max(np.abs(d))), abs(float(x_lat[0]))) + 2.0
            pb = self._packed_boundary
            if pb is None or pb["d_reach"] < d_reach or len(pb["bin"]) != len(self.coordinate_system.reference) + 1:
                pb = self._packed_boundary = pack_road_boundary(self.road_boundary, self.coordinate_system,
                                                                self.vehicle_params, d_reach)
            boundary = pb
        weights, cw = self._weights_nz, self.cost_weights
        sig = tuple(cw.items())   # exact: an in-place swap of two weights keeps length and sum
        if weights is None or self._weights_src is not cw or self._weights_sig != sig:   # replaced, or edited in place
            weights = self._weights_nz = {k: w for k, w in cw.items() if w != 0}
            self._weights_src, self._weights_sig = cw, sig
        prev = self._prev_inputs
        if (prev is not None and prev.cost_weights is weights and prev.coordinate_system is self.coordinate_system
                and prev.road_boundary is boundary and prev.lanelets is self._packed_lanelets and prev.vehicle is self.vehicle_params and prev.N == self.N and prev.dt == self.dT
                and prev.stop_point == (stop_point_s is not None) and prev.draw_traj_set == self._draw_traj_set
                and prev.kinematic_debug == self._kinematic_debug and prev.collision == self.use_prediction):
            # the closed loop's usual step: only the state, the sampling values and the predictions differ from the last inputs
            inp = prev.next_step(low_vel_mode=self._LOW_VEL_MODE, x0_lon=x_lon, x0_lat=x_lat, x0_orientation=self.x_0.orientation,
                                 v_des=self.desired_velocity, t_samp=t, v_samp=v, d_samp=d, obstacles=self._packed_predictions)
        else:
            inp = PlanInputs(N=self.N, dt=self.dT, low_vel_mode=self._LOW_VEL_MODE, x0_lon=x_lon, x0_lat=x_lat,
                             x0_orientation=self.x_0.orientation, v_des=self.desired_velocity, vehicle=self.vehicle_params,
                             coordinate_system=self.coordinate_system, t_samp=t, v_samp=v, d_samp=d,
                             stop_point=stop_point_s is not None, cost_weights=weights,
                             draw_traj_set=self._draw_traj_set, kinematic_debug=self._kinematic_debug, write_bundle=True,
                             write_costmap=True, collision=self.use_prediction, obstacles=self._packed_predictions,
                             road_boundary=boundary, lanelets=self._packed_lanelets)
        self._prev_inputs = inp
        # (what the one-call path above needs to know of THIS step: its level, and that the level's time set came from this handler)
        self._prev_level = samp_level if (self.config.dense_grid is None and stop_point_s is None) else None
        self._prev_sh, self._prev_t = self.sampling_handler, self.sampling_handler.t_sampling
        return inp

    def _create_end_point_trajectory_bundle(self, x_0_lon, x_0_lat, stop_point_s, samp_level: int) -> PlanInputs:
        """Stop-point sampling set of reactive_planner.py:628-671 as engine inputs: T x S x (D u {d0}) with the
        longitudinal quintic to (s, 0, 0), s in LongitudinalPositionSampling((s0 + s_stop) / 2, s_stop)."""
        saved = self.x_cl
        self.x_cl = (list(x_0_lon), list(x_0_lat))
        try:
            return self._inputs_for_level(samp_level, stop_point_s=float(stop_point_s))
        finally:
            self.x_cl = saved

    def plan(self, stop_point_s: Optional[float] = None):
        """One plan step.  stop_point_s: arc length of a stop point ahead of the ego -> the candidates are the
        stop-point set (the reference's C++ back-end switches to it when the behaviour planner supplies a stop point
        with v < 10 m/s, reactive_planner_cpp.py:332-341; a stop point behind the ego raises ValueError there and
        falls back to regular sampling, which is what happens here too)."""
        if self.x_cl is None:
            raise RuntimeError("x_cl should have been set prior to plan()")  # reactive_planner_cpp.py:308-309
        if self.desired_velocity is None:
            raise RuntimeError("desired velocity not set (update_externals(desired_velocity=...))")
        if stop_point_s is not None and stop_point_s < self.x_cl[0][0]:
            self.msg_logger.info("stop point behind current longitudinal position, falling back to regular planning")
            stop_point_s = None  # reactive_planner_cpp.py:263-264,336-341
        optimal_trajectory = None
        t0 = time.time()
        samp_level = self._sampling_min
        while optimal_trajectory is None and samp_level < self._sampling_max:
            optimal_trajectory = self._get_optimal_trajectory(self._inputs_for_level(samp_level, stop_point_s), samp_level)
            samp_level += 1
        return self.plan_finish(optimal_trajectory, t0)

    # -- the same step in phases, so that a batch of planners can share ONE launch (multiagent.AgentBatchHip) --
    def plan_begin(self, stop_point_s: Optional[float] = None) -> PlanInputs:
        """Inputs of the first sampling level of this plan step (checks as in plan())."""
        if self.x_cl is None:
            raise RuntimeError("x_cl should have been set prior to plan()")
        if self.desired_velocity is None:
            raise RuntimeError("desired velocity not set (update_externals(desired_velocity=...))")
        if stop_point_s is not None and stop_point_s < self.x_cl[0][0]:
            stop_point_s = None
        self._stop_point_s = stop_point_s
        return self._inputs_for_level(self._sampling_min, stop_point_s)

    def plan_consume(self, inputs: PlanInputs, res: dict, engine, agent: int = 0, package=False):
        """Take this planner's share of a batched launch (first sampling level); returns the chosen trajectory
        (materialised -- the batch engine's buffers are reused) or None when the level has to escalate.
        package: the winner as the batched call already packaged it (engine.plan_batch_packaged; None = nothing found);
        False: read it here."""
        if package is False:
            package = engine.package(agent, self.x_0.yaw_rate) if getattr(engine, "packaging", False) else None
        best = self._consume_result(inputs, res, engine, agent, package, self._sampling_min)
        if best is not None:
            best.materialise()
        return best

    def plan_escalate(self, t0: float):
        """Remaining sampling levels on the planner's own engine after the batched first level found nothing."""
        optimal_trajectory = None
        samp_level = self._sampling_min + 1
        while optimal_trajectory is None and samp_level < self._sampling_max:
            optimal_trajectory = self._get_optimal_trajectory(
                self._inputs_for_level(samp_level, getattr(self, "_stop_point_s", None)), samp_level)
            samp_level += 1
        return self.plan_finish(optimal_trajectory, t0)

    def plan_finish(self, optimal_trajectory, t0: float):
        """reactive_planner.py:99-130: packaging, standstill fallback, logging hook."""
        self.planning_time = time.time() - t0

        self.trajectory_pair = self._compute_trajectory_pair(optimal_trajectory) if optimal_trajectory is not None else None
        if self.trajectory_pair is not None:
            self.ego_vehicle_history.append(self.trajectory_pair[0])
        if optimal_trajectory is None and self.x_0.velocity <= 0.1:
            self.msg_logger.warning('Planning standstill for the current scenario')
            optimal_trajectory = self._compute_standstill_trajectory()
        # no collision-free trajectory but kinematically feasible ones: the C++ back-end's emergency selection
        # (reactive_planner_cpp.py:404-413; the "risk" mode needs the harm model and stays outside)
        if optimal_trajectory is None and self.config.emergency_mode == "stopping" and self.config.emergency_selection \
                and self.last_step is not None:
            optimal_trajectory = self._select_stopping_trajectory(self.last_step, self.x_cl[1][0])
            if optimal_trajectory is not None:
                self.msg_logger.warning("No optimal trajectory available. Select stopping trajectory!")
                self.trajectory_pair = self._compute_trajectory_pair(optimal_trajectory)
                self.ego_vehicle_history.append(self.trajectory_pair[0])
        if optimal_trajectory is not None and hasattr(optimal_trajectory, "materialise"):
            optimal_trajectory.materialise()  # survives the next step's overwrite of the device bundle
        self.optimal_trajectory = optimal_trajectory
        self.plan_postprocessing(optimal_trajectory, self.planning_time)
        return self.trajectory_pair

    @staticmethod
    def _select_stopping_trajectory(step: PlanStepResult, d_pos: float):
        """reactive_planner_cpp.py:443-466: of the kinematically feasible candidates the one with the lowest end
        velocity, then the shortest horizon, then the lateral end position closest to the current one -- the first
        existing combination of product(unique v, unique t, d sorted by |d - d_pos|)."""
        inp = step.inputs
        if inp.sampling_matrix is not None:
            return None
        feas = np.nonzero(step.mask(_abi.FX_FLAG_VALID) & step.mask(_abi.FX_FLAG_FEASIBLE) & step.mask(_abi.FX_FLAG_RETURNED))[0]
        if len(feas) == 0:
            return None
        t, v, d = np.asarray(inp.t_samp), np.asarray(inp.v_samp), np.asarray(inp.d_samp)
        g = feas + inp.shard_begin
        i_d = g % len(d)
        pair = g // len(d)
        i_v, i_t = pair % len(v), pair // len(v)
        d_sorted = np.unique(d)
        d_rank_of = {float(x): r for r, x in enumerate(d_sorted[np.argsort(np.abs(d_sorted - d_pos), kind="stable")])}
        d_rank = np.array([d_rank_of[float(x)] for x in d[i_d]])
        order = np.lexsort((d_rank, t[i_t], v[i_v]))
        return step.sample(int(feas[order[0]]))

    def plan_postprocessing(self, optimal_trajectory, planning_time, replanning_counter=0):
        """planner.py:637-649: the logging hook (logging_formats.DataLoggingCosts as `self.logger`)."""
        if self.logger is None:
            return
        if optimal_trajectory is not None:
            ego = self.ego_vehicle_history[-1] if self.ego_vehicle_history else None
            self.logger.log(optimal_trajectory, time_step=self.x_0.time_step,
                            infeasible_kinematics=self._infeasible_count_kinematics,
                            percentage_kinematics=self.infeasible_kinematics_percentage, planning_time=planning_time,
                            ego_vehicle=ego, desired_velocity=self.desired_velocity, replanning_counter=replanning_counter)
            self.logger.log_predicition(self.predictions)
        if self.save_all_traj and self.all_traj is not None and replanning_counter == 0:
            self.logger.log_all_trajectories(self.all_traj, self.x_0.time_step)

    def _get_optimal_trajectory(self, inputs: PlanInputs, samp_lvl: int):
        """reactive_planner.py:184-272: feasibility, costs, stable sort, collision walk -- one fused launch."""
        if self.last_step is not None:
            self.last_step.invalidate()
        pkg = None
        if hasattr(self.engine, "plan_step_packaged"):
            # one call across the boundary: in-place update of the resident inputs, evaluation, result, the winner packaged
            res, pkg = self.engine.plan_step_packaged(inputs, self.x_0.yaw_rate)
        else:
            res = self.engine.plan_step(inputs)
        return self._consume_result(inputs, res, self.engine, 0, pkg, samp_lvl)

    def _fallback(self, step, samp_lvl):
        """reactive_planner.py:262-269: last level, nothing collision-free, feasible trajectories exist -> the selector's choice"""
        if self.fallback_selector is None or samp_lvl is None or samp_lvl < self._sampling_max - 1:
            return None
        ids = np.nonzero(step.mask(_abi.FX_FLAG_VALID) & step.mask(_abi.FX_FLAG_FEASIBLE) & step.mask(_abi.FX_FLAG_RETURNED))[0]
        if len(ids) == 0:
            return None
        chosen = self.fallback_selector(_LazySamples(step, ids))
        if chosen is not None:
            self.msg_logger.warning("No optimal trajectory available. Select lowest risk trajectory!")
        return chosen

    def _consume_result(self, inputs: PlanInputs, res: dict, engine, agent: int, package=None, samp_lvl=None):
        if self.last_step is not None:
            self.last_step.invalidate()
        step = PlanStepResult(engine, inputs, res, agent)
        step.package = package
        lr = self.params_harm["log_reg"]["ignore_angle"]
        step.harm_coeff = (lr["const"], lr["speed"])
        self.last_step = step
        self._total_count = res["n_candidates"]
        hist = list(res["reason_hist"]) if self._kinematic_debug else [0] * 11
        hist[0] = res["n_infeasible"]  # reactive_planner.py:233-234
        self._infeasible_count_kinematics = hist
        self.infeasible_kinematics_percentage = res["feasible_percentage"]
        self._collision_counter = res["n_collisions"]
        if self._draw_traj_set or self.save_all_traj:
            self.all_traj = _LazySortedList(step)
        if self.occlusion_module is not None:
            return self._occlusion_walk(step, samp_lvl)
        best = step.best
        if best is None:
            return self._fallback(step, samp_lvl)
        if self.road_boundary_check is None:
            return best
        # host-side walk over the GPU's survivors for checks that stay on the host (planner.py:362-390)
        _, idx = engine.topk(self.config.survivors)
        for g in idx[agent]:
            if g < 0:
                break
            cand = step.sample(int(g) - inputs.shard_begin)
            harm = self.road_boundary_check(cand)
            cand.boundary_harm = harm
            cand._coll_detected = False
            if harm == 0:
                return cand
        return self._fallback(step, samp_lvl)

    def _occlusion_walk(self, step, samp_lvl):
        """trajectories.py:557-560 + planner.py:329-392 with an occlusion module: the module adds its costs to the feasible
        trajectories (creation order), the list is sorted stably, and the first candidate that neither collides nor leaves the road
        AND passes the module's safety assessment is the step's trajectory."""
        ids = np.nonzero(step.mask(_abi.FX_FLAG_COSTED))[0]
        trajs = step.samples(ids)
        if trajs:
            self.occlusion_module.calc_costs(trajs)
            trajs.sort(key=lambda t: t.cost)
        self._collision_counter = 0
        for cand in trajs:
            if cand.valid is False:              # planner.py:338-339: invalidated by the module's calc_costs (harm above its limit)
                continue
            if cand._coll_detected is None:      # kept for drawing / debugging only: not part of the collision walk
                continue
            if cand._coll_detected:
                self._collision_counter += 1
                continue
            harm = self.road_boundary_check(cand) if self.road_boundary_check is not None else (cand.boundary_harm or 0)
            cand.boundary_harm = harm
            if harm != 0:
                continue
            _, ok = self.occlusion_module.trajectory_safety_assessment(cand)
            if ok is not None and not ok:    # (the reference tests `is False`; a NumPy False from the module is a veto too)
                continue
            return cand
        return self._fallback(step, samp_lvl)

    # ------------------------------------------------------------------ standstill (reactive_planner.py:579-626)
    def _compute_standstill_trajectory(self) -> StandstillSample:
        x_0 = self.x_0
        x_0_lon, x_0_lat = self.x_cl
        cs = self.coordinate_system
        N = self.N
        kappa_0 = np.tan(x_0.steering_angle) / self.vehicle_params.wheelbase
        a = np.repeat(0.0, N)
        a[1] = -x_0.velocity / self.dT
        cart = CartesianSample(np.repeat(x_0.position[0], N), np.repeat(x_0.position[1], N), np.repeat(x_0.orientation, N),
                               np.repeat(0.0, N), a, np.repeat(kappa_0, N), np.repeat(0.0, N), current_time_step=N)
        s_idx = int(np.argmax(cs.ref_pos > x_0_lon[0])) - 1
        ref_theta = np.unwrap(cs.ref_theta)
        theta_cl = x_0.orientation - interpolate_angle(x_0_lon[0], cs.ref_pos[s_idx], cs.ref_pos[s_idx + 1],
                                                       ref_theta[s_idx], ref_theta[s_idx + 1])
        curv = CurviLinearSample(np.repeat(x_0_lon[0], N), np.repeat(x_0_lat[0], N), np.repeat(theta_cl, N),
                                 dd=np.repeat(x_0_lat[1], N), ddd=np.repeat(x_0_lat[2], N), ss=np.repeat(x_0_lon[1], N),
                                 sss=np.repeat(x_0_lon[2], N), current_time_step=N)
        T = self.horizon
        lon = _quartic(x_0_lon[0], x_0_lon[1], x_0_lon[2], T, 0.0, 0.0)
        lat = _quintic(x_0_lat[0], x_0_lat[1], x_0_lat[2], x_0_lat[0], 0.0, 0.0, T)
        names = sorted(n for n, w in self.cost_weights.items() if w != 0)
        return StandstillSample(self.horizon, self.dT, cart, curv, PolynomialView(lon, T), PolynomialView(lat, T), names)

    # ------------------------------------------------------------------ output packaging (planner.py:394-447)
    def _compute_trajectory_pair(self, trajectory) -> tuple:
        pkg = getattr(trajectory, "_pkg", None)
        if pkg is not None:
            # packaged by the library (fx_read_package): yaw rate, steering angle and shifted heading are rows of the block;
            # the state objects are built when they are indexed
            b = pkg.block
            n, t0 = b.shape[1], self.x_0.time_step
            YR, ST, OR = _abi.PKG_ROW_YAW_RATE, _abi.PKG_ROW_STEERING, _abi.PKG_ROW_ORIENTATION
            cols = {}

            def col(i):   # one column of the block as floats, shared by the four lists (a closed-loop step reads one or two)
                c = cols.get(i)
                if c is None:
                    c = cols[i] = b[:, i].tolist()
                return c

            def cart(i):
                c = col(i)
                return ReactivePlannerState(t0 + i, np.array((c[0], c[1])), c[OR], c[3], c[4], c[YR], c[ST])

            def curv(i):
                c = col(i)
                return dict(time_step=t0 + i, position=np.array((c[7], c[8])), velocity=c[3], acceleration=c[4], orientation=c[2],
                            yaw_rate=c[5])

            def lon(i):   # (s, s', s''): x_cl of the next cycle is entry 1 (+ replanning counter)
                c = col(i)
                return [c[7], c[10], c[11]]

            def lat(i):   # (d, d', d'')
                c = col(i)
                return [c[8], c[12], c[13]]

            states = _LazyStates(n, cart)
            states._rows_src = (b, _ROWS_XYOV)   # [n][x, y, orientation, velocity]: what a batch of agents shares as predictions
            return (states, _LazyStates(n, curv), _LazyStates(n, lon), _LazyStates(n, lat))
        c, k = trajectory.cartesian, trajectory.curvilinear
        n = len(c.x)
        theta = np.asarray(c.theta, dtype=np.float64)
        yaw_rate = np.empty(n)
        yaw_rate[0] = self.x_0.yaw_rate
        yaw_rate[1:] = (theta[1:] - theta[:-1]) / self.dT
        steer = np.arctan2(self.vehicle_params.wheelbase * np.asarray(c.kappa, dtype=np.float64), 1.0)
        # shift_orientation (planner.py:536-542): into [x_0.orientation - pi, x_0.orientation + pi]
        lo, hi = self.x_0.orientation - np.pi, self.x_0.orientation + np.pi
        orient = theta.copy()
        for _ in range(4):
            orient = np.where(orient < lo, orient + 2 * np.pi, orient)
            orient = np.where(orient > hi, orient - 2 * np.pi, orient)
        pos = np.stack([np.asarray(c.x, dtype=np.float64), np.asarray(c.y, dtype=np.float64)], axis=1)
        sd = np.stack([np.asarray(k.s, dtype=np.float64), np.asarray(k.d, dtype=np.float64)], axis=1)
        t0 = self.x_0.time_step
        v, a, kap = np.asarray(c.v).tolist(), np.asarray(c.a).tolist(), np.asarray(c.kappa).tolist()
        th, yr, st, orl = theta.tolist(), yaw_rate.tolist(), steer.tolist(), orient.tolist()
        cart_list = [ReactivePlannerState(t0 + i, pos[i], orl[i], v[i], a[i], yr[i], st[i]) for i in range(n)]
        cl_list = [dict(time_step=t0 + i, position=sd[i], velocity=v[i], acceleration=a[i], orientation=th[i], yaw_rate=kap[i])
                   for i in range(n)]
        lon_list = np.stack([k.s, k.s_dot, k.s_ddot], axis=1).tolist()
        lat_list = np.stack([k.d, k.d_dot, k.d_ddot], axis=1).tolist()
        return cart_list, cl_list, lon_list, lat_list

    def _compute_cart_traj(self, trajectory) -> list:
        """planner.py:449-486: the Cartesian state list the collision walk builds per candidate -- yaw rate is the CENTRAL
        difference np.gradient(theta) / dT here (x_0.yaw_rate at 0), and the heading is not shifted."""
        c = trajectory.cartesian
        theta = np.asarray(c.theta, dtype=np.float64)
        yaw_rate = np.gradient(theta) / self.dT
        yaw_rate[0] = self.x_0.yaw_rate
        steer = np.arctan2(self.vehicle_params.wheelbase * np.asarray(c.kappa, dtype=np.float64), 1.0)
        pos = np.vstack((c.x, c.y)).T
        t0 = self.x_0.time_step
        v, a = np.asarray(c.v).tolist(), np.asarray(c.a).tolist()
        return [ReactivePlannerState(t0 + i, pos[i], th, v[i], a[i], yr, st)
                for i, (th, yr, st) in enumerate(zip(theta.tolist(), yaw_rate.tolist(), steer.tolist()))]

    @staticmethod
    def shift_orientation(state_list, interval_start=-np.pi, interval_end=np.pi):
        for state in state_list:
            while state.orientation < interval_start:
                state.orientation += 2 * np.pi
            while state.orientation > interval_end:
                state.orientation -= 2 * np.pi
        return state_list

    def record_state_and_input(self, state: ReactivePlannerState):
        """planner.py:245-262: recorded states and the control inputs (acceleration, steering-angle speed) between them."""
        self.record_state_list.append(state)
        rate = ((state.steering_angle - self.record_state_list[-2].steering_angle) / self.dT
                if len(self.record_state_list) > 1 else 0.0)
        self.record_input_list.append(dict(time_step=state.time_step, acceleration=state.acceleration,
                                           steering_angle_speed=rate))

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None


_ROWS_XYOV = [0, 1, _abi.PKG_ROW_ORIENTATION, 3]   # x, y, shifted heading, velocity of the package block


class _LazyStates:
    """Read-only sequence whose items are built on first access and then kept (the state lists of a trajectory pair: a
    closed-loop step reads one or two of the 31 states)."""

    _rows_src = None   # optional (block, row indices): the same states as an [n][4] array (x, y, orientation, velocity)
    _rows = None

    def __init__(self, n: int, make):
        self._make = make
        self._items = [None] * n

    @property
    def rows(self):
        if self._rows is None and self._rows_src is not None:
            self._rows = self._rows_src[0][self._rows_src[1]].T
        return self._rows

    def __eq__(self, other):
        if isinstance(other, (list, tuple, _LazyStates)):
            return len(other) == len(self) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def __len__(self):
        return len(self._items)

    def _get(self, i):
        it = self._items[i]
        if it is None:
            it = self._items[i] = self._make(i if i >= 0 else i + len(self._items))
        return it

    def __getitem__(self, j):
        if j.__class__ is int:
            it = self._items[j]
            if it is None:
                it = self._items[j] = self._make(j if j >= 0 else j + len(self._items))
            return it
        if isinstance(j, slice):
            return [self._get(i) for i in range(*j.indices(len(self._items)))]
        return self._get(j)

    def __iter__(self):
        return (self._get(i) for i in range(len(self._items)))


class _LazySortedList:
    """`all_traj` (reactive_planner.py:245-247): every returned trajectory in stable cost order, created on
    demand instead of as 10^4..10^6 Python objects."""

    def __init__(self, step: PlanStepResult):
        self._step = step
        self._ids = None

    def _order(self):
        if self._ids is None:
            self._ids = self._step.sorted_ids(_abi.FX_FLAG_COSTED)
        return self._ids

    def __len__(self):
        return len(self._order())

    def __getitem__(self, j):
        ids = self._order()
        if isinstance(j, slice):
            return [self._step.sample(int(g)) for g in ids[j]]
        return self._step.sample(int(ids[j]))

    def __iter__(self):
        for g in self._order():
            yield self._step.sample(int(g))


class _LazySamples:
    """A fixed list of candidates of one step as TrajectorySample views, created on demand (the fallback selector's argument:
    the feasible trajectories in creation order)."""

    def __init__(self, step: PlanStepResult, ids):
        self._step, self._ids = step, np.asarray(ids)

    def __len__(self):
        return len(self._ids)

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self._step.sample(int(g)) for g in self._ids[j]]
        return self._step.sample(int(self._ids[j]))

    def __iter__(self):
        for g in self._ids:
            yield self._step.sample(int(g))


def _quartic(xs, vxs, axs, T, vxe, axe):
    b1 = vxe - vxs - axs * T
    b2 = axe - axs
    return np.array([xs, vxs, .5 * axs, (3.0 * b1 - T * b2) / (3.0 * T * T), (T * b2 - 2.0 * b1) / (4.0 * T * T * T), 0.0])


def _quintic(xs, vxs, axs, xe, vxe, axe, T):
    T2, T3, T4, T5 = T * T, T ** 3, T ** 4, T ** 5
    b0 = xe - xs - vxs * T - .5 * axs * T2
    b1 = vxe - vxs - axs * T
    b2 = axe - axs
    return np.array([xs, vxs, .5 * axs, (10.0 * b0 - 4.0 * b1 * T + .5 * b2 * T2) / T3,
                     (-15.0 * b0 + 7.0 * b1 * T - b2 * T2) / T4, (6.0 * b0 - 3.0 * b1 * T + .5 * b2 * T2) / T5])
