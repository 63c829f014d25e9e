"""FrenetPlannerInterfaceHip -- the per-agent glue around the planner: reference path, desired velocity, replanning
counter and state hand-over.

Mirrors cr_scenario_handler/planner_interfaces/frenet_interface.py:33-287 (`FrenetPlannerInterface`) and
cr_scenario_handler/utils/velocity_planner.py:6-168 (`VelocityPlanner`) for the hot path's callers:

    __init__           route centre line -> extend_ref_path_both_ends -> smooth_ref_path (:104-116), x_0 from the
                       planning problem's initial state shifted to the rear axle (state.py:41-75), planner externals
    update_planner     desired velocity from the goal distance / remaining time, predictions hand-over (:172-205)
    step_interface     replan every `replanning_frequency` steps, otherwise advance along the stored trajectory;
                       x_0 <- state 1 of the chosen trajectory, x_cl <- (lon_list[1], lat_list[1]) (:207-287)

`begin_step / finish_step` split step_interface around the plan step so that multiagent.AgentBatchHip can put the
plan steps of many agents into ONE launch.  Behaviour planner, occlusion module and goal-reached bookkeeping of the
reference's agent (agent.py) stay outside (SURVEY.md 8 "out").
"""
import copy
import logging
import time
from typing import Optional

import numpy as np

from . import ref_path
from .commonroad_xml import PlanningProblem, Scenario
from .problem import VehicleParams
from .reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState


def create_from_initial_state(initial_state, wheelbase: float, wb_rear_axle: float) -> ReactivePlannerState:
    """state.py:41-75: centre position -> rear axle, steering angle atan2(wheelbase * yaw_rate, velocity)."""
    o = float(initial_state.orientation)
    pos = np.asarray(initial_state.position, dtype=np.float64) - wb_rear_axle * np.array([np.cos(o), np.sin(o)])
    yaw_rate = float(getattr(initial_state, "yaw_rate", 0.0) or 0.0)
    v = float(initial_state.velocity)
    return ReactivePlannerState(time_step=int(initial_state.time_step), position=pos, orientation=o, velocity=v,
                                acceleration=float(getattr(initial_state, "acceleration", 0.0) or 0.0), yaw_rate=yaw_rate,
                                steering_angle=float(np.arctan2(wheelbase * yaw_rate, v)))


class VelocityPlanner:
    """velocity_planner.py:6-168 on the stdlib scenario types: desired velocity = remaining arc length to the goal /
    remaining time, clipped to +-5 m/s around the current velocity."""

    def __init__(self, scenario: Scenario, planning_problem: PlanningProblem, coordinate_system):
        self.scenario, self.planning_problem, self.coordinate_system = scenario, planning_problem, coordinate_system
        self.DT = scenario.dt
        goal = planning_problem.goals[0] if planning_problem.goals else None
        self.goal = goal
        self.default_goal_velocity = None
        if goal is not None and goal.velocity_interval is not None:  # :18-24
            lo, hi = max(goal.velocity_interval[0], 0.01), max(goal.velocity_interval[1], 0.01)
            self.default_goal_velocity = (lo + hi) / 2
        self.used_goal_metric, self.goal_centers, self.goal_lanelets, self.goal_rects = None, [], [], []
        if goal is not None and goal.lanelet_ids:                    # :52-64
            self.used_goal_metric = "lanelets_of_goal_position"
            for lid in goal.lanelet_ids:
                c = scenario.lanelets[lid].center_vertices
                self.goal_centers.append(c[int(len(c) / 2.0)])
            self.goal_lanelets = [goal.lanelet_ids[0]]
        elif goal is not None and goal.rectangles:                   # :36-40
            self.used_goal_metric = "center"
            self.goal_centers.append(np.asarray(goal.rectangles[0]["center"], dtype=np.float64))
            self.goal_rects = [goal.rectangles[0]]
        elif goal is not None and goal.time_interval is not None:    # :42-43
            self.used_goal_metric = "time_step"
        self.goal_s_position = None
        if self.used_goal_metric != "time_step":                     # :66-80
            for g in self.goal_centers:
                try:
                    s = float(coordinate_system.convert_to_curvilinear_coords(g[0], g[1])[0])
                except ValueError:
                    s = None
                if s is None or self.goal_s_position is None or s < self.goal_s_position:
                    self.goal_s_position = s

    @staticmethod
    def clip_velocity(v_theory, v_now, max_value=50, clip_value=5):
        return max(min(v_theory, min(v_now + clip_value, max_value)), max(v_now - clip_value, 0))

    def _is_in_goal(self, x_0) -> bool:
        p = x_0.position
        if self.goal_lanelets:
            return self.scenario.lanelets[self.goal_lanelets[0]].contains(p)
        for r in self.goal_rects:
            c, o = np.asarray(r["center"]), float(r.get("orientation", 0.0))
            e = np.asarray(p) - c
            lx, ly = e[0] * np.cos(o) + e[1] * np.sin(o), -e[0] * np.sin(o) + e[1] * np.cos(o)
            if abs(lx) < r["length"] / 2 and abs(ly) < r["width"] / 2:
                return True
        return False

    def calc_remaining_time_steps(self, ego_state_time: float, t: float) -> Optional[int]:
        step = int(ego_state_time + t / self.DT)                     # :158-168
        if self.goal is not None and self.goal.time_interval is not None:
            lo, hi = self.goal.time_interval
            return int(((hi - step) + (lo - step)) / 2)
        return None

    def calculate_desired_velocity(self, x_0, s_position) -> float:
        if self.used_goal_metric != "time_step" and self._is_in_goal(x_0):      # :104-109
            return self.clip_velocity(self.default_goal_velocity, x_0.velocity) if self.default_goal_velocity else x_0.velocity
        if self.used_goal_metric == "time_step":
            return x_0.velocity
        if not self.goal_s_position and self.default_goal_velocity:
            return self.clip_velocity(self.default_goal_velocity, x_0.velocity)
        if not self.goal_s_position:
            return x_0.velocity
        distance_to_goal = self.goal_s_position - s_position
        steps = self.calc_remaining_time_steps(x_0.time_step, 0.0)
        remaining_time = round((steps or 0) * self.DT, 3)
        if remaining_time > 0.0:
            return self.clip_velocity(distance_to_goal / remaining_time, x_0.velocity)
        if self.default_goal_velocity:
            return self.clip_velocity(self.default_goal_velocity, x_0.velocity)
        return x_0.velocity


class FrenetPlannerInterfaceHip:
    def __init__(self, agent_id: int, scenario: Scenario, planning_problem: PlanningProblem,
                 config: Optional[PlannerConfig] = None, vehicle: Optional[VehicleParams] = None, engine=None,
                 device: int = 0, msg_logger=None, reference_path: Optional[np.ndarray] = None, use_road_boundary: bool = False):
        self.id = agent_id
        self.scenario, self.planning_problem = scenario, planning_problem
        self.config_plan = config or PlannerConfig()
        self.DT = self.config_plan.dt
        self.replanning_counter = 0
        self.replanning_traj = None
        self.msg_logger = msg_logger or logging.getLogger("Message_logger_" + str(agent_id))
        self.planner = ReactivePlannerHip(self.config_plan, vehicle, engine=engine, msg_logger=self.msg_logger, device=device)
        veh = self.planner.vehicle_params
        self.x_0 = create_from_initial_state(planning_problem.initial_state, veh.wheelbase, veh.wb_rear_axle)
        self.planner.record_state_and_input(self.x_0)
        if reference_path is None:  # :104-116
            # commonroad-route-planner hands out a densely resampled polyline (smooth_ref_path assumes 0.125 m between
            # vertices, utils_coordinate_system.py:117); the lanelet centre lines are resampled to that spacing first
            route = ref_path.resample_polyline(scenario.route_reference_path(planning_problem), 0.125)
            reference_path = ref_path.prepare_reference_path(route)
        self.reference_path = np.asarray(reference_path, dtype=np.float64)
        self.x_cl = None
        self.desired_velocity = None
        if use_road_boundary:
            self.planner.set_road_boundary(scenario.road_boundary_segments())
        self.planner.update_externals(x_0=self.x_0, reference_path=self.reference_path)
        self.x_cl = self.planner.x_cl
        self.velocity_planner = VelocityPlanner(scenario, planning_problem, self.coordinate_system)
        self._t0 = None

    # -- views (frenet_interface.py:140-170) --
    @property
    def all_trajectories(self):
        return self.planner.all_traj

    @property
    def record_state_list(self):
        return self.planner.record_state_list

    @property
    def record_input_list(self):
        return self.planner.record_input_list

    @property
    def vehicle_history(self):
        return self.planner.ego_vehicle_history

    @property
    def coordinate_system(self):
        return self.planner.coordinate_system

    @property
    def optimal_trajectory(self):
        return self.planner.optimal_trajectory

    @property
    def trajectory_pair(self):
        return self.planner.trajectory_pair

    def needs_plan(self) -> bool:
        """True when step_interface will run a plan step (replanning counter 0 or wrapped, :231-234)."""
        f = self.config_plan.replanning_frequency
        c = 0 if int(self.replanning_counter / f) == 1 else self.replanning_counter
        return c == 0 or f < 2

    def update_planner(self, scenario: Optional[Scenario], predictions: dict):
        """frenet_interface.py:172-205 without a behaviour planner."""
        if scenario is not None:
            self.scenario = scenario
        self.desired_velocity = self.velocity_planner.calculate_desired_velocity(self.x_0, self.x_cl[0][0])
        self.planner.update_step(self.x_0, self.x_cl, self.desired_velocity, predictions)

    # -- one step (frenet_interface.py:207-287), split around the plan step --
    def _plans_now(self) -> bool:
        if int(self.replanning_counter / self.config_plan.replanning_frequency) == 1:
            self.replanning_counter = 0
        return self.replanning_counter == 0 or self.config_plan.replanning_frequency < 2

    def begin_step(self):
        """Returns the PlanInputs of this step's first sampling level, or None when the step only advances along the
        stored trajectory."""
        if self._plans_now():
            self._t0 = time.time()
            return self.planner.plan_begin()
        return None

    def finish_step(self, pair, current_timestep=None):
        """pair: the planner's trajectory pair of this step (ignored on non-planning steps)."""
        planner = self.planner
        if self.replanning_counter == 0 or self.config_plan.replanning_frequency < 2:
            if not pair:
                self.msg_logger.critical("No Kinematic Feasible and Optimal Trajectory Available!")
                return None, self.replanning_counter
            k = 1
            self.replanning_traj = pair
            selected = pair[0]
        else:
            k = 1 + self.replanning_counter
            pair = selected = self.replanning_traj
        state = pair[0][k]
        planner.record_state_and_input(state)
        # (copy.deepcopy(record_state_list[-1]) in the reference, :262,274: the state's own __deepcopy__, without the generic walk)
        self.x_0 = state.__deepcopy__() if state.__class__ is ReactivePlannerState else copy.deepcopy(state)
        self.x_cl = (pair[2][k], pair[3][k])
        if k > 1:
            planner.plan_postprocessing(planner.optimal_trajectory, 0.0, replanning_counter=self.replanning_counter)
        if self.msg_logger.isEnabledFor(logging.INFO):   # (three formatted strings per agent and step otherwise)
            self.msg_logger.info("current time step: %s", current_timestep)
            self.msg_logger.info("current velocity: %s", self.x_0.velocity)
            self.msg_logger.info("current target velocity: %s", self.desired_velocity)
        self.replanning_counter += 1
        return selected, self.replanning_counter - 1

    def step_interface(self, current_timestep=None):
        pair = self.planner.plan() if self._plans_now() else None
        return self.finish_step(pair, current_timestep)

    def close(self):
        self.planner.close()
