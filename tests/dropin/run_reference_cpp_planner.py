"""Drop-in check (runs only where /root/reference exists): the REFERENCE's own `ReactivePlannerCpp`
(frenetix_motion_planner/reactive_planner_cpp.py, unmodified, imported from /root/reference) driving THIS package's
`frenetix` module (frenetix_compat.install()).  Its constructor body, setters and `plan()` run as written: functor
registration with the reference's keyword arguments, PoseWithCovariance / PredictedObject construction,
compute_initial_state, the C x 13 sampling matrix, evaluation, the sorted-trajectory split, the collision walk of
planner.py:329-392, feasabilityMap statistics, standstill / stopping fallbacks and _compute_trajectory_pair.

What is substituted, and why:
  * third-party packages that are not installed (commonroad-io, commonroad-drivability-checker, omegaconf, ...) are
    served by the stub finder of tests/golden/ref_harness.py; the CommonRoad state / trajectory containers get the
    small real shims below (plain dataclasses);
  * the two pycrcc calls inside trajectory_collision_check are answered from the handler's own per-trajectory results
    (`_coll_detected`; no road boundary) -- pycrcc is not in the reference tree;
  * Planner.__init__ (reads YAML / JSON configuration, builds the road boundary with commonroad_dc) is bypassed with
    object.__new__ and the attributes it would set; ReactivePlannerCpp's own constructor body is executed verbatim;
  * without a GPU the engine behind the handler is the oracle-backed stand-in (tests/oracle_engine.py); pass --hip to
    use the HIP engine.
Prints one JSON line."""
import dataclasses
import json
import logging
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_harness  # noqa: E402


def install_container_shims():
    import commonroad.scenario  # noqa: F401  (stub package)

    @dataclasses.dataclass
    class KSState:
        time_step: int = 0
        position: np.ndarray = None
        orientation: float = 0.0
        velocity: float = 0.0
        steering_angle: float = 0.0

        def translate_rotate(self, translation, angle):
            return dataclasses.replace(self, position=np.asarray(self.position) + np.asarray(translation),
                                       orientation=self.orientation + angle)

    class _Bag:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class InitialState(_Bag):
        pass

    class CustomState(_Bag):
        pass

    class InputState(_Bag):
        pass

    class Trajectory:
        def __init__(self, initial_time_step, state_list):
            self.initial_time_step, self.state_list = initial_time_step, list(state_list)

        @property
        def final_state(self):
            return self.state_list[-1]

    st = ref_harness._StubModule("commonroad.scenario.state")   # unknown names still resolve to the permissive dummy
    st.KSState, st.InitialState, st.CustomState, st.InputState, st.FloatExactOrInterval = KSState, InitialState, CustomState, InputState, float
    tr = ref_harness._StubModule("commonroad.scenario.trajectory")
    tr.Trajectory = Trajectory
    sys.modules["commonroad.scenario.state"] = st
    sys.modules["commonroad.scenario.trajectory"] = tr
    sys.modules["commonroad.scenario"].state = st
    sys.modules["commonroad.scenario"].trajectory = tr


class O:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def main(use_hip: bool, blocked: bool, record: str = None):
    ref_harness.install()
    install_container_shims()
    from frenetix_motion_planner_amd import VehicleParams, frenetix_compat, synthetic
    frenetix_compat.install(force=True)
    recorder = None
    if record:  # write the adapter's call trace (data only) for the replay test on the GPU box
        from tests.dropin.trace_recorder import CLASSES, Recorder
        recorder = Recorder()
        recorder.wrap({n: sys.modules[n] for n in CLASSES})
    import frenetix
    import frenetix_motion_planner.planner as refplanner
    import frenetix_motion_planner.reactive_planner_cpp as rpc
    from frenetix_motion_planner.sampling_matrix import SamplingHandler

    veh = VehicleParams()
    p = object.__new__(rpc.ReactivePlannerCpp)
    # ---- what Planner.__init__ (planner.py:50-163) would have set
    p.config_plan = O(planning=O(dt=0.1, planning_horizon=3.0, t_min=1.1, emergency_mode="stopping"),
                      debug=O(multiproc=False, num_workers=1), cost=O(cost_weights=None))
    p.config_sim = O(simulation=O(use_multiagent=False, multiprocessing=False, ego_agent_id=42), vehicle=veh)
    p.horizon, p.dT, p.N = 3.0, 0.1, 30
    p.vehicle_params = O(length=veh.length, width=veh.width, wheelbase=veh.wheelbase, wb_rear_axle=veh.wb_rear_axle,
                         a_max=veh.a_max, v_switch=veh.v_switch, delta_max=veh.delta_max, v_delta_max=veh.v_delta_max,
                         v_max=veh.v_max)
    p._low_vel_mode_threshold = 2.0
    p.msg_logger = logging.getLogger("dropin")
    p.msg_logger.setLevel(logging.CRITICAL + 1)
    p.cost_weights = {"distance_to_reference_path": 5.0, "lateral_jerk": 0.2, "longitudinal_jerk": 0.2, "prediction": 0.2,
                      "velocity_offset": 1.0}
    p.logger = p.behavior = p.x_cl = p.set_new_ref_path = p.all_traj = p.road_boundary = p.occlusion_module = None
    p.use_occ_model = p.log_risk = p.save_all_traj = False
    p._draw_traj_set = True
    p.scenario = O(obstacles=[])
    p.ego_vehicle_history, p.record_state_list, p.record_input_list = [], [], []
    p._collision_counter = 0
    p._sampling_min, p._sampling_max = 2, 3
    p.sampling_handler = SamplingHandler(dt=0.1, max_sampling_number=3, t_min=1.1, horizon=3.0, delta_d_max=3.0,
                                         delta_d_min=-3.0, d_ego_pos=False)
    # ---- ReactivePlannerCpp.__init__ body, verbatim (reactive_planner_cpp.py:43-54)
    p.predictionsForCpp = {}
    p.handler = frenetix.TrajectoryHandler(dt=p.config_plan.planning.dt)
    if not use_hip:
        from tests.oracle_engine import OracleEngine
        p.handler._engine = OracleEngine()
    p.trajectory_handler_set_constant_cost_functions()
    p.trajectory_handler_set_constant_feasibility_functions()
    frenetix._frenetix.setup_logger(p.msg_logger)
    # ---- update_externals (planner.py:172-217) through the reference's setters
    ref = synthetic.reference_polyline("straight" if blocked else "arc", 400, 0.5, 0.01)
    p.coordinate_system_cpp = frenetix.CoordinateSystemWrapper(ref)  # set_reference_and_coordinate_system :186-192
    p.coordinate_system = p.coordinate_system_cpp
    p.set_new_ref_path = True
    cs = p.coordinate_system_cpp
    s0 = float(cs.ref_pos[40] + 0.1)
    from frenetix_motion_planner.state import ReactivePlannerState
    x0 = ReactivePlannerState(time_step=0, position=np.asarray(cs.convert_to_cartesian_coords(s0, 0.2)),
                              orientation=float(cs.ref_theta[40]), velocity=10.0, steering_angle=0.0, acceleration=0.0,
                              yaw_rate=0.0)
    refplanner.Planner.set_x_0(p, x0)
    refplanner.Planner.set_x_cl(p, None)           # -> _compute_initial_states -> frenetix.compute_initial_state
    if blocked:   # one wide obstacle parked across the lane: every feasible candidate collides
        n = 31
        preds = {5: dict(pos_list=np.tile([[s0 + 9.0, 0.0]], (n, 1)), cov_list=np.tile(np.eye(2) * 0.1, (n, 1, 1)),
                         orientation_list=np.full(n, np.pi / 2), shape=dict(length=14.0, width=3.0))}
    else:
        preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
    p.set_predictions(preds)                       # PoseWithCovariance / PredictedObject (:56-86)
    refplanner.Planner.set_desired_velocity(p, 12.0, x0.velocity)
    # pycrcc is not in the reference tree: answer its two calls from the handler's per-trajectory results
    refplanner.collision_check_prediction = lambda predictions, scenario, ego_co, frenet_traj, time_step: bool(frenet_traj._coll_detected)
    refplanner.trajectories_collision_static_obstacles = lambda **kw: [-1]
    refplanner.trajectory_preprocess_obb_sum = lambda obj: (obj, False)   # create_coll_object's pycrcc preprocessing (:528)
    # ---- the reference's plan()
    pair = p.plan()
    opt = p.optimal_trajectory
    out = dict(planned=pair is not None, n_matrix=int(p._generate_sampling_matrix(2).shape[0]),
               optimal_id=None if opt is None else int(opt.uniqueId), optimal_cost=None if opt is None else float(opt.cost),
               collisions=int(p.infeasible_count_collision), feasible_percentage=float(p.infeasible_kinematics_percentage),
               infeasible_hist=[float(v) for v in p._infeasible_count_kinematics],
               all_traj=None if p.all_traj is None else len(p.all_traj),
               x_cl=[list(map(float, p.x_cl[0])), list(map(float, p.x_cl[1]))])
    if pair is not None:
        cart = pair[0].state_list
        out.update(n_states=len(cart), x_cl_next=[list(map(float, pair[2][1])), list(map(float, pair[3][1]))],
                   first=[float(cart[0].position[0]), float(cart[0].position[1]), float(cart[0].orientation)],
                   last=[float(cart[-1].position[0]), float(cart[-1].position[1]), float(cart[-1].velocity)],
                   sampling_parameters=[float(v) for v in opt.sampling_parameters],
                   costmap={k: [float(v[0]), float(v[1])] for k, v in opt.costMap.items()})
    if recorder is not None:
        recorder.save(record, out)
    print(json.dumps(out))


if __name__ == "__main__":
    rec = sys.argv[sys.argv.index("--record") + 1] if "--record" in sys.argv else None
    main("--hip" in sys.argv, "--blocked" in sys.argv, rec)
