"""The planner front-end and the frenetix-shaped handler end to end on a real MI355X."""
import sys

import numpy as np
import pytest

from frenetix_motion_planner_amd import VehicleParams, _abi, synthetic

pytestmark = pytest.mark.gpu


def make_planner(v0=10.0, **cfg):
    from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
    rp = ReactivePlannerHip(PlannerConfig(**cfg), VehicleParams())
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs_tmp = synthetic.CoordinateSystem(ref)
    s0, d0 = float(cs_tmp.ref_pos[40] + 0.1), 0.2
    xy = cs_tmp.convert_to_cartesian_coords(s0, d0)
    x0 = ReactivePlannerState(time_step=0, position=xy, orientation=float(cs_tmp.ref_theta[40]), velocity=v0)
    preds = synthetic.synthetic_predictions(cs_tmp, 5, 30, 0.1, s0, np.random.default_rng(1))
    rp.update_externals(reference_path=ref, x_0=x0, desired_velocity=12.0, predictions=preds)
    return rp, x0


def test_plan_matches_oracle_and_packages_output():
    from oracle import oracle
    rp, x0 = make_planner()
    pair = rp.plan()
    assert pair is not None
    cart, cl, lon, lat = pair
    assert len(cart) == 31 and len(lon) == 31 and len(lat[0]) == 3
    best = rp.optimal_trajectory
    # the same plan step through the oracle
    inp = rp._inputs_for_level(2)
    from tests.test_hip_parity import hip_hulls  # noqa: F401
    ref_inp = rp._inputs_for_level(2)
    ref_inp.obstacles = synthetic.pack_predictions(rp.predictions, 31, oracle.build_obstacle_hulls)
    out = oracle.plan_step(ref_inp)
    assert best.uniqueId == out["result"]["best_index"]
    assert abs(best.cost - out["result"]["best_cost"]) < 1e-9 * max(1, abs(best.cost))
    assert rp.infeasible_count_collision == out["result"]["n_collisions"]
    assert rp._infeasible_count_kinematics[0] == out["result"]["n_infeasible"]
    assert abs(rp.infeasible_kinematics_percentage - out["result"]["feasible_percentage"]) < 1e-9
    g = best.uniqueId
    assert np.allclose(best.cartesian.x, out["planes"][g][0], atol=1e-9)
    assert np.allclose([st.position[1] for st in cart], out["planes"][g][1], atol=1e-9)
    assert np.allclose(lon, out["planes"][g][[7, 10, 11]].T, atol=1e-9)
    assert cart[0].yaw_rate == x0.yaw_rate and cart[3].time_step == 3
    # orientations are wrapped into x0.orientation +- pi (planner.py:536-542)
    assert all(abs(st.orientation - x0.orientation) <= np.pi for st in cart)
    # costMap / feasabilityMap / sampling_parameters surface
    assert set(best.costMap) == set(inp.cost_names)
    raw, weighted = best.costMap["distance_to_reference_path"]
    assert abs(weighted - 5.0 * raw) < 1e-12
    assert set(best.feasabilityMap) == {"Curvature Constraint", "Yaw rate Constraint", "Curvature Rate Constraint",
                                        "Acceleration Constraint"}
    assert best.sampling_parameters.shape == (13,) and best.feasible and best.valid
    # all_traj: sorted by cost, lazily materialised
    assert len(rp.all_traj) == out["result"]["n_returned"]
    costs = [t.cost for t in rp.all_traj[:50]]
    assert costs == sorted(costs)
    # x_cl hand-over for the next cycle (frenet_interface.py:255): lon_list[1] / lat_list[1]
    rp.update_externals(x_0=cart[1], x_cl=(lon[1], lat[1]), desired_velocity=12.0)
    assert rp.x_cl == (lon[1], lat[1])
    pair2 = rp.plan()
    assert pair2 is not None
    # the previous optimum stays readable after the device bundle was overwritten
    assert np.allclose(best.cartesian.x, out["planes"][g][0], atol=1e-9)
    rp.close()


def test_sampling_level_escalation_and_standstill():
    # nothing feasible at level 2 (absurd acceleration limit) -> escalates to level 3 -> still nothing -> standstill at v=0
    rp, x0 = make_planner(v0=0.0, sampling_min=2, sampling_max=4)
    rp.vehicle_params.a_max = 1e-9
    pair = rp.plan()
    assert pair is None
    assert rp.optimal_trajectory is not None and rp.optimal_trajectory.uniqueId == 0
    assert len(rp.optimal_trajectory.cartesian.x) == rp.N   # N, not N+1 (reactive_planner.py:608-625)
    assert rp._total_count == 10 * 17 * 18
    rp.close()


def test_road_boundary_callback_walks_survivors():
    rp, _ = make_planner()
    rejected = []

    def check(traj):
        if len(rejected) < 3:
            rejected.append(traj.uniqueId)
            return 0.7
        return 0

    rp.road_boundary_check = check
    rp.plan()
    cost, flags = rp.last_step.cost, rp.last_step.flags
    ok = ((flags & _abi.FX_FLAG_SELECTABLE) != 0) & ((flags & _abi.FX_FLAG_COLLISION) == 0)
    ids = np.nonzero(ok)[0]
    order = ids[np.lexsort((ids, cost[ids]))]
    assert rejected == list(order[:3]) and rp.optimal_trajectory.uniqueId == order[3]
    rp.close()


def test_frenetix_handler_drives_the_engine():
    from frenetix_motion_planner_amd import frenetix_compat as fx
    from frenetix_motion_planner_amd.sampling import SamplingHandler, generate_sampling_matrix, v_sampling_bounds
    from oracle import oracle
    veh = VehicleParams()
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = fx.CoordinateSystemWrapper(ref)
    h = fx.TrajectoryHandler(dt=0.1)
    h.add_feasability_function(fx.CheckYawRateConstraint(deltaMax=veh.delta_max, wheelbase=veh.wheelbase, wholeTrajectory=False))
    h.add_feasability_function(fx.CheckAccelerationConstraint(switchingVelocity=veh.v_switch, maxAcceleration=veh.a_max, wholeTrajectory=False))
    h.add_feasability_function(fx.CheckCurvatureConstraint(deltaMax=veh.delta_max, wheelbase=veh.wheelbase, wholeTrajectory=False))
    h.add_feasability_function(fx.CheckCurvatureRateConstraint(wheelbase=veh.wheelbase, velocityDeltaMax=veh.v_delta_max, wholeTrajectory=False))
    for cls, name, w in ((fx.CalculateLateralJerkCost, "lateral_jerk", 0.2), (fx.CalculateLongitudinalJerkCost, "longitudinal_jerk", 0.2),
                         (fx.CalculateDistanceToReferencePathCost, "distance_to_reference_path", 5.0)):
        h.add_cost_function(cls(name, w))
    h.add_function(fx.FillCoordinates(lowVelocityMode=False, initialOrientation=float(cs.ref_theta[40]), coordinateSystem=cs, horizon=3))
    h.add_cost_function(fx.CalculateVelocityOffsetCost("velocity_offset", 1.0, 12.0, 0.1, 1.1, limit_to_t_min=False, norm_order=2))
    preds = synthetic.synthetic_predictions(cs, 3, 30, 0.1, float(cs.ref_pos[40]), np.random.default_rng(2))
    pobj = {}
    for k, p in preds.items():
        path = [fx.PoseWithCovariance(np.append(p["pos_list"][j], 0.0), np.array([0, 0, np.sin(p["orientation_list"][j] / 2),
                                                                                    np.cos(p["orientation_list"][j] / 2)]),
                                      np.pad(p["cov_list"][j], ((0, 4), (0, 4)))) for j in range(30)]
        pobj[k] = fx.PredictedObject(k, path, p["shape"]["length"], p["shape"]["width"])
    h.add_cost_function(fx.CalculateCollisionProbabilityFast("prediction", 0.2, pobj, veh.length, veh.width, veh.wb_rear_axle))
    # (T u {N dT}) x (V u {s_dot0}) x (D u {d0}) matrix as reactive_planner_cpp.py:228-253 builds it
    sh = SamplingHandler(dt=0.1, max_sampling_number=3, t_min=1.1, horizon=3.0, delta_d_min=-3, delta_d_max=3, d_ego_pos=False)
    sh.set_v_sampling(*v_sampling_bounds(10.0, veh.a_max, 3.0, veh.v_max))
    s0 = float(cs.ref_pos[40] + 0.1)
    t, v, d = sh.ordered_ranges(2, 0.2, cpp_style=True, ss0=10.0, t_full=3.0)
    m = generate_sampling_matrix(t0_range=0.0, t1_range=t, s0_range=s0, ss0_range=10.0, sss0_range=0.0, ss1_range=v,
                                 sss1_range=0, d0_range=0.2, dd0_range=0.0, ddd0_range=0.0, d1_range=d, dd1_range=0.0,
                                 ddd1_range=0.0)
    assert m.shape == (800, 13)
    h.reset_Trajectories()
    h.generate_trajectories(m, False)
    h.evaluate_all_current_functions_concurrent(True)
    trajs = h.get_sorted_trajectories()
    assert len(trajs) == 800
    costs = [tr.cost for tr in trajs]
    assert costs == sorted(costs)
    feasible = [tr for tr in trajs if tr.feasible]
    infeasible = [tr for tr in trajs if not tr.feasible and tr.valid]
    assert feasible and infeasible
    assert all(sum(tr.feasabilityMap.values()) > 0 for tr in infeasible[:20])
    # against the oracle on the same matrix
    step = h._step
    ref_inp = step.inputs
    ref_inp.obstacles = synthetic.pack_predictions(h._predictions(), 31, oracle.build_obstacle_hulls)
    out = oracle.plan_step(ref_inp)
    robust = out["margin"] >= 1e-9
    assert np.array_equal(step.flags[robust], out["flags"][robust])
    best = feasible[0]
    assert np.array_equal(best.sampling_parameters, m[best.uniqueId])
    assert np.allclose(best.curvilinear.s, out["planes"][best.uniqueId][7], atol=1e-9)
    # T = 3.0 rows evaluate the polynomial over the whole horizon (traj_len clamps at N+1)
    full = [tr for tr in feasible if tr.sampling_parameters[1] == 3.0]
    assert full and full[0].actual_traj_length == 31
    h.engine.close()


def test_stop_point_plan_matches_oracle():
    """plan(stop_point_s=...): the stop-point candidate set (reactive_planner.py:628-671) through the engine."""
    from oracle import oracle
    rp, x0 = make_planner(v0=6.0)
    rp.update_externals(desired_velocity=0.0)
    s0 = rp.x_cl[0][0]
    pair = rp.plan(stop_point_s=s0 + 25.0)
    assert pair is not None
    best = rp.optimal_trajectory
    inp = rp._inputs_for_level(2, stop_point_s=s0 + 25.0)
    assert inp.stop_point and inp.as_struct().lon_mode == _abi.FX_LON_STOP_POINT
    # end positions in [(s0 + s_stop) / 2, s_stop], set iteration order
    assert inp.v_samp.min() == pytest.approx(s0 + 12.5) and inp.v_samp.max() == pytest.approx(s0 + 25.0)
    inp.obstacles = synthetic.pack_predictions(rp.predictions, 31, oracle.build_obstacle_hulls)
    out = oracle.plan_step(inp)
    assert best.uniqueId == out["result"]["best_index"]
    assert abs(best.cost - out["result"]["best_cost"]) < 1e-9 * max(1, abs(best.cost))
    g = best.uniqueId
    assert np.allclose(best.curvilinear.s, out["planes"][g][7], atol=1e-9)
    # the winner's longitudinal quintic ends at rest at its sampled position
    c = best.trajectory_long.coeffs if hasattr(best, "trajectory_long") else None
    T = best.sampling_parameters[1]
    i_T = int(round(T / 0.1))
    assert abs(best.curvilinear.s_dot[i_T]) < 1e-6 and abs(best.curvilinear.s_ddot[i_T]) < 1e-5
    # a stop point behind the ego falls back to regular sampling (reactive_planner_cpp.py:263-264,336-341)
    pair2 = rp.plan(stop_point_s=s0 - 5.0)
    ref = rp._inputs_for_level(2)
    assert not ref.stop_point and pair2 is not None
    rp.close()


def test_frenetix_handler_stopping_trajectories():
    from frenetix_motion_planner_amd import frenetix_compat as fx
    from oracle import oracle
    veh = VehicleParams()
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = fx.CoordinateSystemWrapper(ref)
    h = fx.TrajectoryHandler(dt=0.1)
    h.add_feasability_function(fx.CheckYawRateConstraint(deltaMax=veh.delta_max, wheelbase=veh.wheelbase, wholeTrajectory=False))
    h.add_feasability_function(fx.CheckAccelerationConstraint(switchingVelocity=veh.v_switch, maxAcceleration=veh.a_max, wholeTrajectory=False))
    h.add_feasability_function(fx.CheckCurvatureConstraint(deltaMax=veh.delta_max, wheelbase=veh.wheelbase, wholeTrajectory=False))
    h.add_feasability_function(fx.CheckCurvatureRateConstraint(wheelbase=veh.wheelbase, velocityDeltaMax=veh.v_delta_max, wholeTrajectory=False))
    for cls, name, w in ((fx.CalculateLateralJerkCost, "lateral_jerk", 0.2), (fx.CalculateLongitudinalJerkCost, "longitudinal_jerk", 0.2),
                         (fx.CalculateDistanceToReferencePathCost, "distance_to_reference_path", 5.0)):
        h.add_cost_function(cls(name, w))
    h.add_function(fx.FillCoordinates(lowVelocityMode=False, initialOrientation=float(cs.ref_theta[40]), coordinateSystem=cs, horizon=3))
    h.add_cost_function(fx.CalculateVelocityOffsetCost("velocity_offset", 1.0, 0.0, 0.1, 1.1, limit_to_t_min=False, norm_order=2))
    s0 = float(cs.ref_pos[40] + 0.1)
    ps = fx.PlannerState(fx.CartesianPlannerState(np.zeros(2), float(cs.ref_theta[40]), 6.0, 0.0, 0.0),
                         fx.CurvilinearPlannerState([s0, 6.0, 0.0], [0.2, 0.0, 0.0]), veh.wheelbase)
    cfg = fx.SamplingConfiguration(t_min=0.5, t_max=10.0, dt=0.1, d_delta=0.4, sampling_level=4,
                                   time_based_lateral_delta_scaling=True, enforce_time_bounds=True, strict_velocity_sampling=True)
    with pytest.raises(ValueError):
        h.generate_stopping_trajectories(ps, cfg, s0 - 1.0, 0.0, False)
    h.reset_Trajectories()
    h.generate_stopping_trajectories(ps, cfg, s0 + 20.0, 0.0, False)
    h.evaluate_all_current_functions(True)
    trajs = h.get_sorted_trajectories()
    step = h._step
    assert step.inputs.stop_point and len(trajs) > 0
    out = oracle.plan_step(step.inputs)
    assert step.result["best_index"] == out["result"]["best_index"]
    assert step.result["n_feasible"] == out["result"]["n_feasible"]
    costs = [tr.cost for tr in trajs]
    assert costs == sorted(costs)


def test_scenario_file_to_plan(tmp_path):
    """CommonRoad XML -> route centre line -> prepared reference path -> ground-truth predictions -> plan step."""
    from frenetix_motion_planner_amd import commonroad_xml as crx, ref_path
    from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip
    from oracle import oracle
    from tests.fixtures import tiny_commonroad_xml
    f = tmp_path / "tiny.xml"
    f.write_text(tiny_commonroad_xml(lead_x=15.0))
    sc = crx.read_scenario(str(f))
    pp = sc.planning_problems[100]
    reference = ref_path.prepare_reference_path(ref_path.resample_polyline(sc.route_reference_path(pp), 0.125))
    rp = ReactivePlannerHip(PlannerConfig(dt=sc.dt), VehicleParams())
    rp.update_externals(reference_path=reference, x_0=pp.initial_planner_state(), desired_velocity=10.0,
                        predictions=sc.ground_truth_predictions(0, 30))
    rp.set_road_boundary(sc.road_boundary_segments())
    pair = rp.plan()
    assert pair is not None
    best = rp.optimal_trajectory
    inp = rp._inputs_for_level(2)
    assert inp.mode & _abi.FX_MODE_ROAD_BOUNDARY
    inp.obstacles = synthetic.pack_predictions(rp.predictions, 31, oracle.build_obstacle_hulls)
    out = oracle.plan_step(inp)
    assert best.uniqueId == out["result"]["best_index"] and rp.infeasible_count_collision == out["result"]["n_collisions"]
    # the lane is 4 m wide and the left neighbour ends at x = 60: wide lateral end states leave the road
    assert out["boundary"].sum() > 0 and best.leaves_road is False and best.boundary_harm == 0
    g_off = int(np.nonzero(out["boundary"])[0][0])
    off = rp.last_step.sample(g_off)
    i_off = int(out["boundary_step"][g_off])
    assert off.leaves_road and off.boundary_harm == pytest.approx(1.0 / (1.0 + np.exp(4.591 - 0.185 * off.cartesian.v[i_off])))
    # the ego starts 10 m behind a slower car in its lane: the collision stage must have rejected candidates
    assert out["collision"].sum() > 0 and not out["collision"][best.uniqueId]
    x = np.array([st.position[0] for st in pair[0]])
    assert np.all(np.diff(x) > 0) and abs(pair[0][0].position[1] - 0.2) < 1e-9
    rp.close()


def test_last_level_fallback_selector_on_the_engine():
    """The Python back-end's last-level selection among colliding feasible trajectories (reactive_planner.py:262-269) through the
    hook, on the real engine: same choice as on the oracle-backed stand-in."""
    from frenetix_motion_planner_amd.reactive_planner import ReactivePlannerHip
    from tests.test_planner_host import blocked_planner

    def risk(tr):
        sp = tr.sampling_parameters
        return round(abs(sp[10]), 3) + sp[5] / 100.0

    got = []
    for engine in (None, "oracle"):
        rp = blocked_planner(engine=engine, sampling_min=1, sampling_max=3)
        rp.set_fallback_selector(ReactivePlannerHip.min_risk_selector(risk))
        pair = rp.plan()
        best = rp.optimal_trajectory
        assert pair is not None and best is not None and rp.last_step.result["best_index"] == -1
        got.append((best.uniqueId, rp.last_step.result["n_collisions"], rp.last_step.result["n_feasible"]))
        rp.close()
    assert got[0] == got[1] and got[0][1] == got[0][2] > 0


def test_occlusion_module_call_points_on_the_engine():
    """planner.py:271-273, 384-388; trajectories.py:557-560: an occlusion module's added costs and vetoes on the real engine -- the
    module sees the same trajectories in the same order, is asked for the same candidates and the planner returns the same
    trajectory as on the oracle-backed stand-in."""
    from tests.test_planner_host import _ToyOcclusionModule, _open_road_planner
    got = []
    for engine in (None, "oracle"):
        probe = _open_road_planner(engine=engine)
        assert probe.plan() is not None
        v_veto = float(np.sort(np.unique(probe.last_step.inputs.v_samp))[-2])
        probe.close()
        occ = _ToyOcclusionModule(v_veto=v_veto, weight=3.0)
        rp = _open_road_planner(engine=engine)
        rp.set_occlusion_module(occ)
        pair = rp.plan()
        assert pair is not None and len(occ.assessed) >= 1 and occ.assessed[-1] == rp.optimal_trajectory.uniqueId
        got.append((occ.calc_calls, occ.assessed, rp.optimal_trajectory.uniqueId, rp._collision_counter))
        rp.close()
    assert got[0] == got[1]
