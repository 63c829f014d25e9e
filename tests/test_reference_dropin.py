"""The reference's own `ReactivePlannerCpp` (unmodified, imported from /root/reference) planning on top of this package's
`frenetix` module -- tests/dropin/run_reference_cpp_planner.py, run in a subprocess because it installs import stubs.
Only where the reference tree exists (the build container); the engine behind the handler is the oracle stand-in."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "dropin", "run_reference_cpp_planner.py")
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/frenetix_motion_planner"),
                                reason="reference tree not present (GPU box)")


def run(*flags):
    r = subprocess.run([sys.executable, SCRIPT, *flags], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_reference_planner_plans_through_this_frenetix_module():
    from frenetix_motion_planner_amd import VehicleParams, synthetic
    from frenetix_motion_planner_amd.problem import PlanInputs, pack_predictions
    from frenetix_motion_planner_amd.sampling import SamplingHandler, generate_sampling_matrix, v_sampling_bounds
    from oracle import oracle
    got = run()
    assert got["planned"] and got["n_matrix"] == 800 and got["n_states"] == 31 and got["all_traj"] == 800
    # the same step, assembled by hand: C++-style sampling matrix (reactive_planner_cpp.py:228-253) -> oracle
    veh = VehicleParams()
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = synthetic.CoordinateSystem(ref)
    s0 = float(cs.ref_pos[40] + 0.1)
    x_lon, x_lat = got["x_cl"]
    sh = SamplingHandler(dt=0.1, max_sampling_number=3, t_min=1.1, horizon=3.0, delta_d_min=-3.0, delta_d_max=3.0, d_ego_pos=False)
    sh.set_v_sampling(*v_sampling_bounds(10.0, veh.a_max, 3.0, veh.v_max))
    t1 = np.array(list(sh.t_sampling.to_range(2).union({30 * 0.1})))
    v1 = np.array(list(sh.v_sampling.to_range(2).union({x_lon[1]})))
    d1 = np.array(list(sh.d_sampling.to_range(2).union({x_lat[0]})))
    m = generate_sampling_matrix(t0_range=0.0, t1_range=t1, s0_range=x_lon[0], ss0_range=x_lon[1], sss0_range=x_lon[2],
                                 ss1_range=v1, sss1_range=0, d0_range=x_lat[0], dd0_range=x_lat[1], ddd0_range=x_lat[2],
                                 d1_range=d1, dd1_range=0.0, ddd1_range=0.0)
    preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
    inp = PlanInputs(N=30, dt=0.1, low_vel_mode=False, x0_lon=x_lon, x0_lat=x_lat, x0_orientation=float(cs.ref_theta[40]),
                     v_des=12.0, vehicle=veh, coordinate_system=cs, sampling_matrix=m,
                     cost_weights={"distance_to_reference_path": 5.0, "lateral_jerk": 0.2, "longitudinal_jerk": 0.2,
                                   "prediction": 0.2, "velocity_offset": 1.0},
                     draw_traj_set=True, kinematic_debug=True, obstacles=pack_predictions(preds, 31, oracle.build_obstacle_hulls))
    out = oracle.plan_step(inp)
    g = out["result"]["best_index"]
    assert got["optimal_id"] == g and got["optimal_cost"] == pytest.approx(out["result"]["best_cost"], rel=1e-12)
    assert got["collisions"] == out["result"]["n_collisions"]
    n_feas = int((out["feasible"] & out["valid"]).sum())
    n_inf = int((~out["feasible"] & out["valid"]).sum())
    assert got["feasible_percentage"] == pytest.approx(100.0 * n_feas / (n_feas + n_inf))
    inf = ~out["feasible"] & out["valid"]
    per_reason = [int(((out["reasons"][inf] >> r) & 1).sum()) for r in (5, 6, 7, 8)]       # curvature, yaw rate, curvature rate, acceleration
    assert got["infeasible_hist"][5:9] == [float(c) for c in per_reason] and got["infeasible_hist"][0] == float(sum(per_reason))
    pl = out["planes"][g]
    assert np.allclose(got["x_cl_next"][0], pl[[7, 10, 11], 1], atol=1e-12) and np.allclose(got["x_cl_next"][1], pl[[8, 12, 13], 1], atol=1e-12)
    assert np.allclose(got["sampling_parameters"], m[g]) and set(got["costmap"]) == set(inp.cost_names)
    assert np.allclose(got["last"][:2], pl[[0, 1], 30], atol=1e-12)


def test_reference_emergency_selection_through_this_module():
    """every feasible candidate collides -> the reference's own _select_stopping_trajectory picks from this module's
    TrajectorySample objects (sampling_parameters / feasible surface)"""
    got = run("--blocked")
    assert got["planned"] and got["collisions"] > 100 and got["collisions"] == round(got["feasible_percentage"] * 8)
    assert got["sampling_parameters"][5] == pytest.approx(0.001)          # the slowest sampled end velocity
    assert got["sampling_parameters"][10] == pytest.approx(got["x_cl"][1][0])   # lateral end position closest to d_pos
