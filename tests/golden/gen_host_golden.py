"""Golden vectors of the HOST rows around the hot path (rows a21, a22, a23 and the C++ back-end's
emergency pick), produced by calling the REFERENCE's own methods, imported from /root/reference through ref_harness.py:

  init      Planner._compute_initial_states                      planner.py:567-635
  pair      Planner._compute_trajectory_pair + shift_orientation  planner.py:394-447, 536-542
            Planner._compute_cart_traj                            planner.py:449-486
  still     ReactivePlannerPython._compute_standstill_trajectory  reactive_planner.py:579-626   (+ its trajectory pair)
  stopping  ReactivePlannerCpp._select_stopping_trajectory        reactive_planner_cpp.py:443-466

Run in the build container only:     python tests/golden/gen_host_golden.py
Output: tests/golden/host_golden.npz (data only: inputs and the reference's outputs).

What is substituted, and why it does not weaken the pin:
  * CommonRoad's state / trajectory containers are the plain shims of tests/dropin/run_reference_cpp_planner.py;
  * `coordinate_system` is a fake object exposing ref_pos / ref_theta / ref_curv / ref_curv_d of this package's
    CoordinateSystem (a12, pinned separately) and, for `init`, convert_to_curvilinear_coords = this package's inverse
    projection (a11: CCosy, third party, unpinned) -- the (s, d) it returned is stored with the vector, so the Werling
    transform that follows is held to the reference bit for bit;
  * `pair` inputs are the reference's OWN TrajectorySample objects of the plan-step goldens (gen_golden.py scenarios),
    so the HIP engine reproduces them from the committed plan-step fixture of the same name.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
import gen_golden  # noqa: E402
from frenetix_motion_planner_amd import CoordinateSystem, VehicleParams, synthetic  # noqa: E402

OUT = os.path.join(HERE, "host_golden.npz")

# ---- references for `init` / `still`: kind -> reference_polyline kwargs (or the config-1 route)
REFERENCES = {
    "arc": dict(kind="arc"),
    "arc_negk": dict(kind="arc", kappa=-0.02),
    "scurve": dict(kind="scurve", kappa=0.03),
    "straight_jitter": dict(kind="straight", knot_jitter=0.5),
    "arc_rot_pi": dict(kind="arc", heading0=3.0),             # unwrapped heading 3.0 .. 5.0: crosses +pi
    "arc_negk_rot": dict(kind="arc", kappa=-0.02, heading0=-2.9),   # -2.9 .. -6.9: crosses -pi and -2 pi
    "zam_route": None,                                        # BASELINE config 1's route (junction, non-uniform knots)
}

# ---- vehicle states relative to the reference: (s_knot, s_off, d, heading error, v, a, steering angle, LOW_VEL_MODE, wrapped heading)
STATES = {
    "hv": (40, 0.10, 0.30, 0.05, 10.0, 0.5, 0.02, False, True),
    "hv_off": (55, 0.37, 1.80, 0.30, 14.0, -1.5, 0.30, False, True),
    "hv_negd": (25, 0.21, -1.10, -0.20, 7.0, 1.0, -0.15, False, False),
    "lv": (40, 0.10, -0.50, -0.08, 1.5, -0.4, -0.10, True, True),
    "lv_off": (33, 0.45, 0.90, 0.25, 0.8, 0.3, 0.40, True, False),
    "standstill": (40, 0.10, 0.20, 0.01, 0.0, 0.0, 0.05, True, True),
}

PAIR_SCENARIOS = ["arc_hv_l1_prod", "arc_lv_l1_prod", "arc_standstill_l1_debug", "scurve_hv_l1_debug", "arc_rot_hv_l1_prod",
                  "arc_rot_negk_lv_l1_debug", "zam_tjunction_ego_l2_prod", "arc_hv_decel_l1_prod", "arc_stop_l1_prod",
                  "straight_lv_l1_nonuniform_debug"]

STOPPING_SCENARIOS = ["arc_hv_l1_prod", "arc_hv_l2_prod_obs1", "zam_tjunction_ego_l2_prod", "arc_lv_l1_prod", "scurve_hv_l2_kd"]


class FakeCS:
    """what the reference's methods read of `self.coordinate_system`"""

    def __init__(self, cs):
        self._cs = cs
        self.ref_pos, self.ref_theta, self.ref_curv, self.ref_curv_d = cs.ref_pos, cs.ref_theta, cs.ref_curv, cs.ref_curv_d
        self.last_sd = None

    def convert_to_curvilinear_coords(self, x, y):
        self.last_sd = self._cs.convert_to_curvilinear_coords(x, y)
        return self.last_sd


def route_reference():
    inp = gen_golden.scenario_inputs(dict(scenario="ZAM_Tjunction-1_42_T-1", planning_problem=60000, level=2))
    return inp.coordinate_system.reference


def reference_xy(name):
    kw = REFERENCES[name]
    if kw is None:
        return route_reference()
    kw = dict(kw)
    return synthetic.reference_polyline(kw.pop("kind"), 400, 0.5, kw.pop("kappa", 0.01), kw.pop("knot_jitter", 0.0),
                                        heading0=kw.pop("heading0", 0.0))


def wrap_pi(a):
    return float(np.arctan2(np.sin(a), np.cos(a)))


def bare_planner(State, veh, cs, low_vel, horizon=3.0, dt=0.1):
    """a reference ReactivePlannerPython with the attributes the host methods read (Planner.__init__ reads YAML and builds
    the commonroad_dc road boundary: bypassed)"""
    from frenetix_motion_planner.reactive_planner import ReactivePlannerPython
    import logging
    rp = object.__new__(ReactivePlannerPython)
    rp.horizon, rp.dT, rp.N = horizon, dt, int(horizon / dt)
    rp.vehicle_params = ref_harness._Obj(wheelbase=veh.wheelbase, length=veh.length, width=veh.width, wb_rear_axle=veh.wb_rear_axle,
                                         a_max=veh.a_max, v_switch=veh.v_switch, delta_max=veh.delta_max, v_max=veh.v_max)
    rp._LOW_VEL_MODE = bool(low_vel)
    rp.coordinate_system = cs
    rp.msg_logger = logging.getLogger("fx_host_golden")
    rp.msg_logger.setLevel(logging.CRITICAL + 1)
    rp.cost_function = ref_harness._Obj(cost_weights={"distance_to_reference_path": 5.0, "lateral_jerk": 0.2,
                                                      "longitudinal_jerk": 0.2, "prediction": 0.2, "velocity_offset": 1.0})
    return rp


def pair_arrays(pair):
    cart, cl, lon, lat = pair
    cs_, ls_ = cart.state_list, cl.state_list
    return dict(
        cart_time_step=np.array([s.time_step for s in cs_], dtype=np.int64),
        cart_position=np.array([s.position for s in cs_], dtype=np.float64),
        cart_orientation=np.array([s.orientation for s in cs_], dtype=np.float64),
        cart_velocity=np.array([s.velocity for s in cs_], dtype=np.float64),
        cart_acceleration=np.array([s.acceleration for s in cs_], dtype=np.float64),
        cart_yaw_rate=np.array([s.yaw_rate for s in cs_], dtype=np.float64),
        cart_steering_angle=np.array([s.steering_angle for s in cs_], dtype=np.float64),
        cl_time_step=np.array([s.time_step for s in ls_], dtype=np.int64),
        cl_position=np.array([s.position for s in ls_], dtype=np.float64),
        cl_velocity=np.array([s.velocity for s in ls_], dtype=np.float64),
        cl_acceleration=np.array([s.acceleration for s in ls_], dtype=np.float64),
        cl_orientation=np.array([s.orientation for s in ls_], dtype=np.float64),
        cl_yaw_rate=np.array([s.yaw_rate for s in ls_], dtype=np.float64),
        lon_list=np.array(lon, dtype=np.float64), lat_list=np.array(lat, dtype=np.float64),
        initial_time_step=np.array([cart.initial_time_step, cl.initial_time_step], dtype=np.int64))


def gen_init(out, index, State):
    veh = VehicleParams()
    for rname in REFERENCES:
        xy = reference_xy(rname)
        out[f"ref/{rname}"] = xy
        mine = CoordinateSystem(xy)
        for sname, (knot, off, d, e, v, a, delta, low_vel, wrapped) in STATES.items():
            knot = min(knot, len(xy) - 8)
            s = float(mine.ref_pos[knot] + off * (mine.ref_pos[knot + 1] - mine.ref_pos[knot]))
            p = mine.convert_to_cartesian_coords(s, d)
            theta_ref = mine.reference_at(s)[0]
            heading = theta_ref + e
            if wrapped:
                heading = wrap_pi(heading)
            fake = FakeCS(mine)
            rp = bare_planner(State, veh, fake, low_vel)
            x0 = State(time_step=3, position=np.array(p), orientation=heading, velocity=v, steering_angle=delta, acceleration=a,
                       yaw_rate=0.0)
            lon, lat = rp._compute_initial_states(x0)
            key = f"init/{rname}/{sname}"
            out[key + "/in"] = np.array([p[0], p[1], heading, v, a, delta, float(low_vel)])
            out[key + "/sd"] = np.array(fake.last_sd, dtype=np.float64)
            out[key + "/lon"] = np.array(lon, dtype=np.float64)
            out[key + "/lat"] = np.array(lat, dtype=np.float64)
            index["init"].append(f"{rname}/{sname}")
    # error behaviour (planner.py:612-614): facing against the reference -> bare Exception (s' < 0)
    mine = CoordinateSystem(reference_xy("arc"))
    rp = bare_planner(State, veh, FakeCS(mine), False)
    p = mine.convert_to_cartesian_coords(float(mine.ref_pos[40]), 0.2)
    st = State(time_step=0, position=np.array(p), orientation=float(mine.ref_theta[40]) + 3.0, velocity=5.0, steering_angle=0.0,
               acceleration=0.0, yaw_rate=0.0)
    try:
        rp._compute_initial_states(st)
        index["init_err"] = None
    except Exception as ex:  # noqa: BLE001
        index["init_err"] = [type(ex).__name__, str(ex)]
    out["init_err/against/in"] = np.array([p[0], p[1], st.orientation, 5.0, 0.0, 0.0, 0.0])


def reference_step(name):
    """the reference's plan step of a plan-step golden, keeping its TrajectorySample objects"""
    kw, _ = gen_golden.SCENARIOS[name]
    inp = gen_golden.scenario_inputs(kw) if "scenario" in kw else synthetic.make_inputs(**kw)
    prob = gen_golden.to_reference_problem(inp, kw)
    rp = ref_harness.make_planner(prob)
    level = rp._sampling_min
    if prob.get("stop_point_s") is not None:
        bundle = rp._create_end_point_trajectory_bundle(rp.x_cl[0], rp.x_cl[1], prob["stop_point_s"], rp.cost_function, level)
    else:
        bundle = rp._create_trajectory_bundle(rp.x_cl[0], rp.x_cl[1], rp.cost_function, samp_level=level)
    trajs = list(bundle.trajectories)
    returned = rp.check_feasibility(trajs, None, None)
    from frenetix_motion_planner.trajectories import TrajectoryBundle
    feas = [o for o in returned if o.valid is True and o.feasible is True]
    b2 = TrajectoryBundle(list(returned) if rp._draw_traj_set else feas, cost_function=rp.cost_function, multiproc=False,
                          num_workers=1)
    b2.sort()
    walk = [t for t in b2.trajectories if t.feasible is True]
    return inp, prob, rp, trajs, walk


def gen_pair(out, index, State):
    for name in PAIR_SCENARIOS:
        inp, prob, rp, trajs, walk = reference_step(name)
        fx = np.load(os.path.join(HERE, name + ".npz"))
        assert [t.uniqueId for t in walk] == fx["walk_ids"].tolist(), name     # the same step as the committed plan-step golden
        yaw0, ts = 0.0375, 17
        rp.x_0 = State(time_step=ts, position=np.zeros(2), orientation=float(prob["x0_orientation"]),
                       velocity=float(prob.get("x0_velocity", prob["x0_lon"][1])), steering_angle=0.0, acceleration=0.0, yaw_rate=yaw0)
        picks = [walk[0], walk[len(walk) // 2], walk[-1]]
        ids = []
        for tr in picks:
            g = int(tr.uniqueId)
            if g in ids:
                continue
            ids.append(g)
            key = f"pair/{name}/{g}"
            planes = np.stack([tr.cartesian.x, tr.cartesian.y, tr.cartesian.theta, tr.cartesian.v, tr.cartesian.a, tr.cartesian.kappa,
                               tr.cartesian.kappa_dot, tr.curvilinear.s, tr.curvilinear.d, tr.curvilinear.theta, tr.curvilinear.s_dot,
                               tr.curvilinear.s_ddot, tr.curvilinear.d_dot, tr.curvilinear.d_ddot])
            out[key + "/planes"] = planes            # the method's input, in PLANE order
            for k, v in pair_arrays(rp._compute_trajectory_pair(tr)).items():
                out[f"{key}/{k}"] = v
            ct = rp._compute_cart_traj(tr).state_list
            out[key + "/carttraj_yaw_rate"] = np.array([s.yaw_rate for s in ct], dtype=np.float64)
            out[key + "/carttraj_steering_angle"] = np.array([s.steering_angle for s in ct], dtype=np.float64)
            out[key + "/carttraj_orientation"] = np.array([s.orientation for s in ct], dtype=np.float64)   # (not shifted, :462)
            out[key + "/carttraj_time_step"] = np.array([s.time_step for s in ct], dtype=np.int64)
        out[f"pair/{name}/x0"] = np.array([ts, rp.x_0.orientation, yaw0, rp.dT, rp.vehicle_params.wheelbase])
        index["pair"][name] = ids


def gen_still(out, index, State):
    veh = VehicleParams()
    for rname in REFERENCES:
        xy = reference_xy(rname)
        mine = CoordinateSystem(xy)
        for sname, v0, delta, wrapped, lat in (("rest", 0.0, 0.05, True, (0.2, 0.0, 0.0)), ("creep", 0.08, -0.2, False, (-0.4, 0.01, 0.002))):
            knot = min(40, len(xy) - 8)
            s = float(mine.ref_pos[knot] + 0.1)
            p = mine.convert_to_cartesian_coords(s, lat[0])
            heading = mine.reference_at(s)[0] + 0.02
            if wrapped:
                heading = wrap_pi(heading)
            rp = bare_planner(State, veh, FakeCS(mine), True)
            rp.x_0 = State(time_step=5, position=np.array(p), orientation=heading, velocity=v0, steering_angle=delta, acceleration=0.0,
                           yaw_rate=0.011)
            rp.x_cl = ([s, v0, 0.0], list(lat))
            tr = rp._compute_standstill_trajectory()
            key = f"still/{rname}/{sname}"
            out[key + "/in"] = np.array([p[0], p[1], heading, v0, delta, s, v0, 0.0, *lat])
            c, k = tr.cartesian, tr.curvilinear
            out[key + "/cartesian"] = np.stack([c.x, c.y, c.theta, c.v, c.a, c.kappa, c.kappa_dot])
            out[key + "/curvilinear"] = np.stack([k.s, k.d, k.theta, k.s_dot, k.s_ddot, k.d_dot, k.d_ddot])
            out[key + "/coeff_lon"] = np.asarray(tr.trajectory_long.coeffs, dtype=np.float64)
            out[key + "/coeff_lat"] = np.asarray(tr.trajectory_lat.coeffs, dtype=np.float64)
            out[key + "/meta"] = np.array([tr.uniqueId, tr.horizon, tr.dt, c.current_time_step, k.current_time_step,
                                           tr.trajectory_long.delta_tau, tr.trajectory_lat.delta_tau], dtype=np.float64)
            for kk, v in pair_arrays(rp._compute_trajectory_pair(tr)).items():
                out[f"{key}/pair/{kk}"] = v
            out[key + "/x0"] = np.array([5, heading, 0.011, rp.dT, veh.wheelbase])
            index["still"].append(f"{rname}/{sname}")


def gen_stopping(out, index):
    import frenetix_motion_planner.reactive_planner_cpp as rpc
    from frenetix_motion_planner.sampling_matrix import generate_sampling_matrix
    rng = np.random.default_rng(20241008)
    for name in STOPPING_SCENARIOS:
        fx = np.load(os.path.join(HERE, name + ".npz"))
        t, v, d = fx["t_order"], fx["v_order"], fx["d_order"]
        lon, lat = fx["x0_lon"], fx["x0_lat"]
        m = generate_sampling_matrix(t0_range=0.0, t1_range=t, s0_range=lon[0], ss0_range=lon[1], sss0_range=lon[2], ss1_range=v,
                                     sss1_range=0.0, d0_range=lat[0], dd0_range=lat[1], ddd0_range=lat[2], d1_range=d, dd1_range=0.0,
                                     ddd1_range=0.0)
        assert len(m) == len(fx["valid"])
        # creation order of the reference's bundle == row order of the matrix with (t, v, d) in these orders (a6)
        g = np.arange(len(m))
        assert np.array_equal(m[:, 1], t[g // (len(v) * len(d))]) and np.array_equal(m[:, 10], d[g % len(d)])
        pool = np.nonzero(fx["valid"] & fx["feasible"] & fx["returned"])[0]
        cases = []
        for ci, (d_pos, keep) in enumerate(((float(lat[0]), 1.0), (0.0, 1.0), (-1.2, 0.6), (2.9, 0.3), (float(lat[0]), 0.1))):
            ids = pool if keep >= 1.0 else pool[rng.uniform(size=len(pool)) < keep]
            trs = [ref_harness._Obj(sampling_parameters=m[i], uniqueId=int(i)) for i in ids]
            got = rpc.ReactivePlannerCpp._select_stopping_trajectory(trs, m, d_pos)
            out[f"stopping/{name}/{ci}/ids"] = ids.astype(np.int64)
            out[f"stopping/{name}/{ci}/d_pos_chosen"] = np.array([d_pos, -1 if got is None else got.uniqueId])
            cases.append(ci)
        index["stopping"][name] = cases


def main():
    ref_harness.install()
    sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
    from tests.dropin.run_reference_cpp_planner import install_container_shims
    install_container_shims()
    from frenetix_motion_planner.state import ReactivePlannerState as State
    out, index = {}, dict(init=[], pair={}, still=[], stopping={})
    gen_init(out, index, State)
    gen_pair(out, index, State)
    gen_still(out, index, State)
    gen_stopping(out, index)
    out["index"] = np.array(json.dumps(index, sort_keys=True))
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.0f} KiB; init {len(index['init'])}, pair "
          f"{sum(len(v) for v in index['pair'].values())}, still {len(index['still'])}, stopping "
          f"{sum(len(v) for v in index['stopping'].values())}; init_err {index['init_err']}")


if __name__ == "__main__":
    main()
