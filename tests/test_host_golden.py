"""The host rows around the hot path against vectors produced by the REFERENCE's own methods (tests/golden/host_golden.npz,
written by tests/golden/gen_host_golden.py from /root/reference):

  a22  Planner._compute_initial_states       planner.py:567-635     -> CoordinateSystem.frenet_state / _compute_initial_states /
                                                                       frenetix.compute_initial_state
  a21  _compute_trajectory_pair, shift_orientation, _compute_cart_traj   planner.py:394-447, 536-542, 449-486
                                                                    -> the planner's long way, the package block (CPU stand-in
                                                                       and, -m gpu, fx_read_package of the HIP engine)
  a23  _compute_standstill_trajectory        reactive_planner.py:579-626
       _select_stopping_trajectory           reactive_planner_cpp.py:443-466

With the reference's own inputs the derived values are held to 1e-12 (the arithmetic is the same sequence of float64
operations; 1e-12 leaves room for the last bit of tan / atan2).  The HIP engine recomputes the trajectory itself, so its rows
carry the plane tolerance of the parity tests (1e-9 of the row's scale)."""
import json
import os

import numpy as np
import pytest

from frenetix_motion_planner_amd import CoordinateSystem, VehicleParams, _abi
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
from frenetix_motion_planner_amd.trajectories import CartesianSample, CurviLinearSample
from tests.fixtures import GOLDEN_DIR, inputs_from_fixture, load_golden

HG = dict(np.load(os.path.join(GOLDEN_DIR, "host_golden.npz"), allow_pickle=False))
INDEX = json.loads(str(HG["index"]))
EXACT = 1e-12
PAIR_KEYS = ("cart_time_step", "cart_position", "cart_orientation", "cart_velocity", "cart_acceleration", "cart_yaw_rate",
             "cart_steering_angle", "cl_time_step", "cl_position", "cl_velocity", "cl_acceleration", "cl_orientation", "cl_yaw_rate",
             "lon_list", "lat_list")

_CS = {}


def cs_of(rname):
    if rname not in _CS:
        _CS[rname] = CoordinateSystem(HG[f"ref/{rname}"])
    return _CS[rname]


def close(a, b, tol=EXACT):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all(np.abs(a - b) <= tol * (1.0 + np.abs(b))))


def test_vector_set_is_what_the_verdict_asked_for():
    """>= 6 states per row: high velocity, LOW_VEL_MODE, standstill, negative curvature, heading wrap across +-pi, config-1 route"""
    refs = {c.split("/")[0] for c in INDEX["init"]}
    assert {"arc", "arc_negk", "arc_rot_pi", "arc_negk_rot", "zam_route", "scurve"} <= refs
    assert {c.split("/")[1] for c in INDEX["init"]} >= {"hv", "lv", "standstill"}
    assert len(INDEX["init"]) >= 36 and len(INDEX["still"]) >= 12
    assert sum(len(v) for v in INDEX["pair"].values()) >= 24 and sum(len(v) for v in INDEX["stopping"].values()) >= 20
    # the wrap cases do wrap: some state's heading had to be shifted by 2 pi
    shifted = 0
    for name, ids in INDEX["pair"].items():
        for g in ids:
            shifted += bool(np.any(np.abs(HG[f"pair/{name}/{g}/cart_orientation"] - HG[f"pair/{name}/{g}/planes"][2]) > 6.0))
    assert shifted >= 3


# ----------------------------------------------------------------------------------------------------------------- a22
@pytest.mark.parametrize("case", INDEX["init"])
def test_initial_state_against_the_reference(case, monkeypatch):
    rname = case.split("/")[0]
    cs = cs_of(rname)
    x, y, heading, v, a, delta, low_vel = HG[f"init/{case}/in"]
    veh = VehicleParams()
    kappa0 = np.tan(delta) / veh.wheelbase
    lon, lat = cs.frenet_state(x, y, heading, v, a, kappa0, arc_length_lateral=bool(low_vel))
    # the whole chain (inverse projection of this package, then the reference's transform)
    assert close(lon, HG[f"init/{case}/lon"], 1e-9) and close(lat, HG[f"init/{case}/lat"], 1e-9), case
    # the transform alone, on the (s, d) the reference was given: to the last bits
    sd = HG[f"init/{case}/sd"]
    monkeypatch.setattr(CoordinateSystem, "convert_to_curvilinear_coords", lambda self, px, py: sd.copy())
    lon, lat = cs.frenet_state(x, y, heading, v, a, kappa0, arc_length_lateral=bool(low_vel))
    assert close(lon, HG[f"init/{case}/lon"]) and close(lat, HG[f"init/{case}/lat"]), (case, lon, lat)
    # the planner's method and the frenetix-module function are the same computation
    rp = ReactivePlannerHip(PlannerConfig(), veh, engine=object())
    rp.coordinate_system = cs
    rp._LOW_VEL_MODE = bool(low_vel)
    st = ReactivePlannerState(position=np.array([x, y]), orientation=heading, velocity=v, acceleration=a, steering_angle=delta)
    lon2, lat2 = rp._compute_initial_states(st)
    assert close(lon2, HG[f"init/{case}/lon"]) and close(lat2, HG[f"init/{case}/lat"])
    from frenetix_motion_planner_amd import frenetix_compat as fc
    got = fc.compute_initial_state(coordinate_system=cs, x_0=fc.CartesianPlannerState(np.array([x, y]), heading, v, a, delta),
                                   wheelbase=veh.wheelbase, low_velocity_mode=bool(low_vel))
    assert close(got.x0_lon, HG[f"init/{case}/lon"]) and close(got.x0_lat, HG[f"init/{case}/lat"])


def test_initial_state_facing_against_the_reference_raises():
    assert INDEX["init_err"][0] == "Exception" and "negative" in INDEX["init_err"][1]
    x, y, heading, v = HG["init_err/against/in"][:4]
    with pytest.raises(Exception, match="negative"):
        cs_of("arc").frenet_state(x, y, heading, v, 0.0, 0.0, arc_length_lateral=False)


# ----------------------------------------------------------------------------------------------------------------- a21
def _sample_from_planes(p):
    class T:
        pass
    t = T()
    n = p.shape[1]
    t.cartesian = CartesianSample(p[0], p[1], p[2], p[3], p[4], p[5], p[6], n)
    t.curvilinear = CurviLinearSample(p[7], p[8], p[9], dd=p[12], ddd=p[13], ss=p[10], sss=p[11], current_time_step=n)
    return t


def _planner_for_pair(x0row):
    ts, orientation, yaw0, dt, wheelbase = x0row
    rp = ReactivePlannerHip(PlannerConfig(dt=float(dt)), VehicleParams(wheelbase=float(wheelbase)), engine=object())
    rp.x_0 = ReactivePlannerState(time_step=int(ts), orientation=float(orientation), yaw_rate=float(yaw0))
    return rp


def _pair_as_arrays(pair):
    cart, cl, lon, lat = pair
    cart, cl = list(cart), list(cl)
    g = (lambda o, k: o[k]) if isinstance(cl[0], dict) else getattr
    return dict(
        cart_time_step=[s.time_step for s in cart], cart_position=[s.position for s in cart],
        cart_orientation=[s.orientation for s in cart], cart_velocity=[s.velocity for s in cart],
        cart_acceleration=[s.acceleration for s in cart], cart_yaw_rate=[s.yaw_rate for s in cart],
        cart_steering_angle=[s.steering_angle for s in cart],
        cl_time_step=[g(s, "time_step") for s in cl], cl_position=[g(s, "position") for s in cl],
        cl_velocity=[g(s, "velocity") for s in cl], cl_acceleration=[g(s, "acceleration") for s in cl],
        cl_orientation=[g(s, "orientation") for s in cl], cl_yaw_rate=[g(s, "yaw_rate") for s in cl],
        lon_list=[list(r) for r in lon], lat_list=[list(r) for r in lat])


PAIR_CASES = [(name, g) for name, ids in INDEX["pair"].items() for g in ids]


@pytest.mark.parametrize("name,g", PAIR_CASES)
def test_trajectory_pair_long_way_against_the_reference(name, g):
    key = f"pair/{name}/{g}"
    rp = _planner_for_pair(HG[f"pair/{name}/x0"])
    tr = _sample_from_planes(HG[key + "/planes"])
    got = _pair_as_arrays(rp._compute_trajectory_pair(tr))
    for k in PAIR_KEYS:
        assert close(got[k], HG[f"{key}/{k}"]), (name, g, k)
    ct = rp._compute_cart_traj(tr)
    assert close([s.yaw_rate for s in ct], HG[key + "/carttraj_yaw_rate"])
    assert close([s.steering_angle for s in ct], HG[key + "/carttraj_steering_angle"])
    assert close([s.orientation for s in ct], HG[key + "/carttraj_orientation"])
    assert [s.time_step for s in ct] == HG[key + "/carttraj_time_step"].tolist()


def _check_block_against_pair(block, key, tol, t0, rp=None, pkg=None):
    """a package block ([FX_PKG_ROWS][S]: 14 planes, yaw rate, steering angle, shifted heading) against the reference's pair"""
    def rowclose(a, b):
        b = np.asarray(b, dtype=np.float64)
        return np.abs(np.asarray(a) - b).max() <= tol * (1.0 + np.abs(b).max())
    assert rowclose(block[_abi.PKG_ROW_YAW_RATE], HG[key + "/cart_yaw_rate"]), key
    assert rowclose(block[_abi.PKG_ROW_STEERING], HG[key + "/cart_steering_angle"]), key
    assert rowclose(block[_abi.PKG_ROW_ORIENTATION], HG[key + "/cart_orientation"]), key
    assert rowclose(block[0], HG[key + "/cart_position"][:, 0]) and rowclose(block[1], HG[key + "/cart_position"][:, 1])
    assert rowclose(block[3], HG[key + "/cart_velocity"]) and rowclose(block[4], HG[key + "/cart_acceleration"])
    assert rowclose(block[2], HG[key + "/cl_orientation"]) and rowclose(block[5], HG[key + "/cl_yaw_rate"])
    lon, lat = HG[key + "/lon_list"], HG[key + "/lat_list"]
    for r, col in ((7, lon[:, 0]), (10, lon[:, 1]), (11, lon[:, 2]), (8, lat[:, 0]), (12, lat[:, 1]), (13, lat[:, 2])):
        assert rowclose(block[r], col), (key, r)
    if rp is not None:   # the planner's packaged pair: lazily built state objects over the block
        class T:
            pass
        t = T()
        t._pkg = pkg
        got = _pair_as_arrays(rp._compute_trajectory_pair(t))
        for k in PAIR_KEYS:
            want = HG[f"{key}/{k}"]
            assert np.abs(np.asarray(got[k], dtype=np.float64) - want).max() <= tol * (1.0 + np.abs(want).max()), (key, k)
        # x_cl of the next cycle (frenet_interface.py:255)
        assert np.abs(np.asarray(got["lon_list"][1]) - lon[1]).max() <= tol * (1 + np.abs(lon[1]).max())
        assert np.abs(np.asarray(got["lat_list"][1]) - lat[1]).max() <= tol * (1 + np.abs(lat[1]).max())


@pytest.mark.parametrize("name,g", PAIR_CASES)
def test_package_block_of_the_cpu_stand_in_against_the_reference(name, g):
    """tests/oracle_engine._OraclePackage (what the CPU planner tests run on) fed the reference's planes"""
    from tests.oracle_engine import _OraclePackage
    key = f"pair/{name}/{g}"
    ts, orientation, yaw0, dt, wheelbase = HG[f"pair/{name}/x0"]
    planes = HG[key + "/planes"]

    class Inp:
        shard_begin, write_costmap = 0, False
        x0_orientation = float(orientation)
        vehicle = VehicleParams(wheelbase=float(wheelbase))
    Inp.dt = float(dt)
    z = np.zeros(1)
    out = dict(planes=planes[None], cost=z, flags=np.zeros(1, dtype=np.uint32), traj_len=np.zeros(1, dtype=np.int32),
               coeff_lon=np.zeros((1, 6)), coeff_lat=np.zeros((1, 6)), tau_lat=z)
    pkg = _OraclePackage(Inp, out, 0, float(yaw0))
    _check_block_against_pair(pkg.block, key, EXACT, int(ts), rp=_planner_for_pair(HG[f"pair/{name}/x0"]), pkg=pkg)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(INDEX["pair"]))
def test_hip_package_against_the_reference_pair(name):
    """fx_plan_and_package / fx_read_package on the MI355X: the winner of the plan-step golden, its derived rows (yaw rate,
    steering angle, shifted heading) and lon_list[1] / lat_list[1] against the reference's _compute_trajectory_pair vectors"""
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    fx = load_golden(name)
    inp = inputs_from_fixture(fx, build_obstacle_hulls, collision=False)
    ts, orientation, yaw0, dt, wheelbase = HG[f"pair/{name}/x0"]
    assert inp.x0_orientation == orientation and inp.dt == dt and inp.vehicle.wheelbase == wheelbase
    g = INDEX["pair"][name][0]
    assert g == int(fx["walk_ids"][0])
    with FrenetEngine(max_candidates=max(4096, inp.n_candidates), max_ref_knots=1024) as eng:
        res, pkg = eng.plan_step_packaged(inp, yaw_rate0=float(yaw0))
        assert res["best_index"] == g and pkg is not None and pkg.index == g
        _check_block_against_pair(pkg.block, f"pair/{name}/{g}", 1e-9, int(ts), rp=_planner_for_pair(HG[f"pair/{name}/x0"]), pkg=pkg)
        # second entry point: evaluation, then fx_read_package on its own
        eng.set_package(True)
        eng.plan_step(inp)
        pkg2 = eng.package(0, float(yaw0))
        assert pkg2.index == g and np.array_equal(pkg2.block, pkg.block)


# ----------------------------------------------------------------------------------------------------------------- a23
@pytest.mark.parametrize("case", INDEX["still"])
def test_standstill_trajectory_against_the_reference(case):
    rname = case.split("/")[0]
    x, y, heading, v0, delta, s, sd, sdd, d, dd, ddd = HG[f"still/{case}/in"]
    ts, orientation, yaw0, dt, wheelbase = HG[f"still/{case}/x0"]
    rp = ReactivePlannerHip(PlannerConfig(dt=float(dt)), VehicleParams(wheelbase=float(wheelbase)), engine=object())
    rp.coordinate_system = cs_of(rname)
    rp.x_0 = ReactivePlannerState(time_step=int(ts), position=np.array([x, y]), orientation=float(heading), velocity=float(v0),
                                  steering_angle=float(delta), yaw_rate=float(yaw0))
    rp.x_cl = ([s, sd, sdd], [d, dd, ddd])
    tr = rp._compute_standstill_trajectory()
    c, k = tr.cartesian, tr.curvilinear
    want_c, want_k = HG[f"still/{case}/cartesian"], HG[f"still/{case}/curvilinear"]
    assert want_c.shape[1] == rp.N                       # N entries, not N + 1 (reactive_planner.py:608-625)
    assert close(np.stack([c.x, c.y, c.theta, c.v, c.a, c.kappa, c.kappa_dot]), want_c)
    assert close(np.stack([k.s, k.d, k.theta, k.s_dot, k.s_ddot, k.d_dot, k.d_ddot]), want_k)
    assert want_c[4, 1] == -v0 / dt and c.a[1] == want_c[4, 1]
    uid, horizon, dt_, n_c, n_k, tau_lon, tau_lat = HG[f"still/{case}/meta"]
    assert tr.uniqueId == uid == 0 and tr.horizon == horizon and tr.dt == dt_
    assert c.current_time_step == n_c and k.current_time_step == n_k
    assert tr.trajectory_long.delta_tau == tau_lon and tr.trajectory_lat.delta_tau == tau_lat
    assert close(tr.trajectory_long.coeffs, HG[f"still/{case}/coeff_lon"], 1e-10)   # (LAPACK solve vs closed form, as a3 / a4)
    assert close(tr.trajectory_lat.coeffs, HG[f"still/{case}/coeff_lat"], 1e-10)
    got = _pair_as_arrays(rp._compute_trajectory_pair(tr))
    for kk in PAIR_KEYS:
        assert close(got[kk], HG[f"still/{case}/pair/{kk}"]), (case, kk)
    # the frenetix-module entry point (reactive_planner_cpp.py:220-226)
    from frenetix_motion_planner_amd import frenetix_compat as fc
    ps = fc.PlannerState(fc.CartesianPlannerState(np.array([x, y]), float(heading), float(v0), 0.0, float(delta)),
                         fc.CurvilinearPlannerState(np.array([s, sd, sdd]), np.array([d, dd, ddd])), float(wheelbase))
    tr2 = fc.TrajectorySample.compute_standstill_trajectory(cs_of(rname), ps, float(dt), float(horizon))
    assert close(np.stack([tr2.cartesian.x, tr2.cartesian.y, tr2.cartesian.theta, tr2.cartesian.v, tr2.cartesian.a,
                           tr2.cartesian.kappa, tr2.cartesian.kappa_dot]), want_c)


# ------------------------------------------------------------------------------------------- the C++ back-end's emergency pick
STOP_CASES = [(name, ci) for name, cs_ in INDEX["stopping"].items() for ci in cs_]


@pytest.mark.parametrize("name,ci", STOP_CASES)
def test_stopping_selection_against_the_reference(name, ci):
    from frenetix_motion_planner_amd.trajectories import PlanStepResult
    from oracle import oracle
    fx = load_golden(name)
    inp = inputs_from_fixture(fx, oracle.build_obstacle_hulls, collision=False)
    ids = HG[f"stopping/{name}/{ci}/ids"]
    d_pos, chosen = HG[f"stopping/{name}/{ci}/d_pos_chosen"]
    flags = np.zeros(inp.n_candidates, dtype=np.uint32)
    flags[ids] = _abi.FX_FLAG_VALID | _abi.FX_FLAG_FEASIBLE | _abi.FX_FLAG_RETURNED

    class Eng:   # the selection reads the flag word only
        def costs(self, agent=0):
            return np.zeros(inp.n_candidates), flags
    step = PlanStepResult(Eng(), inp, dict(n_candidates=inp.n_candidates, best_index=-1))
    got = ReactivePlannerHip._select_stopping_trajectory(step, float(d_pos))
    assert (-1 if got is None else got.uniqueId) == int(chosen), (name, ci)
    if ci == 0:   # the full pool of the golden is what the engine flags (masks are pinned by test_oracle_golden / test_hip_parity)
        assert np.array_equal(ids, np.nonzero(fx["valid"] & fx["feasible"] & fx["returned"])[0])
