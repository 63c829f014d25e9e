"""
Import harness for the *reference's own* Python hot path (/root/reference), used ONLY to produce
the golden vectors under tests/golden/*.npz (see gen_golden.py) and, in this container, to
cross-check the oracle directly.  /root/reference does not exist on the GPU box, so nothing at test
run time depends on this module unless the reference tree is present.

Method: the reference's third-party dependencies that are not installed here
(commonroad-io, commonroad-drivability-checker, omegaconf, shapely, frenetix, ...) are served as
permissive stub packages by a sys.meta_path finder; the handful of helpers whose *arithmetic* the hot
path really uses get real shims:
  * commonroad.common.validity.{is_natural_number,is_positive,is_real_number,is_real_number_vector}
  * methodtools.lru_cache            -> identity decorator (polynomial_trajectory.py:293,452)
  * scipy.integrate.simps            -> scipy.integrate.simpson (removed alias; same 'simpson' rule)
  * commonroad.common.util.make_valid_orientation -> +-2pi wrap into [-2pi, 2pi] (restated, unpinned)
ReactivePlannerPython is then created with object.__new__ and its attributes set by hand; its
_create_trajectory_bundle / check_feasibility / TrajectoryBundle.sort run unmodified.

The curvilinear->Cartesian projection (commonroad_dc CCosy, C++, not in tree) is replaced by
`NumpyProjection`, an independent numpy restatement of the normative definition in DESIGN.md.
"""
import importlib.abc
import importlib.machinery
import logging
import math
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("FX_REFERENCE_ROOT", "/root/reference")

_STUB_TOPLEVEL = {
    "commonroad", "commonroad_dc", "commonroad_route_planner", "omegaconf", "methodtools", "shapely",
    "frenetix", "prediction", "onnxruntime", "vehiclemodels", "pygeos", "imageio", "triangle", "wale_net",
    "matplotlib", "rich", "pandas_stub_never",
}


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "frenetix_motion_planner"))


class _AnyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any


class _Any(metaclass=_AnyMeta):
    """Dummy class: any attribute resolves, any call returns another dummy."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any()

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Any


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in _STUB_TOPLEVEL and not _really_importable(top):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_REAL_CACHE = {}


def _really_importable(top):
    if top not in _REAL_CACHE:
        _REAL_CACHE[top] = False
        for finder in sys.meta_path:
            if isinstance(finder, _StubFinder):
                continue
            try:
                if finder.find_spec(top, None) is not None:
                    _REAL_CACHE[top] = True
                    break
            except Exception:
                pass
    return _REAL_CACHE[top]


def make_valid_orientation(angle):
    two_pi = 2.0 * np.pi
    while angle > two_pi:
        angle = angle - two_pi
    while angle < -two_pi:
        angle = angle + two_pi
    return angle


_installed = False


def install():
    """Make `import frenetix_motion_planner...` work against /root/reference."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True
    sys.meta_path.append(_StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import numbers

    validity = _StubModule("commonroad.common.validity")
    validity.is_natural_number = lambda x: isinstance(x, numbers.Integral) and x >= 0
    validity.is_positive = lambda x: isinstance(x, numbers.Real) and x > 0
    validity.is_real_number = lambda x: isinstance(x, numbers.Real)
    validity.is_real_number_vector = lambda x, length=None: (
        isinstance(x, (list, tuple, np.ndarray)) and all(isinstance(e, numbers.Real) for e in np.asarray(x).ravel()))
    import commonroad  # noqa: F401  (stub)
    import commonroad.common  # noqa: F401
    sys.modules["commonroad.common.validity"] = validity
    sys.modules["commonroad.common"].validity = validity

    util = _StubModule("commonroad.common.util")
    util.make_valid_orientation = make_valid_orientation
    sys.modules["commonroad.common.util"] = util
    sys.modules["commonroad.common"].util = util

    mt = _StubModule("methodtools")
    mt.lru_cache = lambda *a, **k: (lambda f: f)
    sys.modules["methodtools"] = mt

    import scipy.integrate
    if not hasattr(scipy.integrate, "simps"):
        scipy.integrate.simps = scipy.integrate.simpson
    _installed = True


class NumpyProjection:
    """Independent numpy restatement of the normative (s,d)->(x,y) projection: foot point
    on the polyline segment containing s, offset d along the normalised, linearly interpolated
    vertex normal; None outside [ref_pos[0], ref_pos[-1]]."""

    def __init__(self, ref_xy, ref_pos, ref_theta, ref_curv, ref_curv_d):
        self.ref_xy = np.asarray(ref_xy, dtype=np.float64)
        self.ref_pos = np.asarray(ref_pos, dtype=np.float64)
        self.ref_theta = np.asarray(ref_theta, dtype=np.float64)
        self.ref_curv = np.asarray(ref_curv, dtype=np.float64)
        self.ref_curv_d = np.asarray(ref_curv_d, dtype=np.float64)
        self.normals = vertex_normals(self.ref_xy)

    def convert_to_cartesian_coords(self, s, d):
        rp = self.ref_pos
        if not (rp[0] <= s <= rp[-1]):
            return None
        k = int(np.searchsorted(rp, s, side="right")) - 1
        k = min(max(k, 0), len(rp) - 2)
        lam = (s - rp[k]) / (rp[k + 1] - rp[k])
        p = self.ref_xy[k] + lam * (self.ref_xy[k + 1] - self.ref_xy[k])
        n = self.normals[k] + lam * (self.normals[k + 1] - self.normals[k])
        nn = math.sqrt(n[0] * n[0] + n[1] * n[1])
        return np.array([p[0] + d * (n[0] / nn), p[1] + d * (n[1] / nn)])


def vertex_normals(ref_xy):
    """n_i = left normal of the unit tangent (P[i+1]-P[i-1]) (one-sided at the ends)."""
    P = np.asarray(ref_xy, dtype=np.float64)
    t = np.empty_like(P)
    t[1:-1] = P[2:] - P[:-2]
    t[0] = P[1] - P[0]
    t[-1] = P[-1] - P[-2]
    nrm = np.sqrt(t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1])
    t = t / nrm[:, None]
    return np.stack([-t[:, 1], t[:, 0]], axis=1)


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _ListQueue:
    def __init__(self):
        self.items = []

    def put(self, x):
        self.items.append(x)


def make_planner(prob):
    """prob: dict with the keys documented in gen_golden.py (same names as FxProblem)."""
    install()
    from frenetix_motion_planner.reactive_planner import ReactivePlannerPython
    from frenetix_motion_planner.sampling_matrix import SamplingHandler
    from frenetix_motion_planner.cost_functions.cost_function import AdaptableCostFunction
    import frenetix_motion_planner.cost_functions.partial_cost_functions as pcf

    rp = object.__new__(ReactivePlannerPython)
    rp.horizon = prob["horizon"]
    rp.dT = prob["dt"]
    rp.N = int(prob["horizon"] / prob["dt"])
    veh = prob["vehicle"]
    rp.vehicle_params = _Obj(**veh)
    rp._multiproc = False
    rp._num_workers = 1
    rp._LOW_VEL_MODE = bool(prob["low_vel_mode"])
    rp._draw_traj_set = bool(prob["draw_traj_set"])
    rp._kinematic_debug = bool(prob["kinematic_debug"])
    rp.save_all_traj = False
    rp.occlusion_module = None
    rp.msg_logger = logging.getLogger("fx_ref_harness")
    rp.msg_logger.setLevel(logging.CRITICAL)
    rp.coordinate_system = NumpyProjection(prob["ref_xy"], prob["ref_pos"], prob["ref_theta"], prob["ref_curv"],
                                           prob["ref_curv_d"])
    rp.x_0 = _Obj(orientation=prob["x0_orientation"], velocity=prob.get("x0_velocity", prob["x0_lon"][1]),
                  time_step=0, position=np.zeros(2))
    rp.x_cl = (list(prob["x0_lon"]), list(prob["x0_lat"]))
    rp.desired_velocity = prob["v_des"]
    rp.predictions = prob.get("predictions")
    rp.scenario = _Obj(obstacles=prob.get("scenario_obstacles", []))
    rp.reach_set = None
    rp._sampling_min = prob.get("sampling_level", 2)
    rp._sampling_max = rp._sampling_min + 1
    sh = SamplingHandler(dt=rp.dT, max_sampling_number=max(rp._sampling_max, 5), t_min=prob["t_min"],
                         horizon=rp.horizon, delta_d_min=prob["d_min"], delta_d_max=prob["d_max"], d_ego_pos=False)
    sh.set_v_sampling(prob["v_min"], prob["v_max"])
    if prob.get("custom_sampling"):
        # an explicit (t, v, d) grid instead of a level of the handler (BASELINE configs 2 / 3 / 5 are such grids): the reference's
        # loops take whatever `to_range(level)` hands them -- sets, as its own Sampling classes return them (sampling_matrix.py:
        # `self._sampling_vec[level]`), so the candidate order is CPython's iteration order of these sets
        cs_ = prob["custom_sampling"]

        class _Fixed:
            def __init__(self, values):
                self._set = set(float(x) for x in values)

            def to_range(self, level=0):
                return set(self._set)

        sh.t_sampling, sh.v_sampling, sh.d_sampling = _Fixed(cs_["t"]), _Fixed(cs_["v"]), _Fixed(cs_["d"])
    rp.sampling_handler = sh

    cf = object.__new__(AdaptableCostFunction)
    weights = {k: w for k, w in prob["cost_weights"].items() if w != 0}
    names = sorted(weights.keys())
    cf.cost_weights = weights
    cf.cost_weights_names = names
    cf.functions = {n: getattr(pcf, n + "_costs") for n in names}
    cf.rp = rp
    cf.scenario = rp.scenario
    cf.desired_speed = rp.desired_velocity
    cf.predictions = rp.predictions
    cf.reachset = None
    rp.cost_function = cf
    return rp


def run_reference(prob):
    """Runs _create_trajectory_bundle -> check_feasibility -> sort of the reference and returns
    plain numpy outputs (the golden-vector payload)."""
    rp = make_planner(prob)
    level = rp._sampling_min
    S = rp.N + 1
    t_order = np.array(list(rp.sampling_handler.t_sampling.to_range(level)), dtype=np.float64)
    d_order = np.array(list(rp.sampling_handler.d_sampling.to_range(level).union({rp.x_cl[1][0]})), dtype=np.float64)
    if prob.get("stop_point_s") is not None:
        # stop-point sampling (reactive_planner.py:628-671): the longitudinal samples are end positions
        bundle = rp._create_end_point_trajectory_bundle(rp.x_cl[0], rp.x_cl[1], prob["stop_point_s"], rp.cost_function,
                                                        level)
        v_order = np.array(list(rp.sampling_handler.s_sampling.to_range(level)), dtype=np.float64)
    else:
        v_order = np.array(list(rp.sampling_handler.v_sampling.to_range(level)), dtype=np.float64)
        bundle = rp._create_trajectory_bundle(rp.x_cl[0], rp.x_cl[1], rp.cost_function, samp_level=level)
    trajs = list(bundle.trajectories)
    C = len(trajs)
    assert C == len(t_order) * len(v_order) * len(d_order), (C, len(t_order), len(v_order), len(d_order))
    coeff_lon = np.array([t.trajectory_long.coeffs for t in trajs])
    coeff_lat = np.array([t.trajectory_lat.coeffs for t in trajs])
    tau_lat = np.array([t.trajectory_lat.delta_tau for t in trajs])

    # per-candidate reason flags: one check_feasibility call per candidate, histogram via queue_2
    reasons = np.full(C, -1, dtype=np.int32)
    if rp._kinematic_debug:
        rp._multiproc = True
        for g, tr in enumerate(trajs):
            qa, qb = _ListQueue(), _ListQueue()
            rp.check_feasibility([tr], qa, qb)
            h = np.asarray(qb.items[0]).astype(np.int64)
            assert h.max() <= 1
            reasons[g] = int(sum(int(b) << r for r, b in enumerate(h)))
        rp._multiproc = False

    # histogram comes only through queue_2 (reactive_planner.py:571-575)
    rp._multiproc = True
    q1, q2 = _ListQueue(), _ListQueue()
    kd_saved = rp._kinematic_debug
    rp._kinematic_debug = True if kd_saved else False
    rp.check_feasibility(trajs, q1, q2)
    returned = q1.items[0]
    hist = np.asarray(q2.items[0], dtype=np.int64) if q2.items else None
    rp._multiproc = False
    if hist is None:
        # kinematic_debug False: re-run only to capture the histogram through queue_2 -- changes the
        # break semantics, so not available; tests derive it from per-candidate reasons instead.
        hist = np.full(11, -1, dtype=np.int64)

    valid = np.zeros(C, dtype=bool)
    feasible = np.zeros(C, dtype=bool)
    ret = np.zeros(C, dtype=bool)
    has_cart = np.zeros(C, dtype=bool)
    traj_len = np.zeros(C, dtype=np.int32)
    planes = np.zeros((C, 14, S))
    for t in returned:
        g = t.uniqueId
        ret[g] = True
    for g, t in enumerate(trajs):
        valid[g] = bool(t.valid)
        feasible[g] = bool(t.feasible)
        if ret[g] and getattr(t, "_cartesian", None) is not None:
            has_cart[g] = True
            c, k = t.cartesian, t.curvilinear
            planes[g] = np.stack([c.x, c.y, c.theta, c.v, c.a, c.kappa, c.kappa_dot,
                                  k.s, k.d, k.theta, k.s_dot, k.s_ddot, k.d_dot, k.d_ddot])
            traj_len[g] = t.actual_traj_length

    # pool + sort exactly as _get_optimal_trajectory (reactive_planner.py:229-253)
    from frenetix_motion_planner.trajectories import TrajectoryBundle
    feas = [o for o in returned if o.valid is True and o.feasible is True]
    b2 = TrajectoryBundle(list(returned) if rp._draw_traj_set else feas, cost_function=rp.cost_function,
                          multiproc=False, num_workers=1)
    b2.sort()
    sorted_all = [t.uniqueId for t in b2.trajectories]
    if rp._draw_traj_set:
        walk = [t.uniqueId for t in b2.trajectories if t.feasible is True]
    else:
        walk = list(sorted_all)
    names = rp.cost_function.cost_weights_names
    cost = np.zeros(C)
    costmap = np.zeros((C, len(names)))
    costed = np.zeros(C, dtype=bool)
    for t in b2.trajectories:
        g = t.uniqueId
        costed[g] = True
        cost[g] = t.cost
        costmap[g] = [t.costMap[n][0] for n in names]
    return dict(t_order=t_order, v_order=v_order, d_order=d_order, coeff_lon=coeff_lon, coeff_lat=coeff_lat,
                tau_lat=tau_lat, valid=valid, feasible=feasible, returned=ret, has_cart=has_cart, traj_len=traj_len,
                planes=planes, hist=hist, reasons=reasons, cost=cost, costmap=costmap, costed=costed,
                sorted_ids=np.array(sorted_all, dtype=np.int64), walk_ids=np.array(walk, dtype=np.int64),
                cost_names=np.array(names), n_returned=len(returned), n_feasible=len(feas))
