"""PlanInputs: the shared, candidate-independent inputs of one plan step, packed for the C-ABI.

Everything `ReactivePlannerPython.plan()` reads from `self` (reactive_planner.py:67-130,
planner.py:172-217) ends up here as plain numbers and C-contiguous f64 arrays; `as_struct()` yields the
`FxProblem` of include/fxplan.h with pointers into arrays this object keeps alive.
"""
import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _abi
from .coordinate_system import CoordinateSystem, simpson_even_correction, time_power_table


@dataclass
class VehicleParams:
    """The fields of config_sim.vehicle the hot path reads (configuration.py:58-83).  Defaults are the
    commonroad-vehicle-models id 2 (BMW 320i) table quoted in SURVEY.md 8c -- plain inputs, override
    them exactly like configurations/simulation/vehicle.yaml does."""
    length: float = 4.508
    width: float = 1.610
    wheelbase: float = 2.5789
    wb_rear_axle: float = 1.4227
    a_max: float = 11.5
    v_max: float = 50.8
    v_switch: float = 7.319
    delta_max: float = 1.066
    v_delta_max: float = 0.4

    @property
    def kappa_max(self) -> float:
        return float(np.tan(self.delta_max) / self.wheelbase)  # reactive_planner.py:492

    def as_struct(self) -> _abi.FxVehicle:
        return _abi.FxVehicle(self.a_max, self.v_switch, self.delta_max, self.wheelbase, self.length, self.width,
                              self.wb_rear_axle, self.kappa_max)


DEFAULT_COST_WEIGHTS = {  # configurations/frenetix_motion_planner/cost.yaml:3-17 (non-zero entries)
    "lateral_jerk": 0.2, "longitudinal_jerk": 0.2, "velocity_offset": 1.0,
    "distance_to_reference_path": 5.0, "prediction": 0.2,
}


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _ptr(a, typ=C.c_double):
    return a.ctypes.data_as(C.POINTER(typ)) if a is not None and a.size else C.POINTER(typ)()


_TPOW_CACHE: Dict[tuple, tuple] = {}


def _tpow_cached(dt: float, S: int):
    """(table, pointer, Simpson correction) of a (dt, S) pair -- the same for every plan step of a planner"""
    key = (float(dt), int(S))
    c = _TPOW_CACHE.get(key)
    if c is None:
        t = _f64(time_power_table(dt, S))
        c = _TPOW_CACHE[key] = (t, _ptr(t), (C.c_double * 3)(*simpson_even_correction(dt)))
    return c


def _ref_cached(cs: CoordinateSystem):
    """kernel-side arrays of a coordinate system and their pointers, built once per reference path"""
    c = getattr(cs, "_fx_kernel_arrays", None)
    if c is None:
        arr = {k: _f64(v) for k, v in dict(
            x=cs.reference[:, 0], y=cs.reference[:, 1], nx=cs.normals[:, 0], ny=cs.normals[:, 1], pos=cs.ref_pos,
            theta=cs.ref_theta, curv=cs.ref_curv, curv_d=cs.ref_curv_d).items()}
        c = cs._fx_kernel_arrays = (arr, {k: _ptr(v) for k, v in arr.items()})
    return c


def _dict_ptrs(d: dict, spec):
    """pointers into the arrays of a packed dict (predictions, road boundary), cached inside the dict"""
    c = d.get("_ptrs")
    if c is None:
        c = d["_ptrs"] = {k: _ptr(d[k], t) for k, t in spec}
    return c


_BOUND_UIDS = __import__("itertools").count(1)
_COST_MEMO: dict = {}   # id(weights dict) -> (the dict, (len, sum of values), (names, ids, weights))
MAX_OBSTACLES = 256  # obstacles per agent (FX_MAX_OBSTACLES; beyond 64 the step runs on the generic kernel)


def pack_lanelets(lanelets) -> dict:
    """The lanelets the lane_center_offset cost reads (partial_cost_functions.py:91-117), in the order given -- the lanelet
    network's: per lanelet the closed outline (left vertices, then the right ones reversed: commonroad-io's Lanelet.polygon), its
    bounding box and the centre polyline.  `lanelets`: objects with left_vertices / right_vertices (/ center_vertices) [n][2] --
    commonroad_xml.Lanelet, commonroad-io's Lanelet -- or a Scenario / dict of them.  fxplan.h FxProblem.n_lane defines how
    they are read."""
    if hasattr(lanelets, "lanelets"):
        lanelets = lanelets.lanelets
    if isinstance(lanelets, dict):
        lanelets = list(lanelets.values())
    lanelets = list(lanelets)
    n = len(lanelets)
    bbox = np.zeros((n, 4))
    poly_off, ctr_off = np.zeros(n + 1, np.int32), np.zeros(n + 1, np.int32)
    polys, ctrs = [], []
    for k, ll in enumerate(lanelets):
        left, right = _f64(ll.left_vertices).reshape(-1, 2), _f64(ll.right_vertices).reshape(-1, 2)
        poly = np.vstack([left, right[::-1]])
        cv = getattr(ll, "center_vertices", None)
        ctr = _f64(cv).reshape(-1, 2) if cv is not None else 0.5 * (left + right)
        bbox[k] = (poly[:, 0].min(), poly[:, 0].max(), poly[:, 1].min(), poly[:, 1].max())
        polys.append(poly)
        ctrs.append(ctr)
        poly_off[k + 1] = poly_off[k] + len(poly)
        ctr_off[k + 1] = ctr_off[k] + len(ctr)
    return dict(n=n, bbox=_f64(bbox), poly_off=poly_off, poly=_f64(np.vstack(polys)) if n else np.zeros((0, 2)), ctr_off=ctr_off,
                ctr=_f64(np.vstack(ctrs)) if n else np.zeros((0, 2)), uid=next(_BOUND_UIDS))


class PackedPredictions:
    """Predictions that arrive packed (the arrays `pack_predictions` produces) together with a way to build the reference's
    dict form on demand.  A batch of agents shares almost all of its predictions -- everything but the agent itself -- so the
    simulation packs ALL entries once per step (covariance inverses, OBB-sum hulls) and hands every agent the rows without its
    own (`subset`): a few fancy-index copies instead of a dict walk, inversions and hull building per agent.  Behaves like the
    dict for whoever reads it (the logger: `items()`, `len`, truthiness); the dict is built at the first such access."""

    def __init__(self, packed: dict, make_dict=None):
        self.packed = packed
        self._make, self._dict = make_dict, None

    @staticmethod
    def subset(packed: dict, rows) -> dict:
        """the packed arrays of the obstacles `rows` (in that order) of `packed`"""
        rows = np.asarray(rows, dtype=np.intp)
        if len(rows) == 0:
            z = np.zeros(0)
            return dict(K=0, P=0, pos=z, cov_inv=z, npred=np.zeros(0, np.int32), hull=z, nhull=np.zeros(0, np.int32))
        return dict(K=len(rows), P=packed["P"], pos=packed["pos"][rows], cov_inv=packed["cov_inv"][rows], npred=packed["npred"][rows],
                    hull=packed["hull"][rows], nhull=packed["nhull"][rows])

    def as_dict(self) -> dict:
        if self._dict is None:
            self._dict = self._make() if self._make is not None else {}
        return self._dict

    def __len__(self):   # entries of the dict form: a muted row (MultiAgentSimulation.packed_predictions_for) is not one
        return int(self.packed["K"]) - ("muted_row" in self.packed)

    def __bool__(self):
        return len(self) > 0

    def __iter__(self):
        return iter(self.as_dict())

    def __getitem__(self, k):
        return self.as_dict()[k]

    def __contains__(self, k):
        return k in self.as_dict()

    def keys(self):
        return self.as_dict().keys()

    def items(self):
        return self.as_dict().items()

    def values(self):
        return self.as_dict().values()

    def get(self, k, default=None):
        return self.as_dict().get(k, default)


def pack_predictions(predictions: Optional[Dict], n_samples: int, build_hulls):
    """predictions dict {id: {'pos_list' [P,2], 'cov_list' [P,2,2], 'orientation_list' [P],
    'shape': {'length','width'}}} (prediction_helpers.py:209-261) -> packed arrays.

    Iteration order = dict order, as in get_inv_mahalanobis_dist (collision_probability.py:276-279).
    cov is inverted with np.linalg.inv exactly like :281.  build_hulls(n, pos, yaw, L, W) returns the
    OBB-sum hulls [n-1, 6] (fx_build_obstacle_hulls).  Collision uses the first min(S, P_k)
    predictions (collision_check.py:150)."""
    if isinstance(predictions, PackedPredictions):
        return predictions.packed
    if not predictions:
        z = np.zeros(0)
        return dict(K=0, P=0, pos=z, cov_inv=z, npred=np.zeros(0, np.int32), hull=z, nhull=np.zeros(0, np.int32))
    pack_dict = getattr(build_hulls, "pack_dict", None)
    if pack_dict is not None:   # the dict walked in C (csrc/fx_host_ext.c); None: something needs converting first
        packed = pack_dict(predictions, n_samples, MAX_OBSTACLES)
        if packed is not None:
            return packed
    keys = list(predictions.keys())
    K = len(keys)
    if K > MAX_OBSTACLES:
        raise ValueError(f"{K} predicted obstacles: the engine takes at most {MAX_OBSTACLES} per agent -- cull the predictions on "
                         "the host (e.g. by distance to the reachable set, get_obstacles_in_radius) before packing")
    # only the first n_samples predictions are ever read (prediction i-1 pairs with ego step i <= N, hulls use min(S, n)):
    # a 100-step predictor does not enlarge the tables
    P = max(2, min(n_samples, max(len(predictions[k]["pos_list"]) for k in keys)))
    pack = getattr(build_hulls, "pack", None)
    if pack is not None:   # the library pads, inverts and builds the hulls in one call (fx_pack_predictions)
        entries = []
        for k in keys:
            pr = predictions[k]
            pl = _f64(pr["pos_list"]).reshape(-1, 2)
            hulls = "orientation_list" in pr and "shape" in pr
            sh = pr["shape"] if hulls else None
            entries.append((pl, _f64(pr["cov_list"]).reshape(-1, 2, 2) if len(pl) else pl.reshape(0, 2, 2),
                            _f64(pr["orientation_list"]).reshape(-1) if hulls else None,
                            float(sh["length"]) if hulls else 0.0, float(sh["width"]) if hulls else 0.0))
        return pack(entries, P, n_samples)
    pos = np.zeros((K, P, 2))
    cov_inv = np.zeros((K, P, 4))
    npred = np.zeros(K, np.int32)
    hull = np.zeros((K, P - 1, 6))
    nhull = np.zeros(K, np.int32)
    yaw = np.zeros((K, P))
    n_use = np.zeros(K, np.int32)
    shape = np.zeros((2, K))
    covs, rows = [], []
    for i, k in enumerate(keys):
        pr = predictions[k]
        pl = np.asarray(pr["pos_list"], dtype=np.float64).reshape(-1, 2)
        n_all = len(pl)
        # len(pos_list) decides which ego steps see the obstacle (i < len, collision_probability.py:287): keep the real
        # length, store what can be read
        npred[i] = n_all
        if n_all == 0:
            continue
        n = min(n_all, P)
        pos[i, :n] = pl[:n]
        covs.append(np.asarray(pr["cov_list"], dtype=np.float64)[:n].reshape(n, 2, 2))
        rows.append((i, n))
        if build_hulls is not None and "orientation_list" in pr and "shape" in pr:
            n_use[i] = min(n_samples, n)
            yaw[i, :n_use[i]] = _f64(pr["orientation_list"])[:n_use[i]]
            shape[0, i], shape[1, i] = float(pr["shape"]["length"]), float(pr["shape"]["width"])
    if covs:  # ONE stacked inversion (gesv per matrix, as np.linalg.inv of each list gives; the library's fx_invert_cov2 is
        # that arithmetic bit for bit without the LAPACK call overhead)
        inv2 = getattr(build_hulls, "invert_cov2", None)
        stacked = np.concatenate(covs)
        inv = inv2(stacked) if inv2 is not None else np.linalg.inv(stacked).reshape(-1, 4)
        o = 0
        for i, n in rows:
            cov_inv[i, :n] = inv[o:o + n]
            o += n
    if build_hulls is not None and n_use.any():
        batch = getattr(build_hulls, "batch", None)
        if batch is not None:   # the library builds every obstacle's hulls in one call
            batch(n_use, pos, yaw, shape[0], shape[1], hull, nhull)
        else:
            for i in np.nonzero(n_use)[0]:
                h = build_hulls(int(n_use[i]), _f64(pos[i, :n_use[i]]), _f64(yaw[i, :n_use[i]]), shape[0, i], shape[1, i])
                nhull[i] = len(h)
                hull[i, :len(h)] = h
    return dict(K=K, P=P, pos=_f64(pos), cov_inv=_f64(cov_inv), npred=npred, hull=_f64(hull), nhull=nhull)


def pack_road_boundary(segments, cs: CoordinateSystem, vehicle: "VehicleParams", d_reach: float, max_len: float = 4.0):
    """Road boundary segments [n][4] = (ax, ay, bx, by) -> what the engine takes (planner.py:362-381, 550-565).

    Pieces no longer than max_len as (mid x, mid y, half dx, half dy), and for every reference knot k the pieces whose
    midpoint is within reach + half length of it (CSR bins).  reach bounds the distance between knot k and any point of
    an ego footprint whose foot point lies on reference segment k and whose centre is laterally within d_reach:
    longest reference segment + d_reach + wb_rear_axle + half diagonal.  (The C-ABI twin is fx_build_boundary_bins.)"""
    seg = _f64(segments).reshape(-1, 4)
    a, b = seg[:, :2], seg[:, 2:]
    length = np.linalg.norm(b - a, axis=1)
    n_sub = np.maximum(1, np.ceil(length / max_len)).astype(np.int64)
    idx = np.repeat(np.arange(len(seg)), n_sub)
    k = np.concatenate([np.arange(n) for n in n_sub]) if len(seg) else np.zeros(0, dtype=np.int64)
    t0 = (k / n_sub[idx])[:, None]
    t1 = ((k + 1) / n_sub[idx])[:, None]
    pa = a[idx] + t0 * (b[idx] - a[idx])
    pb = a[idx] + t1 * (b[idx] - a[idx])
    piece = _f64(np.concatenate([0.5 * (pa + pb), 0.5 * (pb - pa)], axis=1)).reshape(-1, 4)
    ref = cs.reference
    seg_len = np.linalg.norm(np.diff(ref, axis=0), axis=1)
    reach = float(seg_len.max() + d_reach + vehicle.wb_rear_axle + 0.5 * np.hypot(vehicle.length, vehicle.width))
    half = np.linalg.norm(piece[:, 2:], axis=1)
    bins = np.zeros(len(ref) + 1, dtype=np.int32)
    items = []
    for kk in range(len(ref)):
        near = np.nonzero(np.linalg.norm(piece[:, :2] - ref[kk], axis=1) <= reach + half)[0]
        items.append(near.astype(np.int32))
        bins[kk + 1] = bins[kk] + len(near)
    item = np.concatenate(items).astype(np.int32) if items else np.zeros(0, dtype=np.int32)
    return dict(n=len(piece), piece=piece, bin=bins, item=np.ascontiguousarray(item), d_reach=float(d_reach), reach=reach)


@dataclass
class PlanInputs:
    # horizon / mode
    N: int
    dt: float
    low_vel_mode: bool
    x0_lon: np.ndarray
    x0_lat: np.ndarray
    x0_orientation: float
    v_des: float
    vehicle: VehicleParams
    coordinate_system: CoordinateSystem
    # sampling: ordered ranges or an explicit C x 13 matrix
    t_samp: Optional[np.ndarray] = None
    v_samp: Optional[np.ndarray] = None
    d_samp: Optional[np.ndarray] = None
    sampling_matrix: Optional[np.ndarray] = None
    # stop-point sampling (reactive_planner.py:628-671): v_samp then holds the sampled END POSITIONS s and the
    # longitudinal polynomial is the quintic to (s, 0, 0)
    stop_point: bool = False
    cost_weights: Dict[str, float] = field(default_factory=lambda: dict(DEFAULT_COST_WEIGHTS))
    draw_traj_set: bool = False
    kinematic_debug: bool = False
    write_bundle: bool = True
    write_costmap: bool = True
    collision: bool = True
    # packed predictions (pack_predictions) and distance_to_obstacles positions
    obstacles: Optional[dict] = None
    dto_pos: Optional[np.ndarray] = None
    # road boundary: segments [n][4] = (ax, ay, bx, by); the ego footprint must not touch them (planner.py:362-381)
    road_boundary: Optional[np.ndarray] = None
    # lanelets of the lane_center_offset cost: pack_lanelets(...) (or anything pack_lanelets takes)
    lanelets: Optional[dict] = None
    # candidate shard of the global grid handled by this engine (multi-GPU); None = everything
    shard: Optional[tuple] = None

    def __post_init__(self):
        self.x0_lon = _f64(self.x0_lon)
        self.x0_lat = _f64(self.x0_lat)
        S = self.N + 1
        self._tpow = _tpow_cached(self.dt, S)[0]
        cs = self.coordinate_system
        self._ref = _ref_cached(cs)[0]
        if self.sampling_matrix is not None:
            self.sampling_matrix = _f64(self.sampling_matrix)
            if self.sampling_matrix.ndim != 2 or self.sampling_matrix.shape[1] != 13:
                raise ValueError("sampling_matrix must be C x 13")
            if self.stop_point:
                raise ValueError("stop-point sampling takes ranges (the C x 13 matrix has no end-position column)")
        else:
            if self.t_samp is None or self.v_samp is None or self.d_samp is None:
                raise ValueError("either sampling_matrix or t_samp/v_samp/d_samp are required")
            self.t_samp, self.v_samp, self.d_samp = _f64(self.t_samp), _f64(self.v_samp), _f64(self.d_samp)
        # a planner hands over the same weights dict step after step: names / ids / weights are derived once per dict object
        # (guarded by the exact (name, weight) items against in-place edits -- ten entries: swapping two weights or moving weight
        # from one term to another keeps length and sum, and would otherwise upload the old cost function)
        cw = self.cost_weights
        sig = tuple(cw.items())
        memo = _COST_MEMO.get(id(cw))
        if memo is not None and memo[0] is cw and memo[1] == sig:
            self.cost_names, self._cost_id, self._cost_w = memo[2]
        else:
            bad = [n for n, w in cw.items() if w != 0 and n not in _abi.COST_ID]
            if bad:
                raise NotImplementedError(f"cost terms outside the hot-path scope: {bad}")
            names = sorted(n for n, w in cw.items() if w != 0)  # cost_function.py:52-60
            self.cost_names = names
            self._cost_id = np.array([_abi.COST_ID[n] for n in names], dtype=np.int32)
            self._cost_w = _f64([cw[n] for n in names])
            if len(_COST_MEMO) > 256:
                _COST_MEMO.clear()
            _COST_MEMO[id(cw)] = (cw, sig, (self.cost_names, self._cost_id, self._cost_w))
        if self.obstacles is None:
            self.obstacles = pack_predictions(None, S, None)
        self._dto = _f64(self.dto_pos).reshape(-1, 2) if self.dto_pos is not None else np.zeros((0, 2))
        if self.lanelets is not None and not (isinstance(self.lanelets, dict) and "poly_off" in self.lanelets):
            self.lanelets = pack_lanelets(self.lanelets)
        self._bound = None
        if self.road_boundary is not None and len(self.road_boundary):
            if isinstance(self.road_boundary, dict):
                self._bound = self.road_boundary  # already packed (pack_road_boundary), shared between steps
            else:
                d_all = np.abs(self.d_samp) if self.sampling_matrix is None else np.abs(self.sampling_matrix[:, 10])
                d_reach = 1.5 * max(float(d_all.max()) if len(d_all) else 0.0, abs(float(self.x0_lat[0]))) + 2.0
                self._bound = pack_road_boundary(self.road_boundary, cs, self.vehicle, d_reach)

    @property
    def n_samples(self) -> int:
        return self.N + 1

    @property
    def n_candidates_global(self) -> int:
        if self.sampling_matrix is not None:
            return int(self.sampling_matrix.shape[0])
        return int(len(self.t_samp) * len(self.v_samp) * len(self.d_samp))

    @property
    def n_candidates(self) -> int:
        """candidates evaluated by this engine (the shard, or the whole grid)"""
        return int(self.shard[1]) if self.shard is not None else self.n_candidates_global

    @property
    def shard_begin(self) -> int:
        return int(self.shard[0]) if self.shard is not None else 0

    @property
    def mode(self) -> int:
        m = 0
        if self.draw_traj_set:
            m |= _abi.FX_MODE_DRAW_TRAJ_SET
        if self.kinematic_debug:
            m |= _abi.FX_MODE_KINEMATIC_DEBUG
        if self.write_bundle:
            m |= _abi.FX_MODE_WRITE_BUNDLE
        if self.write_costmap:
            m |= _abi.FX_MODE_WRITE_COSTMAP
        if self.collision and self.obstacles["K"] > 0:
            m |= _abi.FX_MODE_COLLISION
        if self._bound is not None and self._bound["n"] > 0:
            m |= _abi.FX_MODE_ROAD_BOUNDARY
        if getattr(self.coordinate_system, "pseudo_normal", False):
            m |= _abi.FX_MODE_PROJ_PSEUDO_NORMAL
        return m

    def structure_key(self):
        """Everything of a plan step that fx_update_state cannot change: two inputs with equal keys differ only in the ego
        state, the desired velocity, the sampling values and the obstacle predictions (same counts)."""
        k = self.__dict__.get("_skey")   # (without the shard, which a batch sets after construction; kept by next_step)
        if k is None:
            o = self.obstacles
            k = self._skey = (
                self.N, self.dt, self.mode, bool(self.stop_point),
                (self.vehicle.length, self.vehicle.width, self.vehicle.wheelbase, self.vehicle.wb_rear_axle, self.vehicle.a_max,
                 self.vehicle.v_switch, self.vehicle.delta_max),
                getattr(self.coordinate_system, "uid", None) or id(self.coordinate_system),
                None if self.sampling_matrix is not None else (len(self.t_samp), len(self.v_samp), len(self.d_samp)),
                tuple(self.cost_names), self._cost_w.tobytes(), int(o["K"]), int(o["P"]), self._dto.tobytes(),
                None if self._bound is None else self._bound.setdefault("uid", next(_BOUND_UIDS)),
                None if self.lanelets is None else self.lanelets.setdefault("uid", next(_BOUND_UIDS)))
        return k + (self.shard,)

    def next_step(self, *, low_vel_mode, x0_lon, x0_lat, x0_orientation, v_des, t_samp, v_samp, d_samp, obstacles):
        """The inputs of the same planner's NEXT plan step: everything a closed-loop step changes -- ego state, desired velocity,
        sampling values, predictions -- replaced on a shallow copy; horizon, vehicle, reference, cost function, flags, road
        boundary stay (the caller vouches for that: ReactivePlannerHip._inputs_for_level checks what it can replace).  Skips
        the conversions and look-ups of __post_init__ that cannot have changed; the structure key survives when the new
        arrays have the old lengths and the predictions the old (K, P)."""
        prev = self.__dict__
        new = object.__new__(PlanInputs)
        d = new.__dict__
        d.update(prev)
        o_prev = prev["obstacles"]
        if obstacles is None:
            obstacles = o_prev if not o_prev["K"] else pack_predictions(None, self.N + 1, None)
        d["low_vel_mode"], d["x0_orientation"], d["v_des"], d["obstacles"], d["shard"] = low_vel_mode, x0_orientation, v_des, obstacles, None
        d["x0_lon"], d["x0_lat"] = _f64(x0_lon), _f64(x0_lat)
        d["t_samp"], d["v_samp"], d["d_samp"] = t, v, dd = _f64(t_samp), _f64(v_samp), _f64(d_samp)
        if (len(t) != len(prev["t_samp"]) or len(v) != len(prev["v_samp"]) or len(dd) != len(prev["d_samp"])
                or obstacles["K"] != o_prev["K"] or obstacles["P"] != o_prev["P"]):
            d.pop("_skey", None)
        return new

    def candidate_params(self, g: int):
        """(t0, t1, s0, ss0, sss0, ss1, sss1, d0, dd0, ddd0, d1, dd1, ddd1) of candidate g -- the
        `sampling_parameters` attribute of frenetix.TrajectorySample (reactive_planner_cpp.py:456)."""
        if self.sampling_matrix is not None:
            return self.sampling_matrix[g].copy()
        nD, nV = len(self.d_samp), len(self.v_samp)
        i_d, i_v, i_t = g % nD, (g // nD) % nV, g // (nD * nV)
        if self.stop_point:  # end state (s1, 0, 0): end velocity and acceleration are 0; s1 = v_samp[i_v]
            return np.array([0.0, self.t_samp[i_t], *self.x0_lon, 0.0, 0.0, *self.x0_lat, self.d_samp[i_d], 0.0, 0.0])
        return np.array([0.0, self.t_samp[i_t], *self.x0_lon, self.v_samp[i_v], 0.0, *self.x0_lat,
                         self.d_samp[i_d], 0.0, 0.0])

    def as_struct(self) -> _abi.FxProblem:
        p = _abi.FxProblem()
        p.N, p.dt, p.mode, p.low_vel_mode = self.N, self.dt, self.mode, int(bool(self.low_vel_mode))
        p.lon_mode = _abi.FX_LON_STOP_POINT if self.stop_point else _abi.FX_LON_VELOCITY_KEEPING
        p.x0_lon = (C.c_double * 3)(*self.x0_lon)
        p.x0_lat = (C.c_double * 3)(*self.x0_lat)
        p.x0_orientation, p.v_des = float(self.x0_orientation), float(self.v_des)
        p.veh = self.vehicle.as_struct()
        tp = _tpow_cached(self.dt, self.N + 1)
        p.tpow = tp[1]
        if self.sampling_matrix is not None:
            p.sampling_matrix, p.n_rows = _ptr(self.sampling_matrix), self.sampling_matrix.shape[0]
            p.nT = p.nV = p.nD = 0
        else:
            p.nT, p.nV, p.nD = len(self.t_samp), len(self.v_samp), len(self.d_samp)
            p.t_samp, p.v_samp, p.d_samp = _ptr(self.t_samp), _ptr(self.v_samp), _ptr(self.d_samp)
            p.n_rows = 0
        r, rp = _ref_cached(self.coordinate_system)
        p.M = len(r["pos"])
        p.ref_x, p.ref_y, p.ref_nx, p.ref_ny = rp["x"], rp["y"], rp["nx"], rp["ny"]
        p.ref_pos, p.ref_theta, p.ref_curv, p.ref_curv_d = rp["pos"], rp["theta"], rp["curv"], rp["curv_d"]
        p.n_cost = len(self._cost_id)
        p.cost_id, p.cost_w = _ptr(self._cost_id, C.c_int32), _ptr(self._cost_w)
        p.simpson_corr = tp[2]
        o = self.obstacles
        p.K, p.P = int(o["K"]), int(o["P"])
        op = _dict_ptrs(o, (("pos", C.c_double), ("cov_inv", C.c_double), ("npred", C.c_int32), ("hull", C.c_double),
                            ("nhull", C.c_int32)))
        p.obs_pos, p.obs_cov_inv, p.obs_npred, p.obs_hull, p.obs_nhull = op["pos"], op["cov_inv"], op["npred"], op["hull"], op["nhull"]
        p.n_dto, p.dto_pos = len(self._dto), _ptr(self._dto)
        if self._bound is not None and self._bound["n"] > 0:
            bd = self._bound
            if len(bd["bin"]) != p.M + 1:
                raise ValueError("road boundary bins were built for a different reference path")
            bp = _dict_ptrs(bd, (("piece", C.c_double), ("bin", C.c_int32), ("item", C.c_int32)))
            p.n_bound, p.bound_piece, p.bound_bin, p.bound_item = bd["n"], bp["piece"], bp["bin"], bp["item"]
            p.bound_d_reach = bd["d_reach"]
        else:
            p.n_bound = 0
        ln = self.lanelets
        if ln is not None and ln["n"] > 0:
            lp = _dict_ptrs(ln, (("bbox", C.c_double), ("poly_off", C.c_int32), ("poly", C.c_double), ("ctr_off", C.c_int32),
                                 ("ctr", C.c_double)))
            p.n_lane, p.lane_bbox, p.lane_poly_off, p.lane_poly = ln["n"], lp["bbox"], lp["poly_off"], lp["poly"]
            p.lane_ctr_off, p.lane_ctr = lp["ctr_off"], lp["ctr"]
        else:
            p.n_lane = 0
        if self.shard is not None:
            b, n = int(self.shard[0]), int(self.shard[1])
            if b < 0 or n < 1 or b + n > self.n_candidates_global:
                raise ValueError(f"shard [{b}, {b + n}) outside the grid of {self.n_candidates_global} candidates")
            p.shard_begin, p.shard_count = b, n
        else:
            p.shard_begin, p.shard_count = 0, 0
        return p
