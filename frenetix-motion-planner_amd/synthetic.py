"""Seeded synthetic inputs of the hot path (SURVEY.md 8d "Common synthetic inputs").

Used by the parity tests, the golden-vector generator, __graft_entry__.smoke() and bench.py.
There is no network and no CommonRoad stack on the GPU box, so reference paths, ego states and
obstacle predictions are generated here; shapes and magnitudes follow the reference's defaults
(configurations/frenetix_motion_planner/planning.yaml, cost.yaml; 5.0 x 2.0 m cars, cov 0.1*I as in
prediction_helpers.py:245).
"""
import numpy as np

from .coordinate_system import CoordinateSystem
from .problem import DEFAULT_COST_WEIGHTS, PlanInputs, VehicleParams, pack_predictions
from .sampling import SamplingHandler, v_sampling_bounds
from .sampling import dense_ranges as _dense_ranges

SEED = 20241008


def reference_polyline(kind: str = "arc", n_knots: int = 400, spacing: float = 0.5, kappa: float = 0.01, knot_jitter: float = 0.0,
                       seed: int = SEED, heading0: float = 0.0):
    """straight | arc (constant curvature) | scurve (curvature sign change) polyline.  knot_jitter > 0 (straight / arc): the
    knots sit at deliberately NON-uniform arc lengths -- segment lengths spacing * (1 +- knot_jitter), seeded -- as a route
    polyline does after spline smoothing (the segment lookup and every interpolation weight then differ from knot to knot).
    heading0: the polyline rotated about its first knot so that it starts with that heading (a route whose heading runs through
    +-pi: the unwrapped reference heading leaves (-pi, pi] while vehicle states keep theirs inside)."""
    if heading0 != 0.0:
        p = reference_polyline(kind, n_knots, spacing, kappa, knot_jitter, seed)
        c, sn = np.cos(heading0), np.sin(heading0)
        return np.stack([c * p[:, 0] - sn * p[:, 1], sn * p[:, 0] + c * p[:, 1]], axis=1)
    s = np.arange(n_knots) * spacing
    if knot_jitter > 0.0:
        if kind == "scurve":
            raise ValueError("knot_jitter is for the straight / arc references")
        w = np.random.default_rng([seed, 77]).uniform(1.0 - knot_jitter, 1.0 + knot_jitter, n_knots - 1)
        s = np.concatenate([[0.0], np.cumsum(spacing * w)])
    if kind == "straight":
        return np.stack([s, np.zeros_like(s)], axis=1)
    if kind == "arc":
        r = 1.0 / kappa
        return np.stack([r * np.sin(s * kappa), r * (1 - np.cos(s * kappa))], axis=1)
    if kind == "scurve":
        L = s[-1]
        k = kappa * 2.0 * np.cos(2 * np.pi * s / L)  # curvature changes sign twice
        th = np.concatenate([[0.0], np.cumsum(0.5 * (k[1:] + k[:-1]) * spacing)])
        x = np.concatenate([[0.0], np.cumsum(np.cos(0.5 * (th[1:] + th[:-1])) * spacing)])
        y = np.concatenate([[0.0], np.cumsum(np.sin(0.5 * (th[1:] + th[:-1])) * spacing)])
        return np.stack([x, y], axis=1)
    raise ValueError(kind)


def dense_ranges(n_t: int, n_v: int, n_d: int, v0: float, veh: VehicleParams, horizon: float, dt: float, d0: float,
                 t_min: float = 1.1, d_min: float = -3.0, d_max: float = 3.0):
    """Dense grid of sampling.dense_ranges over the planner's velocity range for v0 (planner.py:304-306).
    BASELINE config 2: n_t=19, n_v=51, n_d=51 -> 19 x 51 x 52 = 50 388 candidates."""
    v_lo, v_hi = v_sampling_bounds(v0, veh.a_max, horizon, veh.v_max)
    return _dense_ranges(n_t, n_v, n_d, v_lo, v_hi, horizon, dt, d0, t_min, d_min, d_max)


def synthetic_predictions(cs: CoordinateSystem, n_obstacles: int, n_pred: int, dt: float, s_center: float,
                          rng: np.random.Generator, corridor=(120.0, 12.0), min_gap: float = 0.0, lead_gap: float = 0.0):
    """K obstacles, constant velocity along/against the reference tangent.
    min_gap > 0: an obstacle that would start within min_gap metres (arc length) of s_center is drawn again, so the
    ego does not start inside an obstacle.  lead_gap > 0: obstacle 0 is a slow lead vehicle (3 m/s) on the reference
    line lead_gap metres ahead, so that the cheapest candidates -- the ones that keep the lane at the desired speed --
    run into it (SURVEY 8d asks that a good share of the otherwise-best candidates collide)."""
    preds = {}
    for k in range(n_obstacles):
        while True:
            s0 = s_center + rng.uniform(-0.15, 0.85) * corridor[0]
            d0 = rng.uniform(-0.5, 0.5) * corridor[1]
            if abs(s0 - s_center) >= min_gap:
                break
        direction = 1.0 if rng.uniform() < 0.7 else -1.0
        speed = rng.uniform(3.0, 12.0)
        if k == 0 and lead_gap > 0:
            s0, d0, direction, speed = s_center + lead_gap, 0.0, 1.0, 3.0
        s0 = float(np.clip(s0, cs.ref_pos[2], cs.ref_pos[-3]))
        seg = cs.segment_of(s0)
        yaw_ref = float(cs.ref_theta[seg])
        yaw = yaw_ref if direction > 0 else yaw_ref + np.pi
        p0 = cs.convert_to_cartesian_coords(s0, d0)
        steps = np.arange(1, n_pred + 1) * dt
        pos = p0[None, :] + steps[:, None] * speed * np.array([np.cos(yaw), np.sin(yaw)])[None, :]
        grow = (1.02 ** np.arange(n_pred))[:, None, None]
        cov = np.tile(np.diag([0.1, 0.1])[None], (n_pred, 1, 1)) * grow
        preds[100 + k] = dict(pos_list=pos, cov_list=cov, orientation_list=np.full(n_pred, yaw),
                              v_list=np.full(n_pred, speed),
                              shape=dict(length=float(rng.uniform(4.5, 5.5)), width=float(rng.uniform(1.8, 2.2))))
    return preds


def offset_road_boundary(cs, half_width: float) -> np.ndarray:
    """Segments [n][4] of the two polylines reference +- half_width * vertex normal."""
    left = cs.reference + half_width * cs.normals
    right = cs.reference - half_width * cs.normals
    return np.concatenate([np.concatenate([left[:-1], left[1:]], axis=1), np.concatenate([right[:-1], right[1:]], axis=1)])


def make_inputs(*, ref_kind="arc", n_knots=400, spacing=0.5, kappa=0.01, v0=10.0, a0=0.0, d0=0.2, dd0=0.0, ddd0=0.0,
                s_knot=40, s_off=0.1, horizon=3.0, dt=0.1, level=None, grid=None, cpp_style=False, v_des=12.0,
                n_obstacles=0, n_pred=30, cost_weights=None, draw_traj_set=False, kinematic_debug=False,
                write_bundle=True, write_costmap=True, collision=True, low_vel_threshold=2.0, hull_builder=None,
                seed=SEED, vehicle=None, x0_orientation=None, as_matrix=False, stop_point_s=None, road_half_width=None,
                obstacle_min_gap=0.0, lead_gap=0.0, knot_jitter=0.0, pseudo_normal=False, vertex_tangent="chord", lanelets=None,
                heading0=0.0, wrap_x0_orientation=False):
    """One agent's PlanInputs on a synthetic reference.

    level: reference sampling level (set-ordered ranges, SamplingHandler) -- or
    grid=(n_t, n_v, n_d): dense grid (BASELINE configs 2/3/5).
    road_half_width: road boundary = the reference offset by +-road_half_width (two polylines of segments).
    stop_point_s: distance ahead of s0 of a stop point -> stop-point sampling (end positions in
    [(s0 + s_stop) / 2, s_stop], reactive_planner.py:637) instead of end velocities."""
    veh = vehicle or VehicleParams()
    cs = CoordinateSystem(reference_polyline(ref_kind, n_knots, spacing, kappa, knot_jitter, seed, heading0),
                          pseudo_normal=pseudo_normal, vertex_tangent=vertex_tangent)
    N = int(horizon / dt)
    s0 = float(cs.ref_pos[s_knot] + s_off)
    low_vel = v0 < low_vel_threshold
    if grid is not None:
        t, v, d = dense_ranges(grid[0], grid[1], grid[2], v0, veh, horizon, dt, d0)
    else:
        sh = SamplingHandler(dt=dt, max_sampling_number=max((level or 2) + 1, 3), t_min=1.1, horizon=horizon,
                             delta_d_min=-3.0, delta_d_max=3.0, d_ego_pos=False)
        sh.set_v_sampling(*v_sampling_bounds(v0, veh.a_max, horizon, veh.v_max))
        t, v, d = sh.ordered_ranges(level if level is not None else 2, d0, cpp_style=cpp_style, ss0=v0, t_full=N * dt)
    if stop_point_s is not None:
        s_stop = s0 + float(stop_point_s)
        if grid is not None:
            v = np.linspace((s0 + s_stop) / 2, s_stop, grid[1])
        else:
            sh.set_s_sampling((s0 + s_stop) / 2, s_stop)
            v = sh.s_sampling.ordered(level if level is not None else 2)
    seg = cs.segment_of(s0)
    if x0_orientation is None:
        x0_orientation = float(cs.ref_theta[seg])
        if wrap_x0_orientation:   # a vehicle state's heading lies in (-pi, pi]; the unwrapped reference heading need not
            x0_orientation = float(np.arctan2(np.sin(x0_orientation), np.cos(x0_orientation)))
    rng = np.random.default_rng(seed)
    preds = synthetic_predictions(cs, n_obstacles, n_pred, dt, s0, rng, min_gap=obstacle_min_gap,
                                  lead_gap=lead_gap) if n_obstacles else None
    weights = dict(cost_weights if cost_weights is not None else DEFAULT_COST_WEIGHTS)
    if not preds:
        weights.pop("prediction", None) if cost_weights is None else None
    obstacles = pack_predictions(preds, N + 1, hull_builder)
    kw = dict(N=N, dt=dt, low_vel_mode=low_vel, x0_lon=[s0, v0, a0], x0_lat=[d0, dd0, ddd0],
              x0_orientation=x0_orientation, v_des=v_des, vehicle=veh, coordinate_system=cs, cost_weights=weights,
              draw_traj_set=draw_traj_set, kinematic_debug=kinematic_debug, write_bundle=write_bundle,
              write_costmap=write_costmap, collision=collision, obstacles=obstacles)
    if road_half_width is not None:
        kw["road_boundary"] = offset_road_boundary(cs, float(road_half_width))
    if lanelets is not None:   # (lane width, knots per lanelet): two lanes along the reference (lane_center_offset cost)
        kw["lanelets"] = lanes_along(cs, *lanelets)
    if as_matrix:
        from .sampling import generate_sampling_matrix
        m = generate_sampling_matrix(t0_range=0.0, t1_range=t, s0_range=s0, ss0_range=v0, sss0_range=a0, ss1_range=v,
                                     sss1_range=0.0, d0_range=d0, dd0_range=dd0, ddd0_range=ddd0, d1_range=d,
                                     dd1_range=0.0, ddd1_range=0.0)
        inp = PlanInputs(sampling_matrix=m, **kw)
    else:
        inp = PlanInputs(t_samp=t, v_samp=v, d_samp=d, stop_point=stop_point_s is not None, **kw)
    inp.predictions = preds
    return inp


def lanes_along(cs: CoordinateSystem, width: float = 3.5, knots_per_lanelet: int = 60):
    """Lanelets for the lane_center_offset cost: the reference's own lane (|d| <= width / 2) and its left neighbour, cut into
    pieces of `knots_per_lanelet` reference knots (consecutive pieces share their end vertices), in network order own lane
    first.  Right of the own lane there is no lanelet (the cost's 5 m branch)."""
    from types import SimpleNamespace
    ref, nrm = np.asarray(cs.reference), np.asarray(cs.normals)
    nrm = nrm / np.linalg.norm(nrm, axis=1)[:, None]
    out = []
    for lo, hi in ((-0.5 * width, 0.5 * width), (0.5 * width, 1.5 * width)):
        for k0 in range(0, len(ref) - 1, knots_per_lanelet):
            sl = slice(k0, min(k0 + knots_per_lanelet + 1, len(ref)))
            out.append(SimpleNamespace(left_vertices=ref[sl] + hi * nrm[sl], right_vertices=ref[sl] + lo * nrm[sl]))
    return out


def stress_agents(n_agents: int, grid=(39, 51, 51), horizon: float = 5.0, n_obstacles: int = 20, first_agent: int = 0,
                  hull_builder=None, write_bundle: bool = False, seed: int = SEED, **kw):
    """BASELINE config 5 ("synthetic stress"): agents that differ by seeded v0 in [2, 20] m/s, d0 in [-1, 1] m and a
    reference curvature in [-0.02, 0.02] 1/m, each with its own 20 predicted obstacles; horizon 5 s (N = 50),
    T = 1.1 .. 4.9 step 0.1 (39) x 51 end velocities x 51 (+ d0) lateral offsets = 103 428 candidates with d0 added
    (101 439 when d0 is one of the 51 samples).  Agent a's draw depends only on (seed, a), so any rank can build any
    agent: `first_agent` selects the slice [first_agent, first_agent + n_agents) of the global agent list."""
    agents = []
    for a in range(first_agent, first_agent + n_agents):
        rng = np.random.default_rng([seed, a])
        v0 = float(rng.uniform(2.0, 20.0))
        d0 = float(np.round(rng.uniform(-1.0, 1.0), 3))
        kappa = float(rng.uniform(-0.02, 0.02))
        if abs(kappa) < 1e-4:
            kappa = 1e-4
        args = dict(ref_kind="arc", n_knots=500, spacing=0.5, kappa=kappa, v0=v0, d0=d0, horizon=horizon, grid=grid,
                    v_des=float(np.clip(v0 + rng.uniform(-2.0, 4.0), 1.0, 25.0)), n_obstacles=n_obstacles,
                    n_pred=int(round(horizon / 0.1)), write_bundle=write_bundle, write_costmap=write_bundle,
                    seed=int(rng.integers(1 << 30)), hull_builder=hull_builder if n_obstacles else None, obstacle_min_gap=12.0)
        args.update(kw)
        agents.append(make_inputs(**args))
    return agents
