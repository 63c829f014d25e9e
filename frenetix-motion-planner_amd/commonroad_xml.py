"""CommonRoad (2020a) scenario reader on the standard library's XML parser.

The reference reads scenarios with commonroad-io (`CommonRoadFileReader(path).open()`,
cr_scenario_handler/utils/general.py) and turns them into planner inputs with commonroad-route-planner and its own
helpers.  This module covers the part of that the hot path needs, with no third-party dependency:

    read_scenario(path)                      -> Scenario (lanelets, obstacles, planning problems, dt)
    Scenario.route_reference_path(pp)        -> centre-line polyline from the initial lanelet to the goal lanelet
                                                (successor-graph BFS; stands in for commonroad-route-planner, which is
                                                not in the reference tree)
    Scenario.ground_truth_predictions(...)   -> the predictions dict of prediction_helpers.get_ground_truth_prediction
                                                (:209-261): pos_list / cov_list / orientation_list / v_list / shape
    PlanningProblem.initial_planner_state()  -> ReactivePlannerState for ReactivePlannerHip.update_externals

Only what the example scenarios (ZAM_Tjunction-*) use is interpreted: rectangle / circle obstacle shapes, `exact`
state values (interval values take their midpoint), lanelet bounds and topology, goal position by lanelet reference
or rectangle.  Traffic signs, lights and intersections are skipped.
"""
import xml.etree.ElementTree as ET
from collections import deque
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


try:   # the CPython helper built next to the package (csrc/fx_host_ext.c); the NumPy expression below is the same test
    from ._fxhost import point_in_polygon as _PIP
except ImportError:
    _PIP = None

@dataclass
class State:
    time_step: int
    position: np.ndarray
    orientation: float = 0.0
    velocity: float = 0.0
    acceleration: float = 0.0
    yaw_rate: float = 0.0
    slip_angle: float = 0.0


@dataclass
class Lanelet:
    lanelet_id: int
    left_vertices: np.ndarray
    right_vertices: np.ndarray
    predecessor: List[int] = field(default_factory=list)
    successor: List[int] = field(default_factory=list)
    adj_left: Optional[int] = None
    adj_left_same_direction: Optional[bool] = None
    adj_right: Optional[int] = None
    adj_right_same_direction: Optional[bool] = None
    lanelet_type: List[str] = field(default_factory=list)

    @property
    def center_vertices(self) -> np.ndarray:
        return 0.5 * (self.left_vertices + self.right_vertices)

    def contains(self, p) -> bool:
        """Point-in-polygon (ray casting) on the lanelet's outline; the outline (and its bounding box, which settles most
        queries) is built once per lanelet."""
        c = self.__dict__.get("_outline")
        if c is None:
            poly = np.vstack([self.left_vertices, self.right_vertices[::-1]])
            xi, yi = poly[:, 0], poly[:, 1]
            xi, yi = np.ascontiguousarray(xi), np.ascontiguousarray(yi)
            c = self.__dict__["_outline"] = (xi, yi, np.roll(xi, 1), np.roll(yi, 1), float(xi.min()), float(xi.max()),
                                             float(yi.min()), float(yi.max()))
        xi, yi, xj, yj, x0, x1, y0, y1 = c
        x, y = float(p[0]), float(p[1])
        if x < x0 or x > x1 or y < y0 or y > y1:
            return False
        if _PIP is not None:   # the same expression, term by term, in C (_fxhost.point_in_polygon): no temporaries
            return _PIP(xi, yi, x, y)
        cross = (yi > y) != (yj > y)
        if not cross.any():
            return False
        with np.errstate(divide="ignore", invalid="ignore"):
            hit = cross & (x < (xj - xi) * (y - yi) / (yj - yi) + xi)
        return bool(np.count_nonzero(hit) & 1)


@dataclass
class Obstacle:
    obstacle_id: int
    role: str                     # "dynamic" | "static"
    obstacle_type: str
    shape: dict                   # {"length", "width"} (rectangle) or {"radius"} (circle)
    initial_state: State
    state_list: List[State] = field(default_factory=list)  # trajectory states, first one at initial time + 1

    @property
    def length(self) -> float:
        return float(self.shape.get("length", 2 * self.shape.get("radius", 0.0)))

    @property
    def width(self) -> float:
        return float(self.shape.get("width", 2 * self.shape.get("radius", 0.0)))

    def state_at_time(self, time_step: int) -> Optional[State]:
        t0 = self.initial_state.time_step
        if time_step == t0:
            return self.initial_state
        if self.role != "dynamic":
            return self.initial_state if time_step >= t0 else None
        k = time_step - t0 - 1
        return self.state_list[k] if 0 <= k < len(self.state_list) else None


@dataclass
class GoalState:
    time_interval: Optional[tuple] = None
    velocity_interval: Optional[tuple] = None
    orientation_interval: Optional[tuple] = None
    lanelet_ids: List[int] = field(default_factory=list)
    rectangles: List[dict] = field(default_factory=list)  # {"length", "width", "orientation", "center"}


@dataclass
class PlanningProblem:
    planning_problem_id: int
    initial_state: State
    goals: List[GoalState]

    def initial_planner_state(self):
        from .reactive_planner import ReactivePlannerState
        s = self.initial_state
        return ReactivePlannerState(time_step=s.time_step, position=np.array(s.position, dtype=np.float64),
                                    orientation=s.orientation, velocity=s.velocity, acceleration=s.acceleration,
                                    yaw_rate=s.yaw_rate)


@dataclass
class Scenario:
    benchmark_id: str
    dt: float
    lanelets: Dict[int, Lanelet]
    obstacles: Dict[int, Obstacle]
    planning_problems: Dict[int, PlanningProblem]

    # -- lanelet network ------------------------------------------------------------------------------------------
    def lanelets_at(self, position) -> List[int]:
        return [i for i, l in self.lanelets.items() if l.contains(position)]

    def route(self, start_ids, goal_ids) -> Optional[List[int]]:
        """Shortest lanelet sequence (fewest lanelets) along successor edges, lane changes to same-direction
        neighbours allowed."""
        goal = set(goal_ids)
        prev = {s: None for s in start_ids}
        q = deque(start_ids)
        while q:
            cur = q.popleft()
            if cur in goal:
                seq = []
                while cur is not None:
                    seq.append(cur)
                    cur = prev[cur]
                return seq[::-1]
            ll = self.lanelets[cur]
            nxt = list(ll.successor)
            if ll.adj_left is not None and ll.adj_left_same_direction:
                nxt.append(ll.adj_left)
            if ll.adj_right is not None and ll.adj_right_same_direction:
                nxt.append(ll.adj_right)
            for n in nxt:
                if n in self.lanelets and n not in prev:
                    prev[n] = cur
                    q.append(n)
        return None

    def route_reference_path(self, planning_problem: PlanningProblem) -> np.ndarray:
        """Concatenated centre lines of the route from the initial position to the first reachable goal lanelet."""
        start = self.lanelets_at(planning_problem.initial_state.position)
        if not start:
            d = {i: np.min(np.linalg.norm(l.center_vertices - planning_problem.initial_state.position, axis=1))
                 for i, l in self.lanelets.items()}
            start = [min(d, key=d.get)]
        goal_ids = []
        for g in planning_problem.goals:
            goal_ids += g.lanelet_ids
            for r in g.rectangles:
                goal_ids += self.lanelets_at(r["center"])
        seq = self.route(start, goal_ids) if goal_ids else None
        if seq is None:  # no goal lanelet reachable: follow successors as far as they go
            seq = [start[0]]
            while self.lanelets[seq[-1]].successor and self.lanelets[seq[-1]].successor[0] not in seq:
                seq.append(self.lanelets[seq[-1]].successor[0])
        pts = [self.lanelets[seq[0]].center_vertices]
        for prev_id, cur_id in zip(seq[:-1], seq[1:]):
            c = self.lanelets[cur_id].center_vertices
            if cur_id in self.lanelets[prev_id].successor:
                if np.allclose(c[0], pts[-1][-1], atol=1e-3):
                    c = c[1:]  # shared vertex of consecutive lanelets
            else:
                pts.pop()      # lane change: continue on the neighbour's centre line instead
            pts.append(c)
        return np.vstack(pts)

    def road_boundary_segments(self, tol: float = 0.05) -> np.ndarray:
        """Outer border of the drivable area as segments [n][4] = (ax, ay, bx, by): every lanelet-bound segment that
        is neither shared with a neighbouring lanelet nor inside another lanelet (turning lanelets of an intersection
        overlap each other).  Stands in for commonroad_dc's `create_road_boundary_obstacle(scenario,
        method="aligned_triangulation")` (planner.py:550-565), whose triangles fill the outside of the same border."""
        segs = []
        ll = list(self.lanelets.values())
        for l in ll:
            for side, bound in (("left", l.left_vertices), ("right", l.right_vertices)):
                nb = l.adj_left if side == "left" else l.adj_right
                if nb is not None and nb in self.lanelets:
                    continue  # shared with the neighbouring lane
                inward = (l.right_vertices - l.left_vertices) if side == "left" else (l.left_vertices - l.right_vertices)
                for i in range(len(bound) - 1):
                    a, b = bound[i], bound[i + 1]
                    if np.allclose(a, b):
                        continue
                    mid = 0.5 * (a + b)
                    n = 0.5 * (inward[i] + inward[i + 1])
                    n = n / max(np.linalg.norm(n), 1e-12)
                    probe = mid - tol * n  # just outside this lanelet
                    if any(o is not l and o.contains(probe) for o in ll):
                        continue           # drivable on the other side: not a border
                    segs.append([a[0], a[1], b[0], b[1]])
        return np.array(segs, dtype=np.float64).reshape(-1, 4)

    # -- predictions ----------------------------------------------------------------------------------------------
    def ground_truth_predictions(self, time_step: int, pred_horizon: int = 50, obstacle_ids=None) -> dict:
        """prediction_helpers.get_ground_truth_prediction (:209-261): the obstacles' recorded futures as a prediction.
        Per obstacle and step ts in [time_step, min(pred_horizon + time_step, len(occupancy_set))): position and
        orientation of the occupancy at ts, covariance 0.1 I, velocity of trajectory.state_list[ts] (the reference
        indexes the state list -- which starts one step after the initial state -- with the absolute step)."""
        out = {}
        for oid in (obstacle_ids if obstacle_ids is not None else list(self.obstacles)):
            ob = self.obstacles[oid]
            len_pred = len(ob.state_list) if ob.role == "dynamic" else pred_horizon
            t1 = min(pred_horizon + time_step, len_pred)
            tab = self._gt_table(oid, max(t1, 0))
            if tab is not None:
                # per-step tables of the obstacle built once (every step exists): a closed loop asks for a window per step
                pos_all, yaw_all, vel_all, cov_all = tab
                sl = slice(min(time_step, t1), t1)
                out[oid] = dict(pos_list=pos_all[sl], cov_list=cov_all[sl], orientation_list=yaw_all[sl], v_list=vel_all[sl],
                                shape=dict(length=ob.length, width=ob.width))
                continue
            pos, cov, yaw, vel = [], [], [], []
            for ts in range(time_step, t1):
                st = ob.state_at_time(ts)
                if st is None:
                    continue
                pos.append(st.position)
                cov.append([[0.1, 0.0], [0.0, 0.1]])
                yaw.append(st.orientation)
                if ob.role == "dynamic":
                    vel.append(ob.state_list[ts].velocity)  # state_list[ts] is the state of step ts + 1 (:248)
                else:
                    vel.append(ob.initial_state.velocity)
            out[oid] = dict(pos_list=np.array(pos, dtype=np.float64).reshape(-1, 2), cov_list=np.array(cov, dtype=np.float64),
                            orientation_list=np.array(yaw, dtype=np.float64), v_list=np.array(vel, dtype=np.float64),
                            shape=dict(length=ob.length, width=ob.width))
        return out

    def _gt_table(self, oid: int, upto: int):
        """(positions [T, 2], orientations [T], velocities [T], covariances [T, 2, 2]) of obstacle `oid` for the steps 0 .. T-1,
        T >= upto -- or None when some step in that range has no state (then the caller walks the steps one by one).  Read-only
        arrays, cached on the scenario: the windows handed out are views."""
        cache = self.__dict__.setdefault("_gt_tables", {})
        tab = cache.get(oid)
        if tab is not None and (tab is False or len(tab[1]) >= upto):
            return tab or None
        ob = self.obstacles[oid]
        # a closed loop asks for a window that ends a few steps later every time: the table is built once over everything the
        # obstacle has (a static obstacle's prediction has no end: in blocks of 256 steps)
        upto = max(upto, len(ob.state_list)) if ob.role == "dynamic" else (upto + 255) // 256 * 256
        pos, yaw, vel = [], [], []
        for ts in range(upto):
            st = ob.state_at_time(ts)
            if st is None:
                cache[oid] = False
                return None
            pos.append(st.position)
            yaw.append(st.orientation)
            vel.append(ob.state_list[ts].velocity if ob.role == "dynamic" else ob.initial_state.velocity)
        tab = (np.array(pos, dtype=np.float64).reshape(-1, 2), np.array(yaw, dtype=np.float64), np.array(vel, dtype=np.float64),
               np.tile(np.array([[0.1, 0.0], [0.0, 0.1]]), (upto, 1, 1)))
        for a in tab:
            a.setflags(write=False)
        cache[oid] = tab
        return tab


# ----------------------------------------------------------------------------------------------------------------
def _num(node, default=0.0) -> float:
    """<exact>v</exact> or the midpoint of <intervalStart>/<intervalEnd>."""
    if node is None:
        return default
    e = node.find("exact")
    if e is not None:
        return float(e.text)
    a, b = node.find("intervalStart"), node.find("intervalEnd")
    if a is not None and b is not None:
        return 0.5 * (float(a.text) + float(b.text))
    return default


def _interval(node):
    if node is None:
        return None
    e = node.find("exact")
    if e is not None:
        return (float(e.text), float(e.text))
    a, b = node.find("intervalStart"), node.find("intervalEnd")
    return (float(a.text), float(b.text)) if a is not None and b is not None else None


def _point(node) -> np.ndarray:
    return np.array([float(node.find("x").text), float(node.find("y").text)], dtype=np.float64)


def _position(node) -> np.ndarray:
    if node is None:
        return np.zeros(2)
    p = node.find("point")
    if p is not None:
        return _point(p)
    for tag in ("rectangle", "circle"):
        s = node.find(tag)
        if s is not None and s.find("center") is not None:
            return _point(s.find("center"))
    return np.zeros(2)


def _state(node) -> State:
    return State(time_step=int(round(_num(node.find("time")))), position=_position(node.find("position")),
                 orientation=_num(node.find("orientation")), velocity=_num(node.find("velocity")),
                 acceleration=_num(node.find("acceleration")), yaw_rate=_num(node.find("yawRate")),
                 slip_angle=_num(node.find("slipAngle")))


def _bound(node) -> np.ndarray:
    return np.array([_point(p) for p in node.findall("point")], dtype=np.float64)


def _shape(node) -> dict:
    if node is None:
        return {}
    r = node.find("rectangle")
    if r is not None:
        return dict(length=float(r.find("length").text), width=float(r.find("width").text))
    c = node.find("circle")
    if c is not None:
        return dict(radius=float(c.find("radius").text))
    return {}


def _adjacent(node):
    if node is None:
        return None, None
    return int(node.get("ref")), node.get("drivingDir", "same") == "same"


def read_scenario(path: str) -> Scenario:
    root = ET.parse(path).getroot()
    if root.tag != "commonRoad":
        raise ValueError(f"{path}: not a CommonRoad scenario (root element <{root.tag}>)")
    lanelets = {}
    for n in root.findall("lanelet"):
        lid = int(n.get("id"))
        al, al_same = _adjacent(n.find("adjacentLeft"))
        ar, ar_same = _adjacent(n.find("adjacentRight"))
        lanelets[lid] = Lanelet(lid, _bound(n.find("leftBound")), _bound(n.find("rightBound")),
                                [int(p.get("ref")) for p in n.findall("predecessor")],
                                [int(p.get("ref")) for p in n.findall("successor")], al, al_same, ar, ar_same,
                                [t.text for t in n.findall("laneletType")])
    obstacles = {}
    for tag, role in (("dynamicObstacle", "dynamic"), ("staticObstacle", "static")):
        for n in root.findall(tag):
            oid = int(n.get("id"))
            traj = n.find("trajectory")
            states = [_state(s) for s in traj.findall("state")] if traj is not None else []
            typ = n.find("type")
            obstacles[oid] = Obstacle(oid, role, typ.text if typ is not None else "unknown", _shape(n.find("shape")),
                                      _state(n.find("initialState")), states)
    problems = {}
    for n in root.findall("planningProblem"):
        pid = int(n.get("id"))
        goals = []
        for g in n.findall("goalState"):
            gs = GoalState(time_interval=_interval(g.find("time")), velocity_interval=_interval(g.find("velocity")),
                           orientation_interval=_interval(g.find("orientation")))
            pos = g.find("position")
            if pos is not None:
                gs.lanelet_ids = [int(l.get("ref")) for l in pos.findall("lanelet")]
                for r in pos.findall("rectangle"):
                    o = r.find("orientation")
                    gs.rectangles.append(dict(length=float(r.find("length").text), width=float(r.find("width").text),
                                              orientation=float(o.text) if o is not None else 0.0,
                                              center=_point(r.find("center")) if r.find("center") is not None else np.zeros(2)))
            goals.append(gs)
        problems[pid] = PlanningProblem(pid, _state(n.find("initialState")), goals)
    return Scenario(root.get("benchmarkID", ""), float(root.get("timeStepSize", "0.1")), lanelets, obstacles, problems)


# ----------------------------------------------------------------------------------------------------------------
# Compact JSON form of a Scenario (what this reader interprets, nothing else) -- the fixture format of
# tests/golden/*.scenario.json: the XML is not on the GPU box, the numbers are.
def _state_row(s: State) -> list:
    return [s.time_step, float(s.position[0]), float(s.position[1]), s.orientation, s.velocity, s.acceleration, s.yaw_rate,
            s.slip_angle]


def _row_state(r) -> State:
    return State(int(r[0]), np.array([r[1], r[2]], dtype=np.float64), float(r[3]), float(r[4]), float(r[5]), float(r[6]),
                 float(r[7]))


def scenario_to_dict(sc: Scenario) -> dict:
    return dict(
        benchmark_id=sc.benchmark_id, dt=sc.dt,
        lanelets=[dict(id=l.lanelet_id, left=l.left_vertices.tolist(), right=l.right_vertices.tolist(),
                       predecessor=l.predecessor, successor=l.successor, adj_left=l.adj_left,
                       adj_left_same=l.adj_left_same_direction, adj_right=l.adj_right,
                       adj_right_same=l.adj_right_same_direction, type=l.lanelet_type) for l in sc.lanelets.values()],
        obstacles=[dict(id=o.obstacle_id, role=o.role, type=o.obstacle_type, shape=o.shape,
                        initial=_state_row(o.initial_state), states=[_state_row(s) for s in o.state_list])
                   for o in sc.obstacles.values()],
        planning_problems=[dict(id=p.planning_problem_id, initial=_state_row(p.initial_state),
                                goals=[dict(time=g.time_interval, velocity=g.velocity_interval,
                                            orientation=g.orientation_interval, lanelets=g.lanelet_ids,
                                            rectangles=[dict(r, center=[float(r["center"][0]), float(r["center"][1])])
                                                        for r in g.rectangles]) for g in p.goals])
                           for p in sc.planning_problems.values()])


def scenario_from_dict(d: dict) -> Scenario:
    lanelets = {l["id"]: Lanelet(l["id"], np.array(l["left"], dtype=np.float64), np.array(l["right"], dtype=np.float64),
                                 list(l["predecessor"]), list(l["successor"]), l["adj_left"], l["adj_left_same"],
                                 l["adj_right"], l["adj_right_same"], list(l["type"])) for l in d["lanelets"]}
    obstacles = {o["id"]: Obstacle(o["id"], o["role"], o["type"], dict(o["shape"]), _row_state(o["initial"]),
                                   [_row_state(r) for r in o["states"]]) for o in d["obstacles"]}
    problems = {}
    for p in d["planning_problems"]:
        goals = [GoalState(time_interval=tuple(g["time"]) if g["time"] else None,
                           velocity_interval=tuple(g["velocity"]) if g["velocity"] else None,
                           orientation_interval=tuple(g["orientation"]) if g["orientation"] else None,
                           lanelet_ids=list(g["lanelets"]),
                           rectangles=[dict(r, center=np.array(r["center"], dtype=np.float64)) for r in g["rectangles"]])
                 for g in p["goals"]]
        problems[p["id"]] = PlanningProblem(p["id"], _row_state(p["initial"]), goals)
    return Scenario(d["benchmark_id"], float(d["dt"]), lanelets, obstacles, problems)


def write_scenario_json(sc: Scenario, path: str):
    import json
    with open(path, "w") as fh:
        json.dump(scenario_to_dict(sc), fh, separators=(",", ":"))


def read_scenario_json(path: str) -> Scenario:
    import json
    with open(path) as fh:
        return scenario_from_dict(json.load(fh))
