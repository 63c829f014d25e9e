"""Route polyline -> planner reference path: the input preparation in front of the hot path.

Behaviour contract (checked against vectors the reference's own functions produced, tests/golden/gen_refpath_golden.py):

    cr_scenario_handler/utils/utils_coordinate_system.py
        extend_path_linearly :21-51     extend_ref_path_both_ends :54-58     extend_points :61-77
        extend_points_end    :80-99     extend_ref_path           :102-108   smooth_ref_path :110-134
        extrapolate_ref_path :158-170   preprocess_ref_path       :173-184

`FrenetPlannerInterface` feeds `smooth_ref_path(extend_ref_path_both_ends(route.reference_path))` to the planner's
coordinate system (`frenet_interface.py:110-114`); `prepare_reference_path` is that composition.

Built from three primitives of this module -- `prolong` (march beyond one end of a polyline), `resample_polyline` (equal
arc-length steps) and `_bspline_through` (interpolating cubic B-spline, SciPy) -- instead of one routine per call site.
`resample_polyline` stands in for `commonroad_dc.geometry.util.resample_polyline` (commonroad-drivability-checker 2024.1, not
in the reference tree): restated from its published behaviour, parity unpinned.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------------------------------
def _as_polyline(points) -> np.ndarray:
    return np.asarray(points, dtype=np.float64).reshape(-1, 2)


def prolong(points, count: int, stride, front: bool) -> np.ndarray:
    """`count` extra vertices beyond the first (`front`) or the last vertex: end -/+ i * stride, i = 1..count, kept in path
    order.  `stride` is a 2-vector or a (scale, 2-vector) pair -- the pair evaluates (i * vector) * scale, which is the order
    in which the reference forms its unit-step products."""
    pts = _as_polyline(points)
    if count <= 0:
        return pts
    scale, vec = stride if isinstance(stride, tuple) else (None, stride)
    k = np.arange(1, count + 1, dtype=np.float64)[:, None] * np.asarray(vec, dtype=np.float64)[None, :]
    if scale is not None:
        k = k * scale
    if front:
        return np.concatenate([(pts[0] - k)[::-1], pts])
    return np.concatenate([pts, pts[-1] + k])


def resample_polyline(polyline, step: float = 2.0) -> np.ndarray:
    """Vertices every `step` metres of arc length, starting on the first vertex; the last vertex closes the result unless the
    final sample already lies on it (within 1e-6).  A sample that falls exactly on a vertex belongs to the segment that
    starts there."""
    pts = _as_polyline(polyline)
    if len(pts) < 2:
        return pts.copy()
    seg = np.linalg.norm(np.diff(pts, axis=0), axis=1)
    # arc length at which segment j starts, accumulated the way a walk along the path accumulates it
    start = np.concatenate([[0.0], np.cumsum(seg)])
    total = start[-1]
    n = int(np.floor(total / step)) + 1
    targets = float(step) * np.arange(1, n + 1, dtype=np.float64)
    targets = targets[targets < total]
    j = np.searchsorted(start, targets, side="right") - 1
    j = np.clip(j, 0, len(seg) - 1)
    # skip zero-length segments (duplicate vertices) a target may land on
    while True:
        bad = (seg[j] == 0.0) & (j < len(seg) - 1)
        if not bad.any():
            break
        j = np.where(bad, j + 1, j)
    rel = np.divide(targets - start[j], seg[j], out=np.zeros_like(targets), where=seg[j] > 0)
    out = np.concatenate([pts[:1], (1.0 - rel)[:, None] * pts[j] + rel[:, None] * pts[j + 1]])
    if np.linalg.norm(out[-1] - pts[-1]) >= 1e-6:
        out = np.concatenate([out, pts[-1:]])
    return out


def _first_occurrences(points: np.ndarray) -> np.ndarray:
    """Drop repeated vertices, keeping the first of each and the path order."""
    _, first = np.unique(points, axis=0, return_index=True)
    return points[np.sort(first)]


def _bspline_through(points: np.ndarray, n_samples: int) -> np.ndarray:
    """Interpolating cubic B-spline through `points` (chord-length parameter, no smoothing), evaluated at `n_samples`
    equally spaced parameter values."""
    from scipy.interpolate import splev, splprep
    tck, u = splprep(points.T, u=None, k=3, s=0.0)
    x, y = splev(np.linspace(u.min(), u.max(), n_samples), tck, der=0)
    return np.stack([x, y], axis=1)


# ---------------------------------------------------------------------------------------------------------------------
# the reference's entry points
# ---------------------------------------------------------------------------------------------------------------------
def extend_path_linearly(points, extension_length=50, at_start=True):
    """Straight continuation of the first / last segment by `extension_length` metres, one vertex per segment length
    (utils_coordinate_system.py:21-51).  Coincident end vertices: the input is handed back untouched."""
    a, b = (points[0], points[1]) if at_start else (points[-2], points[-1])
    heading = np.array([b[0] - a[0], b[1] - a[1]], dtype=np.float64)
    spacing = float(np.sqrt(heading[0] ** 2 + heading[1] ** 2))
    if spacing == 0:
        return points
    return prolong(points, int(extension_length / spacing), (spacing, heading / spacing), front=at_start)


def extend_ref_path_both_ends(ref_path, extension_length=30):
    """Both ends continued by `extension_length` metres (:54-58)."""
    return extend_path_linearly(extend_path_linearly(ref_path, extension_length, at_start=True), extension_length,
                                at_start=False)


def extend_points(points):
    """About five metres of extra vertices in front of the path, spaced like its first segment (:61-77)."""
    first = np.array([points[1][0] - points[0][0], points[1][1] - points[0][1]], dtype=np.float64)
    spacing = float(np.sqrt(first[0] ** 2 + first[1] ** 2))
    return prolong(points, int(5 / spacing), first, front=True)


def extend_ref_path(ref_path, init_pos):
    """The rear-axle start position may project in front of the path: when the first vertex is the closest one, vertices are
    added in front (:102-108)."""
    ref_path = _as_polyline(ref_path)
    gap2 = (ref_path[:, 0] - init_pos[0]) ** 2 + (ref_path[:, 1] - init_pos[1]) ** 2
    nearest = ref_path[int(np.argmin(gap2))]
    return extend_points(ref_path) if (nearest == ref_path[0]).all() else ref_path


def extend_points_end(points, extension_length=30):
    """`extension_length` metres of extra vertices behind the path, spaced like its last segment (:80-99).  Coincident end
    vertices: the input is handed back untouched."""
    last = np.array([points[-1][0] - points[-2][0], points[-1][1] - points[-2][1]], dtype=np.float64)
    spacing = float(np.sqrt(last[0] ** 2 + last[1] ** 2))
    if spacing == 0:
        return points
    return prolong(points, int(extension_length / spacing), last, front=False)


def extrapolate_ref_path(reference_path: np.ndarray, resample_step: float = 0.25) -> np.ndarray:
    """One vertex on the straight line through the last two vertices, 1.3 x their x-distance further on, then resampled
    (:158-170: a reference shorter than the planning horizon would leave the projection domain)."""
    ref = _as_polyline(reference_path)
    line = np.poly1d(np.polyfit(ref[-2:, 0], ref[-2:, 1], 1))
    x = 2.3 * ref[-1, 0] - ref[-2, 0]
    return resample_polyline(np.concatenate([ref, np.array([[x, line(x)]])]), step=resample_step)


def chaikins_corner_cutting(polyline, refinements: int = 1) -> np.ndarray:
    """Chaikin's corner cutting: every refinement replaces each segment (p, q) by the points 3/4 p + 1/4 q and 1/4 p + 3/4 q
    and keeps the two end vertices.  Stands in for `commonroad_dc.geometry.util.chaikins_corner_cutting` (not in the reference
    tree): restated from the published algorithm, parity unpinned."""
    pts = _as_polyline(polyline)
    for _ in range(int(refinements)):
        if len(pts) < 2:
            break
        p, q = pts[:-1], pts[1:]
        cut = np.empty((2 * len(p), 2))
        cut[0::2] = 0.75 * p + 0.25 * q
        cut[1::2] = 0.25 * p + 0.75 * q
        pts = np.concatenate([pts[:1], cut, pts[-1:]])
    return pts


def preprocess_ref_path(ref_path: np.ndarray, resample_step: float = 0.1, max_curv_desired: float = 0.1):
    """Corner cutting + resampling until the largest curvature of the path is at most `max_curv_desired` (:173-184)."""
    from .coordinate_system import compute_curvature_from_polyline
    out = _as_polyline(ref_path).copy()
    max_curv = max_curv_desired + 0.2
    while max_curv > max_curv_desired:
        out = resample_polyline(chaikins_corner_cutting(out), resample_step)
        max_curv = max(compute_curvature_from_polyline(out))
    return out


ROUTE_SPACING = 0.125  # vertex spacing the route planner delivers; smooth_ref_path is written for it (:117)


def smooth_ref_path(reference: np.ndarray, smoothing_interval: float = 4):
    """Cubic B-spline through one route vertex per `smoothing_interval` metres, sampled six times per metre and brought to
    1 m spacing (:110-134).  The path length that sizes the sampling is taken from every second segment, as upstream."""
    route = _first_occurrences(_as_polyline(reference))
    every_other = route[0:-2:2] - route[1:-1:2]
    length = np.round(np.sum(np.sqrt(np.sum(every_other ** 2, axis=1))), 3)
    knots = route[::int(smoothing_interval / ROUTE_SPACING)]
    dense = _bspline_through(knots, int(6 * length))
    return _first_occurrences(resample_polyline(dense, 1))


def prepare_reference_path(route_reference_path: np.ndarray) -> np.ndarray:
    """Route planner polyline -> planner reference path (frenet_interface.py:110-114)."""
    return smooth_ref_path(extend_ref_path_both_ends(_as_polyline(route_reference_path)))
