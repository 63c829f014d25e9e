"""Reference-path preparation -- the step immediately in front of the hot path.

Host-side NumPy/SciPy, same names, arguments and behaviour as the reference's
`cr_scenario_handler/utils/utils_coordinate_system.py`:

    extend_path_linearly        :21-51      extend_ref_path_both_ends   :54-58
    extend_points               :61-77      extend_points_end           :80-99
    extend_ref_path             :102-108    smooth_ref_path             :110-134
    extrapolate_ref_path        :158-169    preprocess_ref_path         :172-184

`FrenetPlannerInterface` runs `smooth_ref_path(extend_ref_path_both_ends(route.reference_path))` before it builds
the planner's coordinate system (`frenet_interface.py:110-114`); `prepare_reference_path` is that composition.

Two helpers come from commonroad-drivability-checker (~2024.1, `commonroad_dc.geometry.util`, not in the reference
tree): `resample_polyline` and `chaikins_corner_cutting`.  They are restated here from the published algorithm --
parity for them is unpinned; everything else is checked against the reference's own functions
(tests/golden/gen_refpath_golden.py).
"""
from copy import deepcopy

import numpy as np

from .coordinate_system import compute_curvature_from_polyline


def _distance(p1, p2) -> float:
    """helper_functions.distance: Euclidean distance of two points."""
    return float(np.sqrt((p1[0] - p2[0]) ** 2 + (p1[1] - p2[1]) ** 2))


# ----------------------------------------------------------------------------------------------------------------
# commonroad_dc.geometry.util (third-party, restated)
# ----------------------------------------------------------------------------------------------------------------
def resample_polyline(polyline: np.ndarray, step: float = 2.0) -> np.ndarray:
    """Walks along the polyline and emits a vertex every `step` metres of arc length; the last vertex is appended
    when the walk does not end on it (closer than 1e-6)."""
    polyline = np.asarray(polyline, dtype=np.float64)
    if len(polyline) < 2:
        return np.array(polyline)
    new_polyline = [polyline[0]]
    current_position = step
    current_length = np.linalg.norm(polyline[0] - polyline[1])
    current_idx = 0
    while current_idx < len(polyline) - 1:
        if current_position >= current_length:
            current_position = current_position - current_length
            current_idx += 1
            if current_idx > len(polyline) - 2:
                break
            current_length = np.linalg.norm(polyline[current_idx + 1] - polyline[current_idx])
        else:
            rel = current_position / current_length
            new_polyline.append((1 - rel) * polyline[current_idx] + rel * polyline[current_idx + 1])
            current_position += step
    if np.linalg.norm(new_polyline[-1] - polyline[-1]) >= 1e-6:
        new_polyline.append(polyline[-1])
    return np.array(new_polyline)


def chaikins_corner_cutting(polyline: np.ndarray, refinements: int = 1) -> np.ndarray:
    """Chaikin's corner cutting: every segment is replaced by its 1/4 and 3/4 points, end points are kept."""
    polyline = np.asarray(polyline, dtype=np.float64)
    for _ in range(refinements):
        L = polyline.repeat(2, axis=0)
        R = np.empty_like(L)
        R[0] = L[0]
        R[2::2] = L[1:-1:2]
        R[1:-1:2] = L[2::2]
        R[-1] = L[-1]
        polyline = L * 0.75 + R * 0.25
    return polyline


# ----------------------------------------------------------------------------------------------------------------
# utils_coordinate_system.py
# ----------------------------------------------------------------------------------------------------------------
def extend_path_linearly(points, extension_length=50, at_start=True):
    """Extend the list of points linearly at the start or end by a given length (:21-51)."""
    if at_start:
        p1, p2 = points[0], points[1]
    else:
        p1, p2 = points[-2], points[-1]
    delta_x = p2[0] - p1[0]
    delta_y = p2[1] - p1[1]
    dist = np.sqrt(delta_x ** 2 + delta_y ** 2)
    if dist == 0:
        return points  # p1 and p2 coincide
    step_x = delta_x / dist
    step_y = delta_y / dist
    num_new_points = int(extension_length / dist)
    if num_new_points == 0:
        return np.asarray(points)  # np.vstack of an empty list raises upstream; nothing to add
    i = np.arange(1, num_new_points + 1, dtype=np.float64)
    if at_start:
        new_points = np.stack([p1[0] - i * step_x * dist, p1[1] - i * step_y * dist], axis=1)
        return np.vstack((new_points[::-1], points))
    new_points = np.stack([p2[0] + i * step_x * dist, p2[1] + i * step_y * dist], axis=1)
    return np.vstack((points, new_points))


def extend_ref_path_both_ends(ref_path, extension_length=30):
    """Extend the reference path on both ends by a default length (:54-58)."""
    extended_start = extend_path_linearly(ref_path, extension_length, at_start=True)
    return extend_path_linearly(extended_start, extension_length, at_start=False)


def extend_points(points):
    """Prepend points along the direction of the first segment, about 5 m worth (:61-77)."""
    p1, p2 = points[0], points[1]
    delta_x = p2[0] - p1[0]
    delta_y = p2[1] - p1[1]
    dist = _distance(points[0], points[1])
    num_new_points = int(5 / dist)
    if num_new_points == 0:
        return np.asarray(points)
    i = np.arange(1, num_new_points + 1, dtype=np.float64)
    new_points = np.stack([p1[0] - i * delta_x, p1[1] - i * delta_y], axis=1)
    return np.vstack((new_points[::-1], points))


def extend_points_end(points, extension_length=30):
    """Append points along the direction of the last segment (:80-99)."""
    p1, p2 = points[-2], points[-1]
    delta_x = p2[0] - p1[0]
    delta_y = p2[1] - p1[1]
    dist = _distance(p1, p2)
    if dist == 0:
        return points
    num_new_points = int(extension_length / dist)
    if num_new_points == 0:
        return np.asarray(points)
    i = np.arange(1, num_new_points + 1, dtype=np.float64)
    new_points = np.stack([p2[0] + i * delta_x, p2[1] + i * delta_y], axis=1)
    return np.vstack((points, new_points))


def extend_ref_path(ref_path, init_pos):
    """Prepend points when the planning position (shifted to the rear axle) is closest to the first vertex (:102-108)."""
    d2 = (ref_path[:, 0] - init_pos[0]) ** 2 + (ref_path[:, 1] - init_pos[1]) ** 2
    close_point = ref_path[int(np.argmin(d2))]  # min(..., key=distance): first minimum
    if close_point[0] == ref_path[0, 0] and close_point[1] == ref_path[0, 1]:
        ref_path = extend_points(ref_path)
    return ref_path


def _unique_rows_keep_order(a: np.ndarray) -> np.ndarray:
    _, idx = np.unique(a, axis=0, return_index=True)
    return a[np.sort(idx)]


def smooth_ref_path(reference: np.ndarray, smoothing_interval: float = 4):
    """Cubic B-spline through every t-th vertex, resampled at 1 m (:110-134)."""
    from scipy.interpolate import splev, splprep
    reference = _unique_rows_keep_order(np.asarray(reference, dtype=np.float64))
    distances = np.sqrt(np.sum((reference[0:-2:2] - reference[1:-1:2]) ** 2, axis=1))
    dist_sum_in_m = np.round(np.sum(distances), 3)
    average_dist_in_m = 0.125
    t = int(smoothing_interval / average_dist_in_m)  # smoothing_interval metres per control point
    reference = reference[::t]
    spline_discretization = int(6 * dist_sum_in_m)
    tck, u = splprep(reference.T, u=None, k=3, s=0.0)
    u_new = np.linspace(u.min(), u.max(), spline_discretization)
    x_new, y_new = splev(u_new, tck, der=0)
    reference = np.array([x_new, y_new]).transpose()
    reference = resample_polyline(reference, 1)
    return _unique_rows_keep_order(reference)


def extrapolate_ref_path(reference_path: np.ndarray, resample_step: float = 0.25) -> np.ndarray:
    """Extrapolates the end of the reference path along its last segment (:158-169)."""
    p = np.poly1d(np.polyfit(reference_path[-2:, 0], reference_path[-2:, 1], 1))
    x = 2.3 * reference_path[-1, 0] - reference_path[-2, 0]
    new_polyline = np.concatenate((reference_path, np.array([[x, p(x)]])), axis=0)
    return resample_polyline(new_polyline, step=resample_step)


def preprocess_ref_path(ref_path: np.ndarray, resample_step: float = 0.1, max_curv_desired: float = 0.1):
    """Corner cutting + resampling until the maximum curvature is below the limit (:172-184)."""
    ref_path_preprocessed = deepcopy(ref_path)
    max_curv = max_curv_desired + 0.2
    while max_curv > max_curv_desired:
        ref_path_preprocessed = np.array(chaikins_corner_cutting(ref_path_preprocessed))
        ref_path_preprocessed = resample_polyline(ref_path_preprocessed, resample_step)
        abs_curv = compute_curvature_from_polyline(ref_path_preprocessed)
        max_curv = max(abs_curv)
    return ref_path_preprocessed


def prepare_reference_path(route_reference_path: np.ndarray) -> np.ndarray:
    """What FrenetPlannerInterface does with the route planner's polyline (frenet_interface.py:110-114)."""
    return smooth_ref_path(extend_ref_path_both_ends(np.asarray(route_reference_path, dtype=np.float64)))
