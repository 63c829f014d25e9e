"""Host-side reference-path preparation.

`CoordinateSystem` mirrors cr_scenario_handler/utils/utils_coordinate_system.py:187-274 of the
reference: it owns the reference polyline and the per-knot arrays ref_pos / ref_theta / ref_curv /
ref_curv_d that the hot path interpolates, and converts between curvilinear (s, d) and Cartesian
(x, y).  In the reference these come from commonroad-drivability-checker (C++ CCosy and
commonroad_dc.geometry.util), which is not in the reference tree; they are restated here
(parity unpinned, DESIGN.md "Third-party arithmetic"):

  pathlength  : cumulative Euclidean segment length
  orientation : atan2 of the forward segment, last value repeated
  curvature   : (x'y'' - x''y') / (x'^2 + y'^2)^(3/2) with np.gradient
  projection  : foot point on the segment containing s, offset d along the normalised linear
                interpolation of the vertex normals (vertex normal = left normal of P[i+1]-P[i-1])

All of this runs once per reference-path change on the host; the arrays are kernel inputs.
"""
import math

import numpy as np

_UIDS = __import__("itertools").count(1)


def compute_pathlength_from_polyline(polyline: np.ndarray) -> np.ndarray:
    seg = np.diff(np.asarray(polyline, dtype=np.float64), axis=0)
    out = np.zeros(len(polyline))
    acc = 0.0
    for i in range(len(seg)):
        acc = acc + math.sqrt(seg[i, 0] * seg[i, 0] + seg[i, 1] * seg[i, 1])
        out[i + 1] = acc
    return out


def compute_orientation_from_polyline(polyline: np.ndarray) -> np.ndarray:
    p = np.asarray(polyline, dtype=np.float64)
    seg = p[1:] - p[:-1]
    th = np.arctan2(seg[:, 1], seg[:, 0])
    return np.append(th, th[-1])


def compute_curvature_from_polyline(polyline: np.ndarray) -> np.ndarray:
    p = np.asarray(polyline, dtype=np.float64)
    x_d = np.gradient(p[:, 0])
    x_dd = np.gradient(x_d)
    y_d = np.gradient(p[:, 1])
    y_dd = np.gradient(y_d)
    return (x_d * y_dd - x_dd * y_d) / ((x_d ** 2 + y_d ** 2) ** (3. / 2.))


def vertex_normals(polyline: np.ndarray, vertex_tangent: str = "chord") -> np.ndarray:
    """Unit left normals at the vertices.  vertex_tangent: "chord" (default) -- direction of P[i+1] - P[i-1] -- or "bisector" --
    the normalised sum of the two adjacent unit segment directions; the two agree on uniformly spaced polylines and differ by
    up to 2 mm at d = 3 m when the knot spacing varies by 30 % (tools/projection_variants.py).  Which one CCosy uses is
    decided by tests/golden/pin_third_party.py once commonroad_dc is importable (parity unpinned)."""
    p = np.asarray(polyline, dtype=np.float64)
    t = np.empty_like(p)
    if vertex_tangent == "bisector":
        u = p[1:] - p[:-1]
        u = u / np.sqrt(u[:, 0] * u[:, 0] + u[:, 1] * u[:, 1])[:, None]
        t[1:-1] = u[1:] + u[:-1]
        t[0], t[-1] = u[0], u[-1]
    elif vertex_tangent == "chord":
        t[1:-1] = p[2:] - p[:-2]
        t[0] = p[1] - p[0]
        t[-1] = p[-1] - p[-2]
    else:
        raise ValueError(f"vertex_tangent must be 'chord' or 'bisector', not {vertex_tangent!r}")
    nrm = np.sqrt(t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1])
    t = t / nrm[:, None]
    return np.stack([-t[:, 1], t[:, 0]], axis=1)


def time_power_table(dt: float, n_samples: int) -> np.ndarray:
    """tpow[k, i] = round(t_i**(k+1), 10) with t_i = round(i*dt, 5): the reference's time grid
    (reactive_planner.py:296-300; np.arange(0, T+dt, dt)[i] == i*dt for every T)."""
    t = np.round(np.arange(n_samples) * dt, 5)
    return np.stack([t] + [np.round(np.power(t, k), 10) for k in (2, 3, 4, 5)]).astype(np.float64)


def simpson_even_correction(dt: float):
    """alpha, beta, eta of scipy.integrate.simpson's correction for an even number of samples with
    h = [dt, dt] (scipy/integrate/_quadrature.py, 'simpson' rule; used by the zero-weight-by-default
    jerk / orientation_offset costs, partial_cost_functions.py:41-46,146-151)."""
    h = np.asarray([dt, dt], dtype=np.float64)
    alpha = (2 * h[1] ** 2 + 3 * h[0] * h[1]) / (6 * (h[1] + h[0]))
    beta = (h[1] ** 2 + 3.0 * h[0] * h[1]) / (6 * h[0])
    eta = (1 * h[1] ** 3) / (6 * h[0] * (h[0] + h[1]))
    return float(alpha), float(beta), float(eta)


def make_valid_orientation(angle: float) -> float:
    """commonroad.common.util.make_valid_orientation (restated): into [-2pi, 2pi]."""
    two_pi = 2.0 * np.pi
    while angle > two_pi:
        angle = angle - two_pi
    while angle < -two_pi:
        angle = angle + two_pi
    return angle


def interpolate_angle(x: float, x1: float, x2: float, y1: float, y2: float) -> float:
    """utils_coordinate_system.py:137-155."""
    delta = y2 - y1
    return make_valid_orientation(delta * (x - x1) / (x2 - x1) + y1)


class CoordinateSystem:
    """Reference path + curvilinear frame (utils_coordinate_system.py:187-274)."""

    def __init__(self, reference: np.ndarray, pseudo_normal: bool = False, vertex_tangent: str = "chord"):
        """pseudo_normal / vertex_tangent: the two open readings of CCosy's projection.  Default: d along the
        NORMALISED interpolated vertex normal, vertex tangent = chord.  pseudo_normal=True offsets along the un-normalised
        interpolated normal (FX_MODE_PROJ_PSEUDO_NORMAL in the kernel and the oracle; the inverse map follows)."""
        self.pseudo_normal = bool(pseudo_normal)
        self.vertex_tangent = vertex_tangent
        ref = np.ascontiguousarray(np.asarray(reference, dtype=np.float64))
        if ref.ndim != 2 or ref.shape[1] != 2 or ref.shape[0] < 3:
            raise ValueError("<CoordinateSystem>: reference must be an (M>=3, 2) polyline")
        self._reference = ref
        self._ref_pos = compute_pathlength_from_polyline(ref)
        if np.any(np.diff(self._ref_pos) <= 0):
            raise ValueError("<CoordinateSystem>: reference has repeated vertices")
        self._ref_curv = compute_curvature_from_polyline(ref)
        self._ref_theta = np.unwrap(compute_orientation_from_polyline(ref))
        self._ref_curv_d = np.gradient(self._ref_curv, self._ref_pos)
        self._ref_curv_dd = np.gradient(self._ref_curv_d, self._ref_pos)
        self._normals = vertex_normals(ref, vertex_tangent)
        self._c_args = None
        self._ref_lists = None   # (reference_at: the knot arrays as Python floats, built on first use)
        self.uid = next(_UIDS)   # identity that is never reused (id() of a collected object can be)

    reference = property(lambda self: self._reference)
    ref_pos = property(lambda self: self._ref_pos)
    ref_curv = property(lambda self: self._ref_curv)
    ref_curv_d = property(lambda self: self._ref_curv_d)
    ref_cruv_dd = property(lambda self: self._ref_curv_dd)  # (sic) reference spelling, :215
    ref_theta = property(lambda self: self._ref_theta)
    normals = property(lambda self: self._normals)

    def reference_at(self, s: float):
        """(theta_ref unwrapped, kappa_ref, kappa_ref') linearly interpolated at arc length s -- the segment is the one the
        reference's `argmax(ref_pos > s) - 1` picks (planner.py:580-597)."""
        rp = self._ref_pos
        # (np.argmax(rp > s): the first knot behind s, 0 when there is none -- as a binary search instead of a mask over the path)
        k = int(np.searchsorted(rp, s, side="right"))
        k = (k if k < len(rp) else 0) - 1
        lst = self._ref_lists   # the four arrays as Python floats: scalar arithmetic on NumPy scalars costs three times as much
        if lst is None:
            lst = self._ref_lists = (rp.tolist(), self._ref_theta.tolist(), np.asarray(self.ref_curv, dtype=np.float64).tolist(),
                                     np.asarray(self.ref_curv_d, dtype=np.float64).tolist())
        rpl, th, kc, kd = lst
        s = float(s)
        lam = (s - rpl[k]) / (rpl[k + 1] - rpl[k])
        theta = interpolate_angle(s, rpl[k], rpl[k + 1], th[k], th[k + 1])   # (th: unwrapped at construction)
        return theta, (kc[k + 1] - kc[k]) * lam + kc[k], (kd[k + 1] - kd[k]) * lam + kd[k]

    def frenet_state(self, x: float, y: float, heading: float, speed: float, acceleration: float, curvature: float,
                     arc_length_lateral: bool):
        """Cartesian vehicle state -> ([s, s', s''], [d, d', d'']) along this reference (Werling, A.3 / A.5; behaviour of
        planner.py:567-635).  With the heading error e = heading - theta_ref, the lateral scale w = 1 - kappa_ref d and the
        curvature mismatch m = curvature * w / cos e - kappa_ref:

            dd/ds   = w tan e                       d2d/ds2 = -(kappa_ref' d + kappa_ref dd/ds) tan e + w m / cos^2 e
            ds/dt   = speed cos e / w               d2s/dt2 = (acceleration - (ds/dt)^2 / cos e * (w tan e m - (kappa_ref' d +
                                                               kappa_ref dd/ds))) * cos e / w

        The lateral derivatives are returned per arc length (LOW_VEL_MODE, `arc_length_lateral`) or per time
        (d' = speed sin e, d'' = s'' dd/ds + s'^2 d2d/ds2).  Raises ValueError outside the projection domain and when the
        vehicle faces against the reference (s' < 0)."""
        import math
        s, d = self.convert_to_curvilinear_coords(x, y).tolist()
        theta_ref, k_ref, k_ref_s = self.reference_at(s)
        e = float(heading) - float(theta_ref)
        speed, acceleration, curvature = float(speed), float(acceleration), float(curvature)
        tan_e, cos_e = float(np.tan(e)), math.cos(e)   # (np.tan, not math.tan: the two differ in the last bit for some arguments)
        w = 1 - k_ref * d
        d_s = w * tan_e
        bend = k_ref_s * d + k_ref * d_s          # d/ds of (kappa_ref d)
        mismatch = curvature * w / cos_e - k_ref
        d_ss = -bend * tan_e + (w / cos_e ** 2) * mismatch
        s_t = speed * cos_e / w
        if s_t < 0:
            raise ValueError("the vehicle faces against the reference path (negative s'): initial state or reference incorrect")
        s_tt = acceleration - (s_t ** 2 / cos_e) * (w * tan_e * mismatch - bend)
        s_tt /= w / cos_e
        if arc_length_lateral:
            lat = [float(d), float(d_s), float(d_ss)]
        else:
            lat = [float(d), float(speed * math.sin(e)), float(s_tt * d_s + s_t ** 2 * d_ss)]
        return [float(s), float(s_t), float(s_tt)], lat

    def segment_of(self, s: float) -> int:
        k = int(np.searchsorted(self._ref_pos, s, side="right")) - 1
        return min(max(k, 0), len(self._ref_pos) - 2)

    def convert_to_cartesian_coords(self, s: float, d: float):
        rp = self._ref_pos
        if not (rp[0] <= s <= rp[-1]):
            return None
        k = self.segment_of(s)
        lam = (s - rp[k]) / (rp[k + 1] - rp[k])
        p = self._reference[k] + lam * (self._reference[k + 1] - self._reference[k])
        n = self._normals[k] + lam * (self._normals[k + 1] - self._normals[k])
        if self.pseudo_normal:
            return np.array([p[0] + d * n[0], p[1] + d * n[1]])
        nn = math.sqrt(n[0] * n[0] + n[1] * n[1])
        return np.array([p[0] + d * (n[0] / nn), p[1] + d * (n[1] / nn)])

    def convert_to_curvilinear_coords(self, x: float, y: float) -> np.ndarray:
        """Inverse of convert_to_cartesian_coords: the (s, d) with smallest |d| whose foot point and
        interpolated normal pass through (x, y).  Raises ValueError outside the projection domain
        (planner.py:574-578 expects that).  Host geometry of the library (fx_cs_to_curvilinear): on every segment the
        collinearity of foot point, interpolated normal and the point is a quadratic in the segment parameter."""
        import ctypes as C
        from ._lib import lib
        if self._c_args is None:   # contiguous views, kept alive with the object
            keep = (np.ascontiguousarray(self._reference), np.ascontiguousarray(self._normals), np.ascontiguousarray(self._ref_pos))
            self._c_args = (keep, len(keep[0]), *[a.ctypes.data for a in keep], (C.c_double * 2)())
        _, M, p_ref, p_nrm, p_pos, out = self._c_args
        if lib().fx_cs_to_curvilinear_ex(M, p_ref, p_nrm, p_pos, float(x), float(y), int(self.pseudo_normal), C.addressof(out)) != 0:
            raise ValueError("<CoordinateSystem>: point outside projection domain")
        return np.array([out[0], out[1]])
