"""The oracle's collision primitives against elementary geometry written independently of them.

pycrcc -- what the reference calls for the collision walk -- is not in the reference tree, so the OBB-sum hull and the
separating-axis test cannot be pinned against it.  What can be checked: the axis test decides exactly what "two convex
quadrilaterals share a point" means (edges cross or a corner lies inside), and the hull of two boxes contains both and is
tight in its own frame.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def corners(box):
    cx, cy, ex, ey, h1, h2 = box
    e, f = np.array([ex, ey]), np.array([-ey, ex])
    c = np.array([cx, cy])
    return np.array([c + h1 * e + h2 * f, c - h1 * e + h2 * f, c - h1 * e - h2 * f, c + h1 * e - h2 * f])


def _orient(a, b, c):
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])


def _segments_cross(p, q, r, s):
    """closed segments pq and rs share a point; with the smallest |orientation| met, as a margin for near-degenerate pairs"""
    d1, d2, d3, d4 = _orient(r, s, p), _orient(r, s, q), _orient(p, q, r), _orient(p, q, s)
    return (d1 * d2 <= 0) and (d3 * d4 <= 0), min(abs(d1), abs(d2), abs(d3), abs(d4))


def _inside(poly, pt):
    """pt in the closed convex polygon (counter-clockwise corners); smallest |orientation| as a margin"""
    o = [_orient(poly[i], poly[(i + 1) % 4], pt) for i in range(4)]
    return all(v >= 0 for v in o), min(abs(v) for v in o)


def polygons_intersect(A, B):
    """(shares a point, margin): edge crossings or containment -- no projections, no axes"""
    hit, margin = False, np.inf
    for i in range(4):
        for j in range(4):
            h, m = _segments_cross(A[i], A[(i + 1) % 4], B[j], B[(j + 1) % 4])
            hit, margin = hit or h, min(margin, m)
    for poly, pts in ((A, B), (B, A)):
        for pt in pts:
            h, m = _inside(poly, pt)
            hit, margin = hit or h, min(margin, m)
    return hit, margin


def _random_box(rng, spread):
    th = rng.uniform(-np.pi, np.pi)
    return np.array([rng.uniform(-spread, spread), rng.uniform(-spread, spread), np.cos(th), np.sin(th),
                     rng.uniform(0.3, 4.0), rng.uniform(0.3, 2.0)])


def test_axis_test_equals_polygon_intersection():
    rng = np.random.default_rng(42)
    L = oracle.lib()
    n_hit = n_checked = 0
    for t in range(20000):
        a, b = _random_box(rng, 4.0), _random_box(rng, 4.0)
        if t % 7 == 0:   # aligned and nested pairs: the containment branch, parallel edges
            b[2:4] = a[2:4] if t % 14 == 0 else (-a[3], a[2])
        if t % 11 == 0:
            b[:2] = a[:2] + rng.uniform(-0.1, 0.1, 2)
        want, margin = polygons_intersect(corners(a), corners(b))
        if margin < 1e-9:
            continue   # touching to rounding: the two formulations may round differently
        got = bool(L.fxo_obb_overlap(_pd(a), _pd(b)))
        assert got == want, (a, b, margin)
        n_checked += 1
        n_hit += want
    assert n_checked > 19000 and 0.2 < n_hit / n_checked < 0.8   # both outcomes well represented


def test_touching_boxes_collide():
    """strict `>` for separation: boxes that share exactly an edge or a corner collide"""
    L = oracle.lib()
    a = np.array([0.0, 0.0, 1.0, 0.0, 2.0, 1.0])
    for b in (np.array([4.0, 0.0, 1.0, 0.0, 2.0, 1.0]),      # shared edge
              np.array([4.0, 2.0, 1.0, 0.0, 2.0, 1.0]),      # shared corner
              np.array([3.0, 0.0, 0.0, 1.0, 1.0, 1.0])):     # shared edge, rotated by a quarter turn
        assert L.fxo_obb_overlap(_pd(a), _pd(b)) == 1
        away = b.copy(); away[0] += 1e-9
        assert L.fxo_obb_overlap(_pd(a), _pd(away)) == 0


def test_hull_contains_both_boxes_and_is_tight():
    rng = np.random.default_rng(7)
    L = oracle.lib()
    out = np.zeros(6)
    for t in range(5000):
        c0, c1 = rng.uniform(-30, 30, 2), None
        th0 = rng.uniform(-np.pi, np.pi)
        th1 = th0 + (rng.uniform(-0.4, 0.4) if t % 5 else rng.uniform(-np.pi, np.pi))
        if t % 97 == 0:
            th1 = th0 + np.pi   # opposite headings: the axis falls back to the first box's
        c1 = c0 + rng.uniform(0.0, 3.0) * np.array([np.cos(th0), np.sin(th0)]) + rng.uniform(-0.3, 0.3, 2)
        hl, hw = rng.uniform(1.0, 3.0), rng.uniform(0.5, 1.2)
        u0, u1 = np.array([np.cos(th0), np.sin(th0)]), np.array([np.cos(th1), np.sin(th1)])
        L.fxo_obb_hull(_pd(c0), _pd(u0), _pd(c1), _pd(u1), hl, hw, _pd(out))
        e, f, c = out[2:4], np.array([-out[3], out[2]]), out[:2]
        assert abs(np.hypot(*e) - 1.0) < 1e-12
        pts = np.concatenate([corners(np.array([*c0, *u0, hl, hw])), corners(np.array([*c1, *u1, hl, hw]))])
        p1, p2 = (pts - c) @ e, (pts - c) @ f
        # containment, and tightness: some corner on each of the four sides
        assert p1.max() <= out[4] + 1e-9 and p1.min() >= -out[4] - 1e-9
        assert p2.max() <= out[5] + 1e-9 and p2.min() >= -out[5] - 1e-9
        assert abs(p1.max() - out[4]) < 1e-9 and abs(p1.min() + out[4]) < 1e-9
        assert abs(p2.max() - out[5]) < 1e-9 and abs(p2.min() + out[5]) < 1e-9
        # the axis bisects the two headings (or is the first heading when they cancel)
        s = u0 + u1
        if np.hypot(*s) > 1e-9:
            assert abs(e[0] * s[1] - e[1] * s[0]) < 1e-9 * max(1.0, np.hypot(*s)) and e @ s > 0
        else:
            assert np.allclose(e, u0, atol=1e-9)


def test_obstacle_hulls_are_hulls_of_consecutive_predictions():
    """fxo_build_obstacle_hulls: hull j = hull of predicted boxes (j, j+1); obstacles with <= 2 predictions are skipped"""
    rng = np.random.default_rng(3)
    L = oracle.lib()
    n = 12
    pos = np.cumsum(rng.uniform(0.2, 1.0, (n, 2)), axis=0)
    yaw = np.cumsum(rng.uniform(-0.1, 0.1, n))
    hulls = oracle.build_obstacle_hulls(n, pos, yaw, 4.6, 1.9)
    assert hulls.shape == (n - 1, 6)
    one = np.zeros(6)
    for j in range(n - 1):
        u0 = np.array([np.cos(yaw[j]), np.sin(yaw[j])]); u1 = np.array([np.cos(yaw[j + 1]), np.sin(yaw[j + 1])])
        L.fxo_obb_hull(_pd(np.ascontiguousarray(pos[j])), _pd(u0), _pd(np.ascontiguousarray(pos[j + 1])), _pd(u1), 2.3, 0.95, _pd(one))
        assert np.allclose(hulls[j], one, rtol=0, atol=1e-12)
    assert len(oracle.build_obstacle_hulls(2, pos[:2], yaw[:2], 4.6, 1.9)) == 0


def _first_contact_by_clipping(planes, pieces, veh, d_reach, n_steps):
    """First step whose footprint rectangle meets a boundary piece, by Liang-Barsky clipping of every piece against
    the rectangle in its own frame -- no separating axes.  Returns (step or -1, smallest clipping margin met)."""
    x, y, th, d = planes[0], planes[1], planes[2], planes[8]
    a, b = veh.length / 2, veh.width / 2
    p0 = pieces[:, :2] - pieces[:, 2:]
    dv = 2.0 * pieces[:, 2:]
    margin = np.inf
    for i in range(n_steps):
        if x[i] == 0.0 and y[i] == 0.0:      # first step outside the projection domain: not tested from here on
            return -1, margin
        if abs(d[i]) > d_reach:
            return i, margin
        cu, su = np.cos(th[i]), np.sin(th[i])
        c = np.array([x[i] + veh.wb_rear_axle * cu, y[i] + veh.wb_rear_axle * su])
        q = p0 - c
        qx, qy = q[:, 0] * cu + q[:, 1] * su, -q[:, 0] * su + q[:, 1] * cu          # piece start in the box frame
        ex, ey = dv[:, 0] * cu + dv[:, 1] * su, -dv[:, 0] * su + dv[:, 1] * cu      # piece direction in the box frame
        t0, t1 = np.zeros(len(q)), np.ones(len(q))
        alive = np.ones(len(q), bool)
        for p, r in ((-ex, qx + a), (ex, a - qx), (-ey, qy + b), (ey, b - qy)):     # p t <= r for the four sides
            par = p == 0
            alive &= ~(par & (r < 0))
            with np.errstate(divide="ignore", invalid="ignore"):
                t = r / p
            t0 = np.where(~par & (p < 0), np.maximum(t0, t), t0)
            t1 = np.where(~par & (p > 0), np.minimum(t1, t), t1)
        gap = t1 - t0
        margin = min(margin, float(np.abs(gap[alive]).min()) if alive.any() else np.inf)
        if (alive & (gap >= 0)).any():
            return i, margin
    return -1, margin


@pytest.mark.parametrize("kw", [dict(ref_kind="arc", v0=10.0, grid=(5, 7, 13), road_half_width=2.6),
                                dict(ref_kind="scurve", kappa=0.03, v0=8.0, grid=(4, 5, 17), road_half_width=2.2, seed=3),
                                dict(ref_kind="arc", v0=9.0, d0=1.2, grid=(4, 5, 13), road_half_width=2.6)])
def test_road_boundary_contacts_by_clipping(kw):
    """The oracle's first off-road step (axis test of DESIGN 4.3 on the binned pieces' brute force) against segment clipping."""
    from frenetix_motion_planner_amd import synthetic
    inp = synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, draw_traj_set=True, kinematic_debug=True, **kw)
    out = oracle.plan_step(inp)
    seg = np.asarray(inp.road_boundary).reshape(-1, 4)       # the raw segments (ax, ay, bx, by), not the engine's pieces
    pieces = np.concatenate([0.5 * (seg[:, :2] + seg[:, 2:]), 0.5 * (seg[:, 2:] - seg[:, :2])], axis=1)
    d_reach = inp._bound["d_reach"]
    checked = hits = 0
    for g in np.nonzero(out["selectable"])[0]:
        step, margin = _first_contact_by_clipping(out["planes"][g], pieces, inp.vehicle, d_reach, inp.N + 1)
        if margin < 1e-9:
            continue   # a piece grazing the footprint to rounding
        assert step == out["boundary_step"][g], (g, step, out["boundary_step"][g])
        checked += 1
        hits += step >= 0
    assert checked > 50 and 0 < hits < checked
