#!/usr/bin/env python3
"""How far apart the plausible readings of CCosy's (s, d) -> (x, y) are (CCosy itself is not in the reference tree):
  A  this package: foot point + d * NORMALISED linear interpolation of the vertex normals, vertex tangent = chord P[i+1] - P[i-1]
  B  the same without the normalisation (d as a pseudo-distance along the un-normalised pseudo-normal)
  C  as A with the vertex tangent = bisector of the adjacent unit segment directions
on the config-1 route (ZAM_Tjunction-1_42_T-1 after prepare_reference_path) and on the synthetic arc, for |d| <= 3 m.  CPU only."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frenetix_motion_planner_amd import commonroad_xml as crx, synthetic
from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem, vertex_normals


def bisector_normals(p):
    u = np.diff(p, axis=0)
    u /= np.linalg.norm(u, axis=1)[:, None]
    t = np.empty_like(p)
    t[1:-1] = u[1:] + u[:-1]
    t[0], t[-1] = u[0], u[-1]
    t /= np.linalg.norm(t, axis=1)[:, None]
    return np.stack([-t[:, 1], t[:, 0]], axis=1)


def project(p, s_knots, normals, s, d, normalise=True):
    k = np.clip(np.searchsorted(s_knots, s, side="right") - 1, 0, len(p) - 2)
    lam = (s - s_knots[k]) / (s_knots[k + 1] - s_knots[k])
    base = p[k] + lam[:, None] * (p[k + 1] - p[k])
    n = normals[k] + lam[:, None] * (normals[k + 1] - normals[k])
    if normalise:
        n = n / np.linalg.norm(n, axis=1)[:, None]
    return base + d[:, None] * n


def report(name, ref):
    cs = CoordinateSystem(ref)
    s_k = cs.ref_pos
    rng = np.random.default_rng(7)
    s = rng.uniform(s_k[1], s_k[-2], 200000)
    d = rng.uniform(-3.0, 3.0, s.size)
    A = project(ref, s_k, vertex_normals(ref), s, d, True)
    B = project(ref, s_k, vertex_normals(ref), s, d, False)
    Cc = project(ref, s_k, bisector_normals(ref), s, d, True)
    seg = np.diff(s_k)
    dth = np.abs(np.diff(np.unwrap(np.arctan2(np.diff(ref[:, 1]), np.diff(ref[:, 0])))))
    print(f"{name}: {len(ref)} knots, segment length {seg.min():.3f} .. {seg.max():.3f} m, largest heading change between segments {dth.max():.4f} rad")
    print(f"   A vs B (normalised vs un-normalised pseudo-normal): max |dx,dy| = {np.abs(A - B).max():.3e} m")
    print(f"   A vs C (chord vs bisector vertex tangent):          max |dx,dy| = {np.abs(A - Cc).max():.3e} m")


sc = crx.read_scenario_json(os.path.join(ROOT, "tests", "golden", "ZAM_Tjunction-1_42_T-1.scenario.json"))
from frenetix_motion_planner_amd import ref_path
route = ref_path.resample_polyline(sc.route_reference_path(sc.planning_problems[60000]), 0.125)   # as FrenetPlannerInterfaceHip.__init__
report("config-1 route (ZAM_Tjunction-1_42_T-1, prepared)", np.asarray(ref_path.prepare_reference_path(route), dtype=np.float64))
report("synthetic arc (kappa 0.01, 0.5 m knots)", synthetic.reference_polyline("arc", 400, 0.5, 0.01))
report("synthetic arc, knot spacing jittered +-30 %", synthetic.reference_polyline("arc", 400, 0.5, 0.01, knot_jitter=0.3))
