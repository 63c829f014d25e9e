#!/usr/bin/env python3
"""Pin the three third-party pieces of the hot path against the real thing -- WHEN it is importable.

The reference calls into commonroad-drivability-checker (C++, not in the reference tree, not installable offline) for
  a11  (s, d) -> (x, y)            pycrccosy.CurvilinearCoordinateSystem.convert_to_cartesian_coords (utils_coordinate_system.py:263-270)
  a18  dynamic-obstacle collision   trajectory_preprocess_obb_sum + trajectories_collision_dynamic_obstacles (collision_check.py:110-200)
  a19  road-boundary collision      create_road_boundary_obstacle + trajectories_collision_static_obstacles (planner.py:362-381,550-565)
and this repository restates them ("parity unpinned").  This script is the one command that pins them once
`commonroad_dc` imports (a machine with the reference's environment):

    python tests/golden/pin_third_party.py            # report only
    python tests/golden/pin_third_party.py --write    # additionally store tests/golden/third_party_pins.npz

What it does
  1. projection: for every golden scenario's reference polyline and a seeded cloud of (s, d), CCosy's Cartesian points against
     the four readings this package implements (CoordinateSystem(pseudo_normal=, vertex_tangent=)); prints max |delta| per
     reading and names the one that matches within 1e-9 m -- make that the default in coordinate_system.py if it is not.
  2. collision: for the golden scenarios with obstacles, pycrcc's verdict per returned trajectory (OBB-sum pre-processing, the
     reference's own call sequence) against the oracle's collision flags; prints the disagreements.
  (a19, the road boundary, needs commonroad-io's scenario objects for create_road_boundary_obstacle: not scripted yet -- same
  pattern: the boundary obstacle from the scenario file, trajectories_collision_static_obstacles per stored trajectory, against
  `leaving_road_at` of the oracle.)
Without commonroad_dc it prints what is missing and exits 0 (nothing to pin here: this container and the GPU boxes).

`tests/test_third_party_pins.py` reads third_party_pins.npz when present and holds the oracle to it.
"""
import argparse
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def have_commonroad_dc():
    try:
        import commonroad_dc.pycrccosy  # noqa: F401
        import commonroad_dc.pycrcc  # noqa: F401
        return True
    except Exception as ex:   # ImportError, or a stub left in sys.modules by the golden harness
        print(f"commonroad_dc not importable ({type(ex).__name__}: {ex}) -- nothing pinned; the oracle stays 'parity unpinned' for "
              "a11 / a18 / a19 (DESIGN.md 4)")
        return False


def golden_references():
    """{name: reference polyline} of every golden plan-step scenario (the arrays the reference planner saw)."""
    from tests.fixtures import golden_names, load_golden
    return {name: np.asarray(load_golden(name)["ref_xy"], dtype=np.float64) for name in golden_names()}


def pin_projection(out):
    from commonroad_dc.pycrccosy import CurvilinearCoordinateSystem
    from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
    readings = [dict(pseudo_normal=pn, vertex_tangent=vt) for pn in (False, True) for vt in ("chord", "bisector")]
    worst = {i: 0.0 for i in range(len(readings))}
    for name, ref in golden_references().items():
        # the reference builds CCosy with these limits (utils_coordinate_system.py:229-230)
        ccosy = CurvilinearCoordinateSystem(ref, 25.0, 0.1)
        rng = np.random.default_rng(abs(hash(name)) % (1 << 31))
        ours = [CoordinateSystem(ref, **r) for r in readings]
        s_all = rng.uniform(ours[0].ref_pos[2], ours[0].ref_pos[-3], 4000)
        d_all = rng.uniform(-3.0, 3.0, s_all.size)
        pts = []
        for s, d in zip(s_all, d_all):
            try:
                pts.append(np.asarray(ccosy.convert_to_cartesian_coords(float(s), float(d))))
            except Exception:
                pts.append(np.array([np.nan, np.nan]))
        pts = np.asarray(pts)
        ok = np.isfinite(pts[:, 0])
        out[f"proj/{name}/s"], out[f"proj/{name}/d"], out[f"proj/{name}/xy"] = s_all[ok], d_all[ok], pts[ok]
        for i, cs in enumerate(ours):
            mine = np.asarray([cs.convert_to_cartesian_coords(float(s), float(d)) for s, d in zip(s_all[ok], d_all[ok])])
            worst[i] = max(worst[i], float(np.abs(mine - pts[ok]).max()))
    print("projection: max |delta| against CCosy over all golden references")
    for i, r in enumerate(readings):
        print(f"   pseudo_normal={r['pseudo_normal']!s:5} vertex_tangent={r['vertex_tangent']:8}  {worst[i]:.3e} m"
              + ("   <- matches" if worst[i] < 1e-9 else ""))
    best = min(worst, key=worst.get)
    print(f"   closest reading: {readings[best]} ({worst[best]:.3e} m); the package default is pseudo_normal=False, vertex_tangent='chord'")
    return readings[best], worst[best]


def pin_collision(out):
    """pycrcc's verdicts for the stored trajectories of the golden scenarios with obstacles, the reference's own sequence
    (collision_check.py:110-200): time-variant collision object of the ego rectangle per step, OBB-sum pre-processing, the
    obstacles' predicted rectangles, trajectories_collision_dynamic_obstacles."""
    import commonroad_dc.pycrcc as pycrcc
    from commonroad_dc.collision.trajectory_queries import trajectory_queries
    from oracle import oracle
    from tests.fixtures import golden_names, inputs_from_fixture, load_golden
    n_dis = n_all = 0
    for name in golden_names():
        inp = inputs_from_fixture(load_golden(name), oracle.build_obstacle_hulls)
        preds = inp.predictions or {}
        if not preds:
            continue
        res = oracle.plan_step(inp)
        veh = inp.vehicle
        t0 = 0
        cos = []
        for oid, pr in preds.items():
            n = len(pr["pos_list"])
            if n <= 2:   # collision_check.py:165-168
                continue
            tv = pycrcc.TimeVariantCollisionObject(t0 + 1)
            for j in range(n):
                px, py = pr["pos_list"][j]
                tv.append_obstacle(pycrcc.RectOBB(0.5 * pr["shape"]["length"], 0.5 * pr["shape"]["width"],
                                                  float(pr["orientation_list"][j]), float(px), float(py)))
            cos.append(trajectory_queries.trajectory_preprocess_obb_sum(tv)[0])
        ids = np.nonzero(res["selectable"])[0]
        verdict = np.zeros(len(ids), dtype=bool)
        for q, c in enumerate(ids):
            x, y, th = res["planes"][c, 0], res["planes"][c, 1], res["planes"][c, 2]
            tv = pycrcc.TimeVariantCollisionObject(t0)
            for i in range(len(x)):
                tv.append_obstacle(pycrcc.RectOBB(0.5 * veh.length, 0.5 * veh.width, float(th[i]),
                                                  float(x[i] + veh.wb_rear_axle * np.cos(th[i])), float(y[i] + veh.wb_rear_axle * np.sin(th[i]))))
            ego = trajectory_queries.trajectory_preprocess_obb_sum(tv)[0]
            verdict[q] = trajectory_queries.trajectories_collision_dynamic_obstacles([ego], cos, method="fcl")[0] != -1
        mine = res["collision"][ids]
        fragile = res["margin"][ids] < 1e-9
        dis = (verdict != mine) & ~fragile
        n_dis += int(dis.sum()); n_all += len(ids)
        out[f"coll/{name}/ids"], out[f"coll/{name}/pycrcc"] = ids, verdict
        print(f"   {name}: {len(ids)} selectable, pycrcc collides {int(verdict.sum())}, oracle {int(mine.sum())}, disagreements {int(dis.sum())}")
    print(f"collision: {n_dis} disagreements over {n_all} trajectories")
    return n_dis


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="store tests/golden/third_party_pins.npz")
    args = ap.parse_args()
    if not have_commonroad_dc():
        return 0
    out = {}
    reading, err = pin_projection(out)
    out["proj/reading"] = np.array([int(reading["pseudo_normal"]), int(reading["vertex_tangent"] == "bisector")])
    out["proj/max_err"] = np.array([err])
    try:
        out["coll/disagreements"] = np.array([pin_collision(out)])
    except Exception as ex:   # the collision API moved between commonroad_dc releases: report, keep the projection pin
        print(f"collision pin not taken ({type(ex).__name__}: {ex})")
    if args.write:
        np.savez_compressed(os.path.join(HERE, "third_party_pins.npz"), **out)
        print("wrote tests/golden/third_party_pins.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())
