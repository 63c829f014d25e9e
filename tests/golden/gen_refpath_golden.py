"""Golden vectors for the reference-path preparation (frenetix_motion_planner_amd/ref_path.py), produced by the
REFERENCE's own functions in cr_scenario_handler/utils/utils_coordinate_system.py (build container only):

    python tests/golden/gen_refpath_golden.py

commonroad_dc is not installed; the helper the prepared path needs from it (resample_polyline) is supplied to the
imported module from the build's restatement, so the vectors pin everything the reference itself does
(de-duplication, decimation, SciPy spline, extension arithmetic) -- not that helper.
Fixtures are data only: input polylines and the reference's output polylines.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402


def polylines():
    rng = np.random.default_rng(7)
    out = {}
    # route-planner style: dense (0.125 m) centre line, straight then a 90 degree left turn of radius 12 m
    s = np.arange(0, 40, 0.125)
    straight = np.stack([s, np.zeros_like(s)], axis=1)
    a = np.arange(0.125 / 12, np.pi / 2, 0.125 / 12)
    arc = np.stack([40 + 12 * np.sin(a), 12 * (1 - np.cos(a))], axis=1)
    s2 = np.arange(0.125, 30, 0.125)
    up = np.stack([np.full_like(s2, 52.0), 12 + s2], axis=1)
    out["turn_left"] = np.vstack([straight, arc, up])
    # gentle S-curve with duplicated vertices and a little jitter
    x = np.arange(0, 120, 0.125)
    y = 3.0 * np.sin(x / 25.0) + rng.normal(0, 1e-3, x.shape)
    sc = np.stack([x, y], axis=1)
    out["scurve_dups"] = np.vstack([sc[:200], sc[199:200], sc[200:600], sc[599:600], sc[600:]])
    # short coarse polyline (1 m spacing)
    t = np.arange(0, 25.0, 1.0)
    out["coarse_diag"] = np.stack([t * 0.8, t * 0.6], axis=1)
    return out


def main():
    ref_harness.install()
    from frenetix_motion_planner_amd import ref_path as mine
    import cr_scenario_handler.utils.utils_coordinate_system as ucs
    ucs.resample_polyline = mine.resample_polyline
    # the two other commonroad_dc helpers preprocess_ref_path calls: the build's restatements (unpinned), so that the vectors pin
    # the reference's own loop
    from frenetix_motion_planner_amd.coordinate_system import compute_curvature_from_polyline
    ucs.chaikins_corner_cutting = mine.chaikins_corner_cutting
    ucs.compute_curvature_from_polyline = compute_curvature_from_polyline
    fx = {}
    for name, pl in polylines().items():
        fx[f"{name}/in"] = pl
        fx[f"{name}/extend_both_30"] = np.asarray(ucs.extend_ref_path_both_ends(pl))
        fx[f"{name}/extend_start_50"] = np.asarray(ucs.extend_path_linearly(pl, 50, at_start=True))
        fx[f"{name}/extend_points"] = np.asarray(ucs.extend_points(pl))
        fx[f"{name}/extend_ref_path_first"] = np.asarray(ucs.extend_ref_path(pl, pl[0] + np.array([-0.3, 0.1])))
        fx[f"{name}/extend_ref_path_mid"] = np.asarray(ucs.extend_ref_path(pl, pl[len(pl) // 2]))
        fx[f"{name}/extend_points_end_30"] = np.asarray(ucs.extend_points_end(pl))
        fx[f"{name}/extrapolate"] = np.asarray(ucs.extrapolate_ref_path(pl))
        if name == "coarse_diag":
            bent = np.vstack([pl[:12], pl[12:] + np.array([0.0, 1.0]) * np.arange(len(pl) - 12)[:, None] * 0.8])
            fx[f"{name}/bent_in"] = bent
            fx[f"{name}/preprocessed"] = np.asarray(ucs.preprocess_ref_path(bent))
        if name != "coarse_diag":
            fx[f"{name}/smooth"] = np.asarray(ucs.smooth_ref_path(pl))
            fx[f"{name}/smooth_8"] = np.asarray(ucs.smooth_ref_path(pl, 8))
            fx[f"{name}/prepared"] = np.asarray(ucs.smooth_ref_path(ucs.extend_ref_path_both_ends(pl)))
    path = os.path.join(HERE, "refpath_golden.npz")
    np.savez_compressed(path, **fx)
    for k, v in fx.items():
        print(f"{k:36s} {v.shape}")
    print(os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
