"""Rebuild PlanInputs from a golden fixture (tests/golden/*.npz)."""
import glob
import json
import os

import numpy as np

from frenetix_motion_planner_amd import CoordinateSystem, PlanInputs, VehicleParams, pack_predictions

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def predictions_from_fixture(fx):
    if "pred_keys" not in fx:
        return None
    preds = {}
    for i, k in enumerate(fx["pred_keys"]):
        preds[int(k)] = dict(pos_list=fx["pred_pos"][i], cov_list=fx["pred_cov"][i], orientation_list=fx["pred_yaw"][i],
                             shape=dict(length=float(fx["pred_shape"][i, 0]), width=float(fx["pred_shape"][i, 1])))
    return preds


def inputs_from_fixture(fx, hull_builder, **override):
    kw = json.loads(str(fx["kw"]))
    cs = CoordinateSystem(fx["ref_xy"])
    preds = predictions_from_fixture(fx)
    N = int(fx["N"])
    weights = {str(n): float(w) for n, w in zip(fx["cost_names"], fx["cost_weight_values"])}
    args = dict(N=N, dt=float(fx["dt"]), low_vel_mode=bool(fx["low_vel_mode"]), x0_lon=fx["x0_lon"],
                x0_lat=fx["x0_lat"], x0_orientation=float(fx["x0_orientation"]), v_des=float(fx["v_des"]),
                vehicle=VehicleParams(), coordinate_system=cs, t_samp=fx["t_order"], v_samp=fx["v_order"],
                d_samp=fx["d_order"], cost_weights=weights, draw_traj_set=bool(kw.get("draw_traj_set", False)),
                kinematic_debug=bool(kw.get("kinematic_debug", False)),
                stop_point=kw.get("stop_point_s") is not None,
                obstacles=pack_predictions(preds, N + 1, hull_builder),
                dto_pos=fx["dto_pos"] if len(fx["dto_pos"]) else None)
    args.update(override)
    inp = PlanInputs(**args)
    inp.predictions = preds
    return inp
