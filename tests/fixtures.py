"""Rebuild PlanInputs from a golden fixture (tests/golden/*.npz)."""
import glob
import json
import os

import numpy as np

from frenetix_motion_planner_amd import CoordinateSystem, PlanInputs, VehicleParams, pack_predictions

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    """plan-step scenarios (INDEX.json of gen_golden.py); other fixtures in the directory have their own tests"""
    import json
    return sorted(json.load(open(os.path.join(GOLDEN_DIR, "INDEX.json"))))


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


def scenario_inputs(kw):
    """PlanInputs of the first plan step of a scenario's planning problem, built by this package's own host side: stdlib XML
    reader fixture -> route polyline -> prepare_reference_path -> initial Frenet state -> velocity planner -> sampling ranges ->
    ground-truth predictions (frenet_interface.py:33-205).  No GPU involved (the hull builder and the projection are host C).
    kw: scenario (fixture name under tests/golden), planning_problem, level, draw_traj_set, kinematic_debug."""
    from frenetix_motion_planner_amd import commonroad_xml as crx
    from frenetix_motion_planner_amd.frenet_interface import FrenetPlannerInterfaceHip
    from frenetix_motion_planner_amd.reactive_planner import PlannerConfig
    sc = crx.read_scenario_json(os.path.join(GOLDEN_DIR, kw["scenario"] + ".scenario.json"))
    cfg = PlannerConfig()
    cfg.draw_traj_set = bool(kw.get("draw_traj_set", False))
    cfg.kinematic_debug = bool(kw.get("kinematic_debug", False))
    cfg.sampling_min, cfg.sampling_max = kw["level"], kw["level"] + 1
    itf = FrenetPlannerInterfaceHip(kw["planning_problem"], sc, sc.planning_problems[kw["planning_problem"]], config=cfg)
    preds = sc.ground_truth_predictions(0, itf.planner.N)
    itf.update_planner(None, preds)
    inp = itf.begin_step()
    inp.predictions = preds
    inp.x0_velocity = float(itf.x_0.velocity)   # the v-range is built from the Cartesian speed (planner.py:304-306)
    return inp


import textwrap  # noqa: E402


def _pts(pts):
    return "".join(f"<point><x>{x}</x><y>{y}</y></point>" for x, y in pts)


def _state(x, y, th, t, v, extra=""):
    return (f"<position><point><x>{x}</x><y>{y}</y></point></position><orientation><exact>{th}</exact></orientation>"
            f"<time><exact>{t}</exact></time><velocity><exact>{v}</exact></velocity>{extra}")


def tiny_commonroad_xml(lead_x: float = 30.0) -> str:
    """A small authored CommonRoad 2020a scenario: two consecutive lanelets + a left neighbour, one dynamic and one
    static obstacle, one planning problem whose goal is lanelet 2."""
    xs = np.arange(0, 60.1, 10.0)
    l1 = _pts([(x, 2.0) for x in xs]), _pts([(x, -2.0) for x in xs])
    xs2 = np.arange(60, 120.1, 10.0)
    l2 = _pts([(x, 2.0) for x in xs2]), _pts([(x, -2.0) for x in xs2])
    l3 = _pts([(x, 6.0) for x in xs]), _pts([(x, 2.0) for x in xs])
    traj = "".join(f"<state>{_state(lead_x + 0.8 * k, 0.1, 0.0, k, 8.0 + 0.01 * k)}</state>" for k in range(1, 41))
    xml = textwrap.dedent(f"""\
        <?xml version='1.0' encoding='UTF-8'?>
        <commonRoad timeStepSize="0.1" commonRoadVersion="2020a" benchmarkID="ZAM_Tiny-1_1_T-1">
          <lanelet id="1"><leftBound>{l1[0]}</leftBound><rightBound>{l1[1]}</rightBound><successor ref="2"/>
            <adjacentLeft ref="3" drivingDir="same"/><laneletType>urban</laneletType></lanelet>
          <lanelet id="2"><leftBound>{l2[0]}</leftBound><rightBound>{l2[1]}</rightBound><predecessor ref="1"/></lanelet>
          <lanelet id="3"><leftBound>{l3[0]}</leftBound><rightBound>{l3[1]}</rightBound><adjacentRight ref="1" drivingDir="same"/></lanelet>
          <dynamicObstacle id="7"><type>car</type><shape><rectangle><length>4.5</length><width>1.9</width></rectangle></shape>
            <initialState>{_state(lead_x, 0.1, 0.0, 0, 8.0, "<acceleration><exact>0.0</exact></acceleration>")}</initialState>
            <trajectory>{traj}</trajectory></dynamicObstacle>
          <staticObstacle id="9"><type>parkedVehicle</type><shape><rectangle><length>4.0</length><width>1.8</width></rectangle></shape>
            <initialState>{_state(80, -1.0, 0.0, 0, 0.0)}</initialState></staticObstacle>
          <planningProblem id="100"><initialState>{_state(5, 0.2, 0.01, 0, 9.0, "<yawRate><exact>0.0</exact></yawRate><slipAngle><exact>0.0</exact></slipAngle>")}</initialState>
            <goalState><position><lanelet ref="2"/></position><time><intervalStart>50</intervalStart><intervalEnd>60</intervalEnd></time>
              <velocity><intervalStart>0</intervalStart><intervalEnd>12</intervalEnd></velocity></goalState></planningProblem>
        </commonRoad>""")
    return xml


