"""stdlib CommonRoad reader: a small authored scenario (always) and the reference's example
scenario when /root/reference is present (build container)."""
import os

import numpy as np
import pytest

from frenetix_motion_planner_amd import commonroad_xml as crx
from frenetix_motion_planner_amd import ref_path

EXAMPLE = "/root/reference/example_scenarios/ZAM_Tjunction-1_42_T-1.xml"


from tests.fixtures import tiny_commonroad_xml  # noqa: E402


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    p = tmp_path_factory.mktemp("cr") / "tiny.xml"
    p.write_text(tiny_commonroad_xml())
    return crx.read_scenario(str(p))


def test_reads_network_obstacles_and_problem(tiny):
    sc = tiny
    assert sc.dt == 0.1 and sc.benchmark_id == "ZAM_Tiny-1_1_T-1"
    assert set(sc.lanelets) == {1, 2, 3} and sc.lanelets[1].successor == [2] and sc.lanelets[2].predecessor == [1]
    assert sc.lanelets[1].adj_left == 3 and sc.lanelets[1].adj_left_same_direction and sc.lanelets[3].adj_right == 1
    assert np.allclose(sc.lanelets[1].center_vertices[:, 1], 0.0) and sc.lanelets[1].lanelet_type == ["urban"]
    ob = sc.obstacles[7]
    assert ob.role == "dynamic" and ob.length == 4.5 and ob.width == 1.9 and len(ob.state_list) == 40
    assert ob.state_at_time(0) is ob.initial_state and ob.state_at_time(1).position[0] == pytest.approx(30.8)
    assert ob.state_at_time(41) is None and sc.obstacles[9].state_at_time(30).position[0] == 80.0
    pp = sc.planning_problems[100]
    assert pp.initial_state.velocity == 9.0 and pp.goals[0].lanelet_ids == [2] and pp.goals[0].time_interval == (50.0, 60.0)
    st = pp.initial_planner_state()
    assert st.velocity == 9.0 and st.orientation == 0.01 and np.allclose(st.position, [5, 0.2])
    assert sc.lanelets_at([5, 0.2]) == [1] and sc.lanelets_at([5, 3.0]) == [3] and sc.lanelets_at([5, 50.0]) == []


def test_route_and_reference_path(tiny):
    sc = tiny
    assert sc.route([1], [2]) == [1, 2] and sc.route([3], [2]) == [3, 1, 2] and sc.route([2], [3]) is None
    ref = sc.route_reference_path(sc.planning_problems[100])
    assert np.allclose(ref[:, 1], 0.0) and ref[0, 0] == 0.0 and ref[-1, 0] == 120.0
    assert np.all(np.diff(ref[:, 0]) > 0)  # the shared vertex of consecutive lanelets appears once
    dense = ref_path.resample_polyline(ref, 0.125)
    prepared = ref_path.prepare_reference_path(dense)
    seg = np.linalg.norm(np.diff(prepared, axis=0), axis=1)
    assert abs(np.median(seg) - 1.0) < 0.02 and prepared[0, 0] < -25 and prepared[-1, 0] > 145  # extended both ends


def test_road_boundary_segments(tiny):
    b = tiny.road_boundary_segments()
    assert b.shape == (24, 4)
    ys = sorted(set(np.round(b[:, 1], 6)))
    assert ys == [-2.0, 2.0, 6.0]                                   # outer borders only; the shared lane line is not one
    on_mid = b[np.isclose(b[:, 1], 2.0)]
    assert on_mid[:, [0, 2]].min() >= 60.0                          # y = 2 is a border only where lanelet 3 has ended
    assert np.isclose(np.linalg.norm(b[:, 2:] - b[:, :2], axis=1).sum(), 120 + 60 + 60)


def test_ground_truth_predictions(tiny):
    """prediction_helpers.py:209-261 including its state_list[ts] velocity indexing."""
    pr = tiny.ground_truth_predictions(time_step=3, pred_horizon=30)
    p7 = pr[7]
    assert p7["pos_list"].shape == (30, 2) and p7["cov_list"].shape == (30, 2, 2)
    assert p7["pos_list"][0, 0] == pytest.approx(30 + 0.8 * 3) and np.allclose(p7["cov_list"][5], np.eye(2) * 0.1)
    assert p7["v_list"][0] == pytest.approx(8.0 + 0.01 * 4)  # state_list[3] is the state of step 4
    assert p7["shape"] == dict(length=4.5, width=1.9)
    late = tiny.ground_truth_predictions(time_step=35, pred_horizon=30)[7]
    assert len(late["pos_list"]) == 5                              # range(35, min(65, 40))
    p9 = pr[9]
    assert p9["pos_list"].shape == (27, 2) and np.all(p9["pos_list"] == [80.0, -1.0])  # range(3, min(33, 30))


@pytest.mark.skipif(not os.path.exists(EXAMPLE), reason="reference example scenarios not present (GPU box)")
def test_example_scenario_of_the_reference():
    sc = crx.read_scenario(EXAMPLE)
    assert sc.benchmark_id == "ZAM_Tjunction-1_42_T-1" and sc.dt == 0.1
    assert len(sc.lanelets) == 12 and len(sc.obstacles) == 5 and list(sc.planning_problems) == [60000]
    pp = sc.planning_problems[60000]
    assert pp.initial_state.velocity == pytest.approx(5.6347706) and pp.goals[0].lanelet_ids == [50203]
    ref = sc.route_reference_path(pp)
    assert sc.lanelets[50203].contains(ref[-1]) or np.linalg.norm(ref[-1] - sc.lanelets[50203].center_vertices[-1]) < 1e-9
    d0 = np.min(np.linalg.norm(ref - pp.initial_state.position, axis=1))
    assert d0 < 3.0
    prepared = ref_path.prepare_reference_path(ref_path.resample_polyline(ref, 0.125))
    from frenetix_motion_planner_amd import CoordinateSystem
    cs = CoordinateSystem(prepared)
    s, d = cs.convert_to_curvilinear_coords(*pp.initial_state.position)
    assert abs(d) < 2.0 and 25 < s < cs.ref_pos[-1] - 30
    preds = sc.ground_truth_predictions(0, 30)
    assert set(preds) == set(sc.obstacles) and all(len(p["pos_list"]) == 30 for p in preds.values())
    border = sc.road_boundary_segments()
    total = sum(len(l.left_vertices) + len(l.right_vertices) - 2 for l in sc.lanelets.values())
    assert 0 < len(border) < total                                  # shared and overlapped bounds are dropped
    # the initial footprint centre is inside the road: no border segment within half a vehicle width of it
    p0 = pp.initial_state.position
    mid = 0.5 * (border[:, :2] + border[:, 2:])
    assert np.min(np.linalg.norm(mid - p0, axis=1)) > 0.9
