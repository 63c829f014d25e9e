"""Reference-path preparation against vectors produced by the reference's own
utils_coordinate_system.py functions (tests/golden/gen_refpath_golden.py)."""
import os

import numpy as np
import pytest

from frenetix_motion_planner_amd import CoordinateSystem, ref_path

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "refpath_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})


def close(a, b, tol):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.max(np.abs(a - b)) <= tol, np.max(np.abs(a - b))


@pytest.mark.parametrize("name", NAMES)
def test_extensions_match_reference(name):
    pl = GOLD[f"{name}/in"]
    close(ref_path.extend_ref_path_both_ends(pl), GOLD[f"{name}/extend_both_30"], 1e-12)
    close(ref_path.extend_path_linearly(pl, 50, at_start=True), GOLD[f"{name}/extend_start_50"], 1e-12)
    close(ref_path.extend_points(pl), GOLD[f"{name}/extend_points"], 1e-12)
    close(ref_path.extend_ref_path(pl, pl[0] + np.array([-0.3, 0.1])), GOLD[f"{name}/extend_ref_path_first"], 1e-12)
    close(ref_path.extend_ref_path(pl, pl[len(pl) // 2]), GOLD[f"{name}/extend_ref_path_mid"], 0.0)


@pytest.mark.parametrize("name", NAMES)
def test_end_extension_extrapolation_and_preprocessing_match_reference(name):
    """extend_points_end (:80-99), extrapolate_ref_path (:158-170), preprocess_ref_path (:173-184)"""
    pl = GOLD[f"{name}/in"]
    close(np.asarray(ref_path.extend_points_end(pl)), GOLD[f"{name}/extend_points_end_30"], 1e-12)
    close(ref_path.extrapolate_ref_path(pl), GOLD[f"{name}/extrapolate"], 1e-9)
    if f"{name}/preprocessed" in GOLD.files:
        out = ref_path.preprocess_ref_path(GOLD[f"{name}/bent_in"])
        close(out, GOLD[f"{name}/preprocessed"], 1e-9)
        from frenetix_motion_planner_amd.coordinate_system import compute_curvature_from_polyline
        assert max(compute_curvature_from_polyline(out)) <= 0.1


def test_resample_against_hand_computed_vectors():
    """resample_polyline stands in for commonroad_dc's helper (unpinned): vectors worked out by hand, independent of the
    implementation -- walk the path, drop a vertex every `step` metres of arc length, close with the last vertex."""
    # 3 m east, then 4 m north (7 m): samples at arc length 0, 2, 4, 6 and the end
    L = np.array([[0.0, 0.0], [3.0, 0.0], [3.0, 4.0]])
    assert np.allclose(ref_path.resample_polyline(L, 2.0), [[0, 0], [2, 0], [3, 1], [3, 3], [3, 4]], atol=1e-12)
    # 2.5 m straight, step 1: 0, 1, 2 and the end vertex at 2.5
    assert np.allclose(ref_path.resample_polyline(np.array([[0.0, 0.0], [2.5, 0.0]]), 1.0), [[0, 0], [1, 0], [2, 0], [2.5, 0]], atol=1e-12)
    # the last sample lands exactly on the end: no duplicate
    assert np.allclose(ref_path.resample_polyline(np.array([[0.0, 0.0], [4.0, 0.0]]), 2.0), [[0, 0], [2, 0], [4, 0]], atol=1e-12)
    # a 3-4-5 diagonal followed by a short leg: arc 0 .. 5 on the diagonal (unit vector (0.6, 0.8)), 5 .. 6.5 east
    D = np.array([[0.0, 0.0], [3.0, 4.0], [4.5, 4.0]])
    want = [[0, 0], [1.2, 1.6], [2.4, 3.2], [4.0, 4.0], [4.5, 4.0]]       # arc 0, 2, 4, 6 (= 1 m past the corner), end
    assert np.allclose(ref_path.resample_polyline(D, 2.0), want, atol=1e-12)
    # step longer than the path: just the two ends
    assert np.allclose(ref_path.resample_polyline(np.array([[1.0, 1.0], [1.0, 2.0]]), 5.0), [[1, 1], [1, 2]], atol=1e-12)


def test_chaikin_keeps_the_ends_and_cuts_the_corner():
    sq = np.array([[0.0, 0.0], [4.0, 0.0], [4.0, 4.0]])
    assert np.allclose(ref_path.chaikins_corner_cutting(sq), [[0, 0], [1, 0], [3, 0], [4, 1], [4, 3], [4, 4]])
    assert len(ref_path.chaikins_corner_cutting(sq, 3)) == 24   # n vertices -> 2 n per refinement


@pytest.mark.parametrize("name", [n for n in NAMES if f"{n}/smooth" in GOLD.files])
def test_smoothing_matches_reference(name):
    pl = GOLD[f"{name}/in"]
    close(ref_path.smooth_ref_path(pl), GOLD[f"{name}/smooth"], 1e-9)
    close(ref_path.smooth_ref_path(pl, 8), GOLD[f"{name}/smooth_8"], 1e-9)
    prepared = ref_path.prepare_reference_path(pl)
    close(prepared, GOLD[f"{name}/prepared"], 1e-9)
    # what the planner needs from it: ~1 m spacing, no duplicate vertices, a usable coordinate system
    seg = np.linalg.norm(np.diff(prepared, axis=0), axis=1)
    assert seg.min() > 1e-6 and abs(np.median(seg) - 1.0) < 0.02
    cs = CoordinateSystem(prepared)
    assert np.all(np.diff(cs.ref_pos) > 0) and np.max(np.abs(cs.ref_curv)) < 0.2
    s, d = cs.convert_to_curvilinear_coords(*pl[len(pl) // 2])
    assert abs(d) < 0.2
    xy = cs.convert_to_cartesian_coords(s, d)
    assert np.linalg.norm(xy - pl[len(pl) // 2]) < 1e-6


def test_resample_properties():
    """The third-party helper (restated, unpinned): defining properties."""
    pl = GOLD["turn_left/in"]
    r = ref_path.resample_polyline(pl, 1.0)
    assert np.array_equal(r[0], pl[0]) and np.allclose(r[-1], pl[-1])
    seg = np.linalg.norm(np.diff(r, axis=0), axis=1)
    assert np.all(seg[:-1] <= 1.0 + 1e-9) and np.all(seg[:-1] > 0.99)  # chord <= arc step; last one is the remainder
    assert len(ref_path.resample_polyline(pl[:1], 1.0)) == 1
    # a sample that falls exactly on a vertex belongs to the segment that starts there; an integer step is fine
    sq = np.array([[0.0, 0.0], [2.0, 0.0], [2.0, 2.0]])
    assert np.allclose(ref_path.resample_polyline(sq, 1), [[0, 0], [1, 0], [2, 0], [2, 1], [2, 2]])
    # duplicate vertices are stepped over
    dup = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 0.0], [3.0, 0.0]])
    assert np.allclose(ref_path.resample_polyline(dup, 1.0), [[0, 0], [1, 0], [2, 0], [3, 0]])


def test_degenerate_extensions():
    p = np.array([[0.0, 0.0], [0.0, 0.0], [1.0, 0.0]])
    assert ref_path.extend_path_linearly(p, 10, at_start=True) is p  # coincident first points: unchanged
    far = np.array([[0.0, 0.0], [100.0, 0.0]])
    assert np.array_equal(ref_path.extend_path_linearly(far, 30, at_start=False), far)  # segment longer than the extension
