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
