"""Holds the oracle to the pins tests/golden/pin_third_party.py takes from commonroad_dc (CCosy projection, pycrcc collision
verdicts) -- when that file exists.  In this container and on the GPU boxes commonroad_dc is not installable, the pins file is
absent and these tests skip: the three third-party pieces stay "parity unpinned" (DESIGN.md 4)."""
import os

import numpy as np
import pytest

from tests.fixtures import GOLDEN_DIR, golden_names, inputs_from_fixture, load_golden

PINS = os.path.join(GOLDEN_DIR, "third_party_pins.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PINS), reason="tests/golden/third_party_pins.npz absent: commonroad_dc was never "
                                "importable where tests/golden/pin_third_party.py --write ran")


def test_projection_reading_matches_ccosy():
    from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
    z = np.load(PINS)
    pn, bis = (int(v) for v in z["proj/reading"])
    assert float(z["proj/max_err"][0]) < 1e-6, "no implemented reading of the projection matches CCosy to the north star's 1e-6 m"
    for name in golden_names():
        if f"proj/{name}/xy" not in z.files:
            continue
        cs = CoordinateSystem(load_golden(name)["ref_xy"], pseudo_normal=bool(pn), vertex_tangent="bisector" if bis else "chord")
        mine = np.asarray([cs.convert_to_cartesian_coords(float(s), float(d)) for s, d in zip(z[f"proj/{name}/s"], z[f"proj/{name}/d"])])
        assert np.abs(mine - z[f"proj/{name}/xy"]).max() < 1e-6
    # the package default must be the matching reading
    assert (pn, bis) == (0, 0), "make the matching reading the default of CoordinateSystem (and regenerate the golden x / y)"


def test_collision_verdicts_match_pycrcc():
    from oracle import oracle
    z = np.load(PINS)
    for name in golden_names():
        if f"coll/{name}/ids" not in z.files:
            continue
        res = oracle.plan_step(inputs_from_fixture(load_golden(name), oracle.build_obstacle_hulls))
        ids, verdict = z[f"coll/{name}/ids"], z[f"coll/{name}/pycrcc"].astype(bool)
        robust = res["margin"][ids] >= 1e-9
        assert np.array_equal(res["collision"][ids][robust], verdict[robust]), name
