"""Road-boundary stage: the footprint-vs-boundary test of DESIGN.md 4.3.
CPU: oracle behaviour and the host-side packing.  GPU: HIP path == oracle."""
import numpy as np
import pytest

from frenetix_motion_planner_amd import _abi, synthetic
from frenetix_motion_planner_amd.problem import pack_road_boundary

CASES = {
    "arc_narrow": dict(ref_kind="arc", v0=10.0, grid=(7, 11, 13), road_half_width=2.6),
    "arc_narrow_debug_obs": dict(ref_kind="arc", v0=10.0, grid=(7, 11, 13), road_half_width=2.4, n_obstacles=4,
                                 draw_traj_set=True, kinematic_debug=True),
    "scurve_tight": dict(ref_kind="scurve", kappa=0.03, v0=8.0, grid=(6, 9, 17), road_half_width=2.2, seed=3),
    "lowvel_all_off": dict(ref_kind="arc", v0=1.5, v_des=3.0, grid=(5, 9, 11), road_half_width=0.95),
    "stop_narrow": dict(ref_kind="arc", v0=3.0, d0=1.3, grid=(7, 11, 13), stop_point_s=10.0, v_des=0.0, road_half_width=2.4,
                        draw_traj_set=True, kinematic_debug=True),
    "offset_start": dict(ref_kind="arc", v0=9.0, d0=1.2, grid=(6, 9, 13), road_half_width=2.6),
}


def oracle_inputs(kw):
    from oracle import oracle
    return synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_boundary_semantics(name):
    from oracle import oracle
    inp = oracle_inputs(CASES[name])
    assert inp.mode & _abi.FX_MODE_ROAD_BOUNDARY
    out = oracle.plan_step(inp)
    free = oracle.plan_step(oracle_inputs({k: v for k, v in CASES[name].items() if k != "road_half_width"}))
    b, step = out["boundary"], out["boundary_step"]
    assert np.array_equal(b, step >= 0) and b.sum() > 0
    if name == "lowvel_all_off":  # the ego itself (d0 = 0.2, half width 0.8) already overlaps a 0.95 m half-width road
        assert np.array_equal(b, out["selectable"]) and out["result"]["best_index"] == -1 and np.all(step[b] == 0)
    else:
        assert (~b & out["selectable"]).sum() > 0
    assert np.all(out["selectable"][b])                      # only walked candidates carry the flag
    # everything but the boundary bit and the winner is unchanged by the stage
    assert np.array_equal(out["flags"] & ~np.uint32(_abi.FX_FLAG_BOUNDARY), free["flags"])
    assert np.array_equal(out["cost"], free["cost"])
    best = out["result"]["best_index"]
    if best >= 0:
        assert not b[best] and not out["collision"][best]
        # first selectable, collision-free, in-road candidate of the cost order (planner.py:336-390)
        ok = out["selectable"] & ~out["collision"] & ~b
        ids = np.nonzero(ok)[0]
        assert best == ids[np.lexsort((ids, out["cost"][ids]))][0]
    # wide lateral end states leave a 2.x m half-width road; the centre ones stay inside
    d_end = inp.d_samp[np.arange(inp.n_candidates) % len(inp.d_samp)]
    sel = out["selectable"]
    if name != "lowvel_all_off":
        assert b[sel & (np.abs(d_end) >= 2.9)].all()
        assert not b[sel & (np.abs(d_end) < 0.3) & (np.abs(inp.x0_lat[0]) < 0.5)].any()


def test_packing_bins_cover_every_contact():
    """Brute force over all pieces (the oracle) never finds a contact outside the bin of the step's reference segment."""
    from oracle import oracle
    inp = oracle_inputs(CASES["scurve_tight"])
    bd = inp._bound
    out = oracle.plan_step(inp)
    veh, cs = inp.vehicle, inp.coordinate_system
    hl, hw = veh.length / 2, veh.width / 2
    hits = np.nonzero(out["boundary"])[0][:40]
    for g in hits:
        i = out["boundary_step"][g]
        x, y, th, s = (out["planes"][g][k][i] for k in (0, 1, 2, 7))
        k = cs.segment_of(s)
        u = np.array([np.cos(th), np.sin(th)])
        c = np.array([x, y]) + veh.wb_rear_axle * u
        n = np.array([-u[1], u[0]])
        touching = []
        for j, q in enumerate(bd["piece"]):
            e, h = q[:2] - c, q[2:]
            ex, ey, hx, hy = e @ u, e @ n, h @ u, h @ n
            sep = abs(ex) > hl + abs(hx) or abs(ey) > hw + abs(hy) or abs(ex * hy - ey * hx) > hl * abs(hy) + hw * abs(hx)
            if not sep:
                touching.append(j)
        assert touching, (g, i)
        in_bin = set(bd["item"][bd["bin"][k]:bd["bin"][k + 1]].tolist())
        assert set(touching) <= in_bin


def test_pack_road_boundary_layout():
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(3, 5, 5), road_half_width=3.0)
    bd = inp._bound
    cs = inp.coordinate_system
    assert bd["piece"].shape[1] == 4 and len(bd["bin"]) == len(cs.reference) + 1 and bd["bin"][-1] == len(bd["item"])
    assert np.all(np.linalg.norm(bd["piece"][:, 2:], axis=1) <= 2.0 + 1e-9)          # max_len 4 m -> half <= 2 m
    assert np.all(np.diff(bd["bin"]) >= 0) and bd["item"].min() >= 0 and bd["item"].max() < bd["n"]
    st = inp.as_struct()
    assert st.n_bound == bd["n"] and st.bound_d_reach == bd["d_reach"] and st.mode & _abi.FX_MODE_ROAD_BOUNDARY
    # long segments are split, pieces tile them
    one = pack_road_boundary(np.array([[0.0, 5.0, 10.0, 5.0]]), cs, inp.vehicle, 5.0, max_len=4.0)
    assert one["n"] == 3 and np.allclose(one["piece"][:, 0], [10 / 6, 5.0, 50 / 6]) and np.allclose(one["piece"][:, 2], 10 / 6)
    other = synthetic.make_inputs(ref_kind="arc", n_knots=200, v0=10.0, grid=(3, 5, 5))
    with pytest.raises(ValueError):
        type(inp)(**{**{f: getattr(other, f) for f in ("N", "dt", "low_vel_mode", "x0_lon", "x0_lat", "x0_orientation", "v_des",
                                                       "vehicle", "coordinate_system", "t_samp", "v_samp", "d_samp")},
                     "road_boundary": bd}).as_struct()


# ------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def eng():
    from frenetix_motion_planner_amd.engine import FrenetEngine
    e = FrenetEngine(max_candidates=4096, max_steps=50, max_ref_knots=1024, max_obstacles=32, max_pred_steps=64, max_agents=4)
    yield e
    e.close()


def hip_inputs(kw):
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls
    return synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,mapping", [(0, 0), (1, 0), (2, 1), (2, 2), (4, 2), (8, 1)])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_boundary_matches_oracle(eng, name, lanes, mapping):
    from oracle import oracle
    out = oracle.plan_step(oracle_inputs(CASES[name]))
    inp = hip_inputs(CASES[name])
    eng.set_tuning(lanes, 0, 0, 0, mapping)
    try:
        res = eng.plan_step(inp)
        cost, flags = eng.costs()
        steps = eng.boundary_steps()
    finally:
        eng.set_tuning(0, 0, 0)
    robust = out["margin"] >= 1e-9
    assert robust.mean() > 0.8
    assert np.array_equal(flags[robust], out["flags"][robust])
    assert np.array_equal(steps[robust], out["boundary_step"][robust])
    assert res["best_index"] == out["result"]["best_index"] and res["n_collisions"] == out["result"]["n_collisions"]
    tc, ti = eng.topk(8)
    ok = out["selectable"] & ~out["collision"] & ~out["boundary"]
    ids = np.nonzero(ok)[0]
    want = ids[np.lexsort((ids, out["cost"][ids]))][:8]
    if robust.all():
        assert np.array_equal(ti[0][:len(want)], want)


@pytest.mark.gpu
def test_generic_kernel_and_batch_with_boundary(eng):
    inps = [hip_inputs(CASES[n]) for n in ("arc_narrow", "scurve_tight", "stop_narrow")]
    inps.append(hip_inputs(dict(ref_kind="arc", v0=10.0, grid=(5, 7, 9))))  # an agent without a boundary in the same launch
    singles = []
    for variant in (1, 2):
        eng.set_tuning(0, 0, variant)
        try:
            per = []
            for inp in inps[:3]:
                r = eng.plan_step(inp)
                per.append((r, eng.costs()[1].copy(), eng.boundary_steps().copy()))
            singles.append(per)
        finally:
            eng.set_tuning(0, 0, 0)
    for a, b in zip(*singles):  # generic == grid, bitwise
        assert a[0]["best_index"] == b[0]["best_index"] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    batch = eng.plan_batch(inps)
    for k, (r, f, st) in enumerate(singles[1]):
        assert batch[k]["best_index"] == r["best_index"]
        assert np.array_equal(eng.costs(k)[1], f) and np.array_equal(eng.boundary_steps(k), st)
    with pytest.raises(Exception):
        eng.boundary_steps(3)


@pytest.mark.gpu
def test_library_bins_equal_package_bins():
    from frenetix_motion_planner_amd.engine import build_boundary_bins
    inp = synthetic.make_inputs(ref_kind="scurve", kappa=0.03, v0=8.0, grid=(3, 5, 5), road_half_width=2.2)
    bd = inp._bound
    piece, bins, item = build_boundary_bins(inp.coordinate_system.reference, inp.road_boundary, 4.0, bd["reach"])
    assert np.allclose(piece, bd["piece"], atol=1e-12) and np.array_equal(bins, bd["bin"]) and np.array_equal(item, bd["item"])
