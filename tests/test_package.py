"""The winner package (fx_set_package / fx_read_package / fx_plan_and_package, include/fxplan.h): what the planner reads of
the chosen trajectory, gathered by the device behind the selection -- against the classic read-back of the same candidate
and against the oracle; the planner's packaged trajectory pair against planner.py:394-447 computed the long way."""
import numpy as np
import pytest

from frenetix_motion_planner_amd import _abi, synthetic

pytestmark = pytest.mark.gpu


def _engine(**kw):
    from frenetix_motion_planner_amd.engine import FrenetEngine
    return FrenetEngine(**kw)


def _inputs(**kw):
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls
    base = dict(ref_kind="arc", v0=9.0, grid=(5, 9, 11), n_obstacles=3, hull_builder=build_obstacle_hulls)
    base.update(kw)
    return synthetic.make_inputs(**base)


@pytest.mark.parametrize("kw", [dict(), dict(n_obstacles=0), dict(grid=(9, 21, 21), n_obstacles=8, ref_kind="scurve"),
                                dict(lead_gap=18.0), dict(v0=0.5)])
def test_package_equals_read_back_and_oracle(kw):
    from oracle import oracle
    inp = _inputs(**kw)
    with _engine(max_candidates=8192) as eng:
        ref = eng.plan_step(inp)
        res, pkg = eng.plan_step_packaged(inp, yaw_rate0=0.125)
        assert res["best_index"] == ref["best_index"] and res["best_cost"] == ref["best_cost"]
        if ref["best_index"] < 0:
            assert pkg is None
            return
        cand = eng.candidate(ref["best_index"])
        assert pkg.index == ref["best_index"] and pkg.cost == ref["best_cost"] == cand["cost"]
        assert pkg.flags == cand["flags"] and pkg.traj_len == cand["traj_len"]
        assert np.array_equal(pkg.planes, cand["planes"])
        assert np.array_equal(pkg.lon, cand["lon"]) and np.array_equal(pkg.lat, cand["lat"])
        assert np.array_equal(pkg.raw_costs, cand["raw_costs"])
        # derived rows: planner.py:394-447
        th, kap = pkg.planes[2], pkg.planes[5]
        yaw = np.concatenate([[0.125], np.diff(th) / inp.dt])
        assert np.array_equal(pkg.block[_abi.PKG_ROW_YAW_RATE], yaw)
        assert np.allclose(pkg.block[_abi.PKG_ROW_STEERING], np.arctan2(inp.vehicle.wheelbase * kap, 1.0), rtol=0, atol=1e-15)
        orl = pkg.block[_abi.PKG_ROW_ORIENTATION]
        assert np.all(np.abs(orl - inp.x0_orientation) <= np.pi + 1e-12)
        assert np.allclose(np.cos(orl), np.cos(th), atol=1e-12) and np.allclose(np.sin(orl), np.sin(th), atol=1e-12)
    out = oracle.plan_step(inp)
    assert out["result"]["best_index"] == ref["best_index"]
    err = np.abs(out["planes"][ref["best_index"]] - pkg.planes) / (1.0 + np.abs(pkg.planes).max(axis=1, keepdims=True))
    assert err.max() < 1e-9


def test_resident_update_path_equals_fresh_upload():
    """plan_step_packaged rewrites the resident inputs in place when only the state changed: same result as a new engine"""
    a = _inputs(seed=3)
    b = _inputs(seed=4, v0=7.5)           # same structure: other ego state, other predictions
    b.coordinate_system = a.coordinate_system
    b._ref = a._ref
    assert a.structure_key() == b.structure_key()
    with _engine(max_candidates=8192) as eng:
        eng.plan_step_packaged(a)
        res, pkg = eng.plan_step_packaged(b)       # in-place update
        assert eng._resident_key is not None
    with _engine(max_candidates=8192) as eng2:
        ref, rpkg = eng2.plan_step_packaged(b)     # full upload
    assert res["best_index"] == ref["best_index"] and res["best_cost"] == ref["best_cost"]
    for k in ("n_feasible", "n_returned", "n_collisions", "reason_hist"):
        assert res[k] == ref[k]
    if pkg is not None:
        assert np.array_equal(pkg.block, rpkg.block)
    c = _inputs(seed=4, grid=(5, 9, 13))   # other grid: new upload, transparently
    c.coordinate_system, c._ref = a.coordinate_system, a._ref
    assert c.structure_key() != a.structure_key()


def test_batched_packages():
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls
    agents = [synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a, grid=(3, 5, 7), n_obstacles=a % 3,
                                    seed=a) for a in range(4)]
    with _engine(max_candidates=4096, max_agents=4) as eng:
        eng.set_package(True)
        res = eng.plan_batch(agents)
        for a, r in enumerate(res):
            pkg = eng.package(a)
            if r["best_index"] < 0:
                assert pkg is None
                continue
            cand = eng.candidate(r["best_index"], a)
            assert pkg.index == r["best_index"] and np.array_equal(pkg.planes, cand["planes"])
            assert np.array_equal(pkg.raw_costs, cand["raw_costs"]) and pkg.cost == cand["cost"]
        eng.set_package(False)
        eng.evaluate(); eng.finish()
        with pytest.raises(Exception):
            eng.package(0)


def test_planner_pair_from_package_equals_the_long_way():
    from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
    from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = CoordinateSystem(ref)
    s0 = float(cs.ref_pos[40] + 0.1)
    x0 = ReactivePlannerState(0, np.asarray(cs.convert_to_cartesian_coords(s0, 0.2)), float(cs.ref_theta[40]), 10.0, 0.0, 0.07, 0.01)
    preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
    p = ReactivePlannerHip(PlannerConfig(sampling_min=2, sampling_max=3))
    try:
        p.update_externals(reference_path=ref, x_0=x0, desired_velocity=12.0, predictions=preds)
        for step in range(3):   # the second and third step take the in-place update
            pair = p.plan()
            best = p.optimal_trajectory
            assert best._pkg is not None
            best._pkg, keep = None, best._pkg
            long_way = p._compute_trajectory_pair(best)
            best._pkg = keep
            assert len(pair[0]) == len(long_way[0]) == 31
            for i in (0, 1, 7, 30, -1):
                s1, s2 = pair[0][i], long_way[0][i]
                assert s1.time_step == s2.time_step and np.array_equal(s1.position, s2.position)
                assert s1.velocity == s2.velocity and s1.acceleration == s2.acceleration and s1.yaw_rate == s2.yaw_rate
                assert abs(s1.orientation - s2.orientation) < 1e-15 and abs(s1.steering_angle - s2.steering_angle) < 1e-15
                assert pair[1][i]["yaw_rate"] == long_way[1][i]["yaw_rate"] and np.array_equal(pair[1][i]["position"], long_way[1][i]["position"])
            assert pair[2] == long_way[2] and pair[3] == long_way[3]
            t0 = p.x_0.time_step
            assert [s.time_step for s in pair[0][2:5]] == [t0 + 2, t0 + 3, t0 + 4]
            x1 = pair[0][1]
            p.update_externals(x_0=x1, x_cl=(pair[2][1], pair[3][1]), predictions=preds)
    finally:
        p.close()


def test_plan_batch_in_place_update_equals_fresh_upload():
    """plan_batch rewrites the resident agents in place when their structures are unchanged (fx_update_state per agent)"""
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls

    def agents(shift):
        out = []
        for a in range(4):
            inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a + shift, grid=(3, 5, 7),
                                        n_obstacles=1 + a % 3, seed=a + 10 * int(shift > 0), d0=0.1 * a + 0.05 * shift)
            out.append(inp)
        return out

    first, second = agents(0.0), agents(0.7)
    for a in range(4):     # same reference objects -> same structure
        second[a].coordinate_system, second[a]._ref = first[a].coordinate_system, first[a]._ref
    with _engine(max_candidates=4096, max_agents=4) as eng:
        eng.plan_batch(first)
        assert eng._resident_keys is not None
        res = eng.plan_batch(second)           # in place
        assert [i.structure_key() for i in second] == eng._resident_keys
        cands = [eng.candidate(r["best_index"], a) if r["best_index"] >= 0 else None for a, r in enumerate(res)]
    with _engine(max_candidates=4096, max_agents=4) as eng2:
        eng2.upload(second); eng2.evaluate(); ref = eng2.finish()
        for a in range(4):
            for k in ("best_index", "best_cost", "n_feasible", "n_returned", "n_collisions", "reason_hist"):
                assert res[a][k] == ref[a][k], (a, k)
            if cands[a] is not None:
                assert np.array_equal(cands[a]["planes"], eng2.candidate(ref[a]["best_index"], a)["planes"])


def test_plan_batch_packaged_one_call_and_two_halves():
    """fx_plan_batch_packaged (through _fxhost.plan_batch) and its halves fx_plan_batch_begin / fx_plan_batch_end against the piecewise
    form (plan_batch + package per agent): first a fresh upload, then the in-place update of every agent; results as dicts with the
    keys of FxResult.as_dict()."""
    import ctypes as C
    from frenetix_motion_planner_amd import _abi
    from frenetix_motion_planner_amd._lib import lib
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls

    def agents(shift):
        out = []
        for a in range(4):
            out.append(synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a + shift, grid=(3, 5, 7),
                                             n_obstacles=1 + a % 3, seed=a + 10 * int(shift > 0), d0=0.1 * a + 0.05 * shift))
        return out

    first, second = agents(0.0), agents(0.7)
    for a in range(4):
        second[a].coordinate_system, second[a]._ref = first[a].coordinate_system, first[a]._ref
    yaw = [0.01 * a for a in range(4)]
    with _engine(max_candidates=4096, max_agents=4) as ref_eng:
        ref_eng.set_package(True)
        want = []
        for batch in (first, second):
            res = ref_eng.plan_batch(batch)
            want.append((res, [ref_eng.package(a, yaw[a]) for a in range(4)]))

    def same(got, exp):
        (res, pk), (res0, pk0) = got, exp
        for a in range(4):
            assert set(res[a]) == set(res0[a])
            for k in res0[a]:
                if k != "kernel_ms":
                    assert res[a][k] == res0[a][k], (a, k)
            assert (pk[a] is None) == (pk0[a] is None)
            if pk[a] is not None:
                assert pk[a].index == pk0[a].index and pk[a].cost == pk0[a].cost and pk[a].flags == pk0[a].flags
                assert np.array_equal(pk[a].block, pk0[a].block) and np.array_equal(pk[a].raw_costs, pk0[a].raw_costs)
                assert np.array_equal(pk[a].lon, pk0[a].lon) and pk[a].traj_len == pk0[a].traj_len

    with _engine(max_candidates=4096, max_agents=4) as eng:
        same(eng.plan_batch_packaged(first, yaw), want[0])      # upload
        same(eng.plan_batch_packaged(second, yaw), want[1])     # in place
    with _engine(max_candidates=4096, max_agents=4) as eng:
        tok = eng.plan_batch_begin(first)
        same(eng.plan_batch_end(tok, yaw), want[0])
        tok = eng.plan_batch_begin(second)
        assert eng._resident_keys == [i.structure_key() for i in second]
        same(eng.plan_batch_end(tok, yaw), want[1])
        # the C-ABI refuses what does not fit the uploaded batch, and leaves the context usable
        res = (_abi.FxResult * 3)()
        pkg = (_abi.FxPackage * 3)()
        assert lib().fx_plan_batch_packaged(eng._ctx, 3, None, None, res, pkg, None) != 0
        assert b"agents" in lib().fx_last_error()
        assert lib().fx_plan_batch_begin(eng._ctx, 5, None) != 0
        same(eng.plan_batch_packaged(second, yaw), want[1])
    with _engine(max_candidates=4096, max_agents=4) as fresh:
        res = (_abi.FxResult * 4)()
        pkg = (_abi.FxPackage * 4)()
        assert lib().fx_plan_batch_packaged(fresh._ctx, 4, None, None, res, pkg, None) != 0   # nothing uploaded yet


@pytest.mark.parametrize("fused", [True, False])
def test_published_blocks_are_never_torn(fused):
    """The result block and the winner package reach pinned host memory through system-scope stores that every wave drains
    (`s_waitcnt vmcnt(0)`) before the sequence word is sent -- no L2 write-back fence (fx_tail.h).  4 000 steps alternating between
    two agents' inputs with different winners: whenever the host sees a step's sequence word, every word of THAT step's result and
    package is there (a torn publication would pair one step's winner with the other's arrays)."""
    a = _inputs(seed=3)
    b = _inputs(seed=4, v0=7.5)           # same structure: other ego state, other predictions
    b.coordinate_system = a.coordinate_system
    with _engine(max_candidates=4096) as eng:
        eng.set_fused_selection(fused)
        want = []
        for inp in (a, b):
            res, pkg = eng.plan_step_packaged(inp, yaw_rate0=0.0)
            assert pkg is not None
            want.append((res, pkg.index, pkg.cost, pkg.block.copy(), pkg.lon.copy(), pkg.raw_costs.copy()))
        assert want[0][1] != want[1][1]
        for k in range(4000):
            inp, (res0, idx, cost, block, lon, raw) = (a, want[0]) if k % 2 == 0 else (b, want[1])
            res, pkg = eng.plan_step_packaged(inp, yaw_rate0=0.0)
            assert res["best_index"] == idx and res["best_cost"] == cost and res["n_collisions"] == res0["n_collisions"], k
            assert res["n_feasible"] == res0["n_feasible"] and res["reason_hist"] == res0["reason_hist"], k
            assert pkg.index == idx and pkg.cost == cost, k
            assert np.array_equal(pkg.block, block) and np.array_equal(pkg.lon, lon) and np.array_equal(pkg.raw_costs, raw), k


def test_state_updates_written_by_the_host_equal_the_staging_kernel(monkeypatch):
    """OPT-IN path (FX_STAGE=bar): where the device memory is mapped into the process (large BAR) a state update is written by the
    host straight into the device arena -- posted writes + HDP flush in front of the evaluation launch, no staging kernel (fx_api.hip,
    probe_host_writes / host_stage; refused at fx_create where the mapping, the HDP flush register or the probe's kernel read-back
    is missing).  The same alternating inputs through a context that stages with the kernel (FX_STAGE=kernel), through the default
    one (stream-ordered staging: kernel or DMA) and through the opt-in one: every step's result and package are identical, at a
    planner-sized step (5 obstacles) and at config 3's size (20 obstacles: 140 KB per update) -- every step rewrites lines the
    previous step's kernels have read."""
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd._lib import FxError
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls
    cases = []
    a = _inputs(seed=3); b = _inputs(seed=4, v0=7.5); b.coordinate_system = a.coordinate_system
    cases.append((a, b, 300))
    kw = dict(ref_kind="arc", grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, hull_builder=build_obstacle_hulls)
    c = synthetic.make_inputs(v0=10.0, seed=11, **kw); d = synthetic.make_inputs(v0=9.0, seed=12, **kw); d.coordinate_system = c.coordinate_system
    cases.append((c, d, 60))
    for x, y, n in cases:
        monkeypatch.setenv("FX_STAGE", "kernel")
        ek = _engine(max_candidates=x.n_candidates + 64)
        monkeypatch.setenv("FX_STAGE", "bar")
        try:
            eb = _engine(max_candidates=x.n_candidates + 64)
        except FxError:
            eb = None   # (no large BAR / no HDP flush register on this box: the opt-in is refused loudly, nothing to compare)
        monkeypatch.delenv("FX_STAGE")
        ea = _engine(max_candidates=x.n_candidates + 64)
        try:
            import time
            paths = set()
            for k in range(n):
                inp = x if k % 2 == 0 else y
                rk, pk = ek.plan_step_packaged(inp, yaw_rate0=0.0)
                for e in (ea, eb):
                    if e is None:
                        continue
                    if e is eb and k % 3 == 0:
                        time.sleep(2e-4)   # host writes need the context's stream IDLE (hipStreamQuery), else the staging kernel runs: both happen
                    ra, pa = e.plan_step_packaged(inp, yaw_rate0=0.0)
                    if e is eb:
                        paths.add(e.step_info()["staging"])
                    assert rk == ra, k
                    assert (pk is None) == (pa is None)
                    if pk is not None:
                        assert pk.index == pa.index and np.array_equal(pk.block, pa.block) and np.array_equal(pk.raw_costs, pa.raw_costs), k
            assert ek.step_info()["staging"] == "kernel"
            assert ea.step_info()["staging"] in ("kernel", "dma")      # the default is stream-ordered
            cost_k, flags_k = ek.costs()
            for e in (ea, eb):
                if e is None:
                    continue
                cost_a, flags_a = e.costs()
                assert np.array_equal(flags_k, flags_a) and np.array_equal(cost_k, cost_a)
            if eb is not None:
                assert "host_writes" in paths and paths <= {"host_writes", "kernel"}, paths
        finally:
            ek.close(); ea.close()
            if eb is not None:
                eb.close()
