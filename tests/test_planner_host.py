"""Planner front-end logic that needs no GPU: driven through the oracle-backed stand-in engine (tests/oracle_engine.py)."""
from itertools import product

import numpy as np
import pytest

from frenetix_motion_planner_amd import VehicleParams, _abi, synthetic
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig, ReactivePlannerHip, ReactivePlannerState
from tests.oracle_engine import OracleEngine


def blocked_planner(engine="oracle", **cfg):
    """An ego whose every candidate collides: one wide obstacle parked across the lane right in front of it.
    engine: "oracle" (CPU stand-in) or None (the planner creates its FrenetEngine: GPU tests)."""
    rp = ReactivePlannerHip(PlannerConfig(**cfg), VehicleParams(), engine=OracleEngine() if engine == "oracle" else engine)
    ref = synthetic.reference_polyline("straight", 400, 0.5)
    x0 = ReactivePlannerState(time_step=0, position=np.array([20.0, 0.2]), orientation=0.0, velocity=8.0)
    n = 31
    wall = dict(pos_list=np.tile([[27.0, 0.0]], (n, 1)), cov_list=np.tile(np.eye(2) * 0.1, (n, 1, 1)),
                orientation_list=np.full(n, np.pi / 2), shape=dict(length=14.0, width=3.0))
    rp.update_externals(reference_path=ref, x_0=x0, desired_velocity=8.0, predictions={5: wall})
    return rp


def test_emergency_stopping_selection():
    """reactive_planner_cpp.py:404-413,443-466: nothing collision-free -> first existing feasible combination of
    product(unique v, unique t, d sorted by |d - d_pos|)."""
    rp = blocked_planner(emergency_selection=True)
    pair = rp.plan()
    step = rp.last_step
    assert step.result["best_index"] == -1 and step.result["n_feasible"] > 0
    assert step.result["n_collisions"] == step.result["n_feasible"]
    best = rp.optimal_trajectory
    assert pair is not None and best is not None and best.feasible
    # the reference's selection, restated over TrajectorySample views
    inp = step.inputs
    feas = [step.sample(int(g)) for g in np.nonzero(step.mask(_abi.FX_FLAG_FEASIBLE) & step.mask(_abi.FX_FLAG_VALID) &
                                                    step.mask(_abi.FX_FLAG_RETURNED))[0]]
    table = {}
    for tr in feas:
        sp = tr.sampling_parameters
        table.setdefault(sp[5], {}).setdefault(sp[1], {})[sp[10]] = tr
    v_list, t_list = np.unique(inp.v_samp), np.unique(inp.t_samp)
    d_list = np.unique(inp.d_samp)
    d_list = d_list[np.argsort(np.abs(d_list - rp.x_cl[1][0]), kind="stable")]
    want = next(table[v][t][d] for v, t, d in product(v_list, t_list, d_list) if v in table and t in table[v] and d in table[v][t])
    assert best.uniqueId == want.uniqueId
    assert best.sampling_parameters[5] == min(tr.sampling_parameters[5] for tr in feas)
    # the Python back-end's behaviour (no emergency selection): no trajectory
    rp2 = blocked_planner()
    assert rp2.plan() is None and rp2.optimal_trajectory is None


def test_last_level_fallback_selector_min_risk():
    """reactive_planner.py:262-269: at the LAST sampling level, when every feasible trajectory collides, the Python back-end
    returns sorted(feasible, key=ego risk + obstacle risk)[0].  The hook gets the feasible trajectories in creation order; with a
    toy risk function (|lateral end offset| + end velocity / 100) the choice equals the restated rule, ties go to the first."""
    calls = []

    def risk(tr):
        sp = tr.sampling_parameters
        return round(abs(sp[10]), 3) + sp[5] / 100.0

    rp = blocked_planner(sampling_min=1, sampling_max=3)   # levels 1 and 2: the hook may only fire at level 2
    sel = ReactivePlannerHip.min_risk_selector(risk)

    def selector(feasible):
        calls.append((rp.last_step.inputs.n_candidates, len(feasible)))
        return sel(feasible)

    rp.set_fallback_selector(selector)
    pair = rp.plan()
    step = rp.last_step
    assert len(calls) == 1 and calls[0][0] == step.inputs.n_candidates   # once, on the last level's step
    assert step.result["best_index"] == -1 and step.result["n_collisions"] == step.result["n_feasible"] == calls[0][1] > 0
    feas = [step.sample(int(g)) for g in np.nonzero(step.mask(_abi.FX_FLAG_FEASIBLE) & step.mask(_abi.FX_FLAG_VALID) &
                                                    step.mask(_abi.FX_FLAG_RETURNED))[0]]
    want = sorted(feas, key=risk)[0]
    best = rp.optimal_trajectory
    assert pair is not None and best is not None and best.uniqueId == want.uniqueId and best.feasible
    assert len(pair[0]) == rp.N + 1
    # the batched path's phases take the same decision (single-level planner: the first level is the last)
    rp1 = blocked_planner(sampling_min=2, sampling_max=3)
    rp1.set_fallback_selector(ReactivePlannerHip.min_risk_selector(risk))
    inp = rp1.plan_begin()
    res = rp1.engine.plan_batch([inp])[0]
    best1 = rp1.plan_consume(inp, res, rp1.engine, 0)
    assert best1 is not None and best1.uniqueId == want.uniqueId
    # without a selector (and without the C++ back-end's emergency selection) the step has no trajectory
    assert blocked_planner(sampling_min=1, sampling_max=3).plan() is None


def test_plan_phases_equal_plan():
    """plan_begin / plan_consume / plan_finish (what AgentBatchHip drives) == plan()."""
    import time
    kw = dict(ref_kind="arc", v0=9.0)
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = synthetic.CoordinateSystem(ref)
    xy = cs.convert_to_cartesian_coords(float(cs.ref_pos[40]) + 0.1, 0.2)
    x0 = ReactivePlannerState(time_step=0, position=xy, orientation=float(cs.ref_theta[40]), velocity=9.0)
    preds = synthetic.synthetic_predictions(cs, 4, 30, 0.1, float(cs.ref_pos[40]), np.random.default_rng(3))
    outs = []
    for phased in (False, True):
        eng = OracleEngine()
        rp = ReactivePlannerHip(PlannerConfig(), VehicleParams(), engine=eng)
        rp.update_externals(reference_path=ref, x_0=x0, desired_velocity=11.0, predictions=preds)
        if phased:
            inp = rp.plan_begin()
            res = eng.plan_batch([inp])
            best = rp.plan_consume(inp, res[0], eng, 0)
            pair = rp.plan_finish(best, time.time())
        else:
            pair = rp.plan()
        outs.append((rp.optimal_trajectory.uniqueId, np.array(pair[2]), rp.infeasible_count_collision,
                     list(rp._infeasible_count_kinematics)))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2:] == outs[1][2:]


def test_planner_logging_hook(tmp_path):
    """planner.py:637-649: plan() feeds the logger -- logs.csv gets one line per plan step, predictions.csv likewise,
    trajectories.csv / trajectories.db every returned trajectory when save_all_traj is on."""
    import sqlite3
    from frenetix_motion_planner_amd.logging_formats import DataLoggingCosts
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = synthetic.CoordinateSystem(ref)
    s0 = float(cs.ref_pos[40]) + 0.1
    x0 = ReactivePlannerState(time_step=7, position=cs.convert_to_cartesian_coords(s0, 0.2), orientation=float(cs.ref_theta[40]),
                              velocity=9.0)
    preds = synthetic.synthetic_predictions(cs, 3, 30, 0.1, s0, np.random.default_rng(5))
    rp = ReactivePlannerHip(PlannerConfig(save_all_traj=True, sampling_min=1, sampling_max=2), VehicleParams(), engine=OracleEngine())
    rp.logger = DataLoggingCosts(str(tmp_path), save_all_traj=True, cost_params=dict(rp.cost_weights))
    rp.update_externals(reference_path=ref, x_0=x0, desired_velocity=11.0, predictions=preds)
    assert rp.plan() is not None
    n_all = len(rp.all_traj)
    rp.logger.close()
    logs = open(tmp_path / "logs.csv").read().splitlines()
    assert len(logs) == 2 and logs[1].startswith("7;") and logs[1].split(";")[4] == "True"
    assert len(logs[0].split(";")) == len(logs[1].split(";"))
    traj = open(tmp_path / "trajectories.csv").read().splitlines()
    assert len(traj) == 1 + n_all and traj[1].split(";")[0] == "7"
    assert open(tmp_path / "predictions.csv").read().count("\n") == 1
    con = sqlite3.connect(tmp_path / "trajectories.db")
    assert con.execute("SELECT COUNT(*) FROM trajectories").fetchone()[0] == n_all
    best = con.execute("SELECT id FROM costs ORDER BY costs_cumulative_weighted LIMIT 1").fetchone()[0]
    feas_best = con.execute("SELECT c.id FROM costs c JOIN infeasability i ON i.id = c.id AND i.time_step = c.time_step "
                            "WHERE i.feasible = 1 ORDER BY c.costs_cumulative_weighted LIMIT 1").fetchone()[0]
    assert feas_best == rp.optimal_trajectory.uniqueId or rp.infeasible_count_collision > 0
    assert best is not None
    con.close()


def test_prediction_dict_walked_in_c_equals_the_general_path():
    """csrc/fx_host_ext.c (`_fxhost`): the predictions dict walked in C and packed by fx_pack_predictions gives the arrays the
    general Python path gives, bit for bit; anything that is not a contiguous float64 array (lists, float32, strided views)
    falls back to that path; an empty prediction, a missing shape (no hulls) and a long-horizon predictor are handled alike."""
    import numpy as np
    from frenetix_motion_planner_amd import problem, synthetic
    from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
    from frenetix_motion_planner_amd import engine as eng
    assert eng._fxhost(), "the _fxhost extension is part of the build (make -C frenetix-motion-planner_amd/csrc)"
    cs = CoordinateSystem(synthetic.reference_polyline("arc", 400, 0.5, 0.01))
    base = synthetic.synthetic_predictions(cs, 6, 45, 0.1, 20.0, np.random.default_rng(3))
    keys = list(base)
    variants = {"plain": base}
    v = {k: dict(p) for k, p in base.items()}
    v[keys[1]] = dict(v[keys[1]], pos_list=v[keys[1]]["pos_list"][:0], cov_list=v[keys[1]]["cov_list"][:0],
                      orientation_list=v[keys[1]]["orientation_list"][:0])          # an obstacle without predictions
    del v[keys[2]]["shape"]                                                         # no shape: no hulls for that one
    v[keys[3]] = dict(v[keys[3]], pos_list=v[keys[3]]["pos_list"][:2], cov_list=v[keys[3]]["cov_list"][:2],
                      orientation_list=v[keys[3]]["orientation_list"][:2])          # <= 2 predictions: skipped by the collision stage
    variants["ragged"] = v
    lists = {k: dict(p, pos_list=p["pos_list"].tolist()) for k, p in base.items()}   # python lists: general path
    f32 = {k: dict(p, cov_list=p["cov_list"].astype(np.float32)) for k, p in base.items()}
    strided = {k: dict(p, pos_list=np.asfortranarray(p["pos_list"])) for k, p in base.items()}
    for name, preds in dict(variants, lists=lists, f32=f32, strided=strided).items():
        for n_samples in (31, 51):
            fast = problem.pack_predictions(preds, n_samples, eng.build_obstacle_hulls)
            saved = eng.build_obstacle_hulls.pack_dict
            try:
                del eng.build_obstacle_hulls.pack_dict
                general = problem.pack_predictions(preds, n_samples, eng.build_obstacle_hulls)
            finally:
                eng.build_obstacle_hulls.pack_dict = saved
            assert fast["K"] == general["K"] and fast["P"] == general["P"], name
            for k in ("pos", "cov_inv", "npred", "hull", "nhull"):
                assert np.array_equal(np.asarray(fast[k]), np.asarray(general[k])), (name, n_samples, k)
    # the C walk was really taken for the array-valued dicts and refused for the others
    h, addr = eng._fxhost()[:2]
    assert h.pack_predictions(addr, base, 31, 256) is not None and h.pack_predictions(addr, lists, 31, 256) is None
    assert h.pack_predictions(addr, f32, 31, 256) is None and h.pack_predictions(addr, strided, 31, 256) is None
    assert h.pack_predictions(addr, base, 31, 3) is None      # more obstacles than allowed: the Python path words the error


def test_state_update_struct_filled_in_c_equals_the_python_path():
    """_fxhost.state_update writes the same FxStateUpdate the ctypes assignments do; anything that is not a plain contiguous
    float64 / int32 array sends the caller to the long way (engine.FrenetEngine._state_update_of)."""
    import ctypes as C
    from frenetix_motion_planner_amd import _abi, engine, synthetic
    h = engine._fxhost()
    if not h:
        pytest.skip("_fxhost extension not built")
    for kw in (dict(n_obstacles=0), dict(n_obstacles=3), dict(n_obstacles=2, n_pred=2)):
        inp = synthetic.make_inputs(ref_kind="arc", v0=8.0, grid=(3, 5, 5), **kw)
        fast = engine.FrenetEngine._state_update_of(inp)
        saved, engine._FXHOST = engine._FXHOST, False
        try:
            slow = engine.FrenetEngine._state_update_of(inp)
        finally:
            engine._FXHOST = saved
        for name, _ in _abi.FxStateUpdate._fields_:
            a, b = getattr(fast, name), getattr(slow, name)
            assert (a or 0) == (b or 0), name
    a, b = np.arange(3.0), np.arange(4.0)
    u = _abi.FxStateUpdate()
    with pytest.raises((TypeError, ValueError, BufferError)):
        h[0].state_update(C.addressof(u), a, a, 0.0, 1.0, 0, b, b, b[::2], None, None, None, None, None)   # not contiguous
    with pytest.raises(TypeError):
        h[0].state_update(C.addressof(u), a, a, 0.0, 1.0, 0, b, b, np.arange(4, dtype=np.int32), None, None, None, None, None)
    with pytest.raises(ValueError):
        h[0].state_update(0, a, a, 0.0, 1.0, 0, b, b, b, None, None, None, None, None)


def test_next_inputs_in_c_equals_the_python_path(monkeypatch):
    """`_fxhost.next_inputs` (velocity set from its bounds, lateral set + current d, state arrays, copy of the last inputs -- one
    extension call) against PlanInputs.next_step behind sampling.dense_ranges: every field of the inputs, the arrays bit for bit,
    the structure key kept exactly when the Python path keeps it (same lengths, same (K, P)) and dropped when d0 joins or leaves
    the lateral set."""
    from frenetix_motion_planner_amd import reactive_planner as rpm
    from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
    assert rpm._NEXT_INPUTS is not None, "the _fxhost extension is part of the build (make -C frenetix-motion-planner_amd/csrc)"
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = CoordinateSystem(ref)
    rng = np.random.default_rng(5)

    def planner():
        p = ReactivePlannerHip(PlannerConfig(sampling_min=0, sampling_max=1, dense_grid=(5, 7, 9)), VehicleParams(), engine=OracleEngine())
        s0 = float(cs.ref_pos[40] + 0.1)
        x0 = ReactivePlannerState(0, np.asarray(cs.convert_to_cartesian_coords(s0, 0.2)), float(cs.ref_theta[40]), 9.0, 0.0, 0.0, 0.0)
        p.update_externals(reference_path=ref, x_0=x0, desired_velocity=10.0, predictions=synthetic.synthetic_predictions(
            cs, 3, 30, 0.1, s0, np.random.default_rng(1)))
        return p, x0

    pc, x0 = planner()
    pp, _ = planner()
    calls = []
    real = rpm._NEXT_INPUTS
    for step in range(8):
        v = float(rng.uniform(1.0, 15.0))
        d0 = [0.2, 0.0, -0.75, 0.31, 0.31, 3.0, 0.1234, 0.0][step]   # 0.0, -0.75, 3.0 are values of linspace(-3, 3, 9): no append
        x_cl = ([float(cs.ref_pos[40] + 0.1 + step), v, float(rng.normal(0, 0.3))], [d0, float(rng.normal(0, 0.1)), 0.0])
        st = ReactivePlannerState(step, x0.position.copy(), x0.orientation + 0.01 * step, v, 0.0, 0.0, 0.0)
        preds = synthetic.synthetic_predictions(cs, 3, 30, 0.1, x_cl[0][0], np.random.default_rng(step))
        out = []
        for p, fn in ((pc, lambda *a: (calls.append(1), real(*a))[1]), (pp, None)):
            monkeypatch.setattr(rpm, "_NEXT_INPUTS", fn)
            p.update_externals(x_0=st, x_cl=x_cl, desired_velocity=v + 1.0, predictions=preds)
            out.append(p.plan_begin())
        a, b = out
        assert type(a) is type(b) and set(a.__dict__) - {"_skey"} == set(b.__dict__) - {"_skey"}
        for k, vb in b.__dict__.items():
            va = a.__dict__.get(k)
            if k == "_skey":
                continue
            if isinstance(vb, np.ndarray):
                assert isinstance(va, np.ndarray) and va.dtype == vb.dtype and np.array_equal(va, vb), (step, k)
            elif k == "obstacles":
                assert all(np.array_equal(np.asarray(va[q]), np.asarray(vb[q])) for q in ("K", "P", "pos", "cov_inv", "npred", "hull", "nhull"))
            elif k in ("coordinate_system", "vehicle", "_ref", "_tpow", "cost_names", "_cost_id", "_cost_w", "_bound", "lanelets"):
                pass   # carried over by reference on both paths (per-planner objects)
            else:
                assert va == vb, (step, k, va, vb)
        assert ("_skey" in a.__dict__) == ("_skey" in b.__dict__), step
        ka, kb = a.structure_key(), b.structure_key()
        assert [x for i, x in enumerate(ka) if i != 5] == [x for i, x in enumerate(kb) if i != 5]   # (entry 5: the planners' own coordinate systems)
        assert a.t_samp.flags.c_contiguous and a.v_samp.flags.writeable and a.x0_lon.shape == (3,)
    assert len(calls) == 7   # every step after the first took the extension call (the first has no previous inputs)


def test_update_step_equals_update_externals():
    """the closed loop's inlined update (ReactivePlannerHip.update_step) leaves the planner exactly where update_externals does,
    and hands everything it does not cover to update_externals"""
    from frenetix_motion_planner_amd.coordinate_system import CoordinateSystem
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = CoordinateSystem(ref)
    s0 = float(cs.ref_pos[40] + 0.1)
    x0 = ReactivePlannerState(0, np.asarray(cs.convert_to_cartesian_coords(s0, 0.2)), float(cs.ref_theta[40]), 9.0, 0.0, 0.0, 0.0)
    preds = synthetic.synthetic_predictions(cs, 3, 30, 0.1, s0, np.random.default_rng(1))

    def fresh():
        p = ReactivePlannerHip(PlannerConfig(sampling_min=1, sampling_max=2), VehicleParams(), engine=OracleEngine())
        p.update_externals(reference_path=ref, x_0=x0, desired_velocity=10.0, predictions=preds)
        return p

    a, b = fresh(), fresh()
    for v in (9.5, 1.2, 0.0, 30.0):
        st = ReactivePlannerState(1, x0.position.copy(), x0.orientation, v, 0.1, 0.0, 0.0)
        x_cl = ([s0 + 1.0, v, 0.1], [0.3, 0.0, 0.0])
        a.update_externals(x_0=st, x_cl=x_cl, desired_velocity=v + 1.0, predictions=preds)
        b.update_step(st, x_cl, v + 1.0, preds)
        for k in ("x_0", "x_cl", "_LOW_VEL_MODE", "desired_velocity", "use_prediction", "predictions", "_packed_predictions"):
            assert getattr(a, k) is getattr(b, k) or getattr(a, k) == getattr(b, k), k
        va, vb = a.sampling_handler.v_sampling, b.sampling_handler.v_sampling
        assert (va.minimum, va.maximum, va.max_density) == (vb.minimum, vb.maximum, vb.max_density)
    # what it does not cover takes the long way: no x_cl yet (the initial state is computed), a new reference path
    c = ReactivePlannerHip(PlannerConfig(sampling_min=1, sampling_max=2), VehicleParams(), engine=OracleEngine())
    c.set_reference_and_coordinate_system(ref)
    c.update_step(x0, None, 10.0, preds)
    assert c.x_cl is not None and np.allclose(c.x_cl[0], a_cl0 := fresh().x_cl[0])


def test_plan_batch_marshalling_against_a_stand_in_library():
    """`_fxhost.plan_batch` / `plan_batch_begin` / `plan_batch_end` without a GPU: the function pointer they are handed is a ctypes
    callback with the C-ABI's signature (fx_plan_batch_packaged / _begin / _end) that checks what arrives -- every agent's
    FxStateUpdate pointing at the inputs' own arrays, yaw rates, one block pointer per agent -- and fills results, packages and
    blocks; the extension must hand those back as the result dicts of FxResult.as_dict(), pass an error code through, and refuse
    malformed arguments."""
    import ctypes as C
    from frenetix_motion_planner_amd import _fxhost
    agents = [synthetic.make_inputs(ref_kind="arc", v0=6.0 + a, grid=(2, 3, 3), n_obstacles=a % 2 + 1, seed=a,
                                    hull_builder=__import__("frenetix_motion_planner_amd.engine", fromlist=["x"]).build_obstacle_hulls)
              for a in range(3)]
    S = agents[0].n_samples
    seen = {}
    PU = C.POINTER(C.POINTER(_abi.FxStateUpdate))
    PD, PPD = C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_double))

    def check_updates(n, upd):
        for a in range(n):
            u, inp = upd[a].contents, agents[a]
            assert u.x0_lon == inp.x0_lon.ctypes.data and u.x0_lat == inp.x0_lat.ctypes.data and u.t_samp == inp.t_samp.ctypes.data
            assert u.v_samp == inp.v_samp.ctypes.data and u.d_samp == inp.d_samp.ctypes.data
            assert u.v_des == inp.v_des and u.low_vel_mode == int(inp.low_vel_mode)
            assert u.obs_pos == inp.obstacles["pos"].ctypes.data and u.obs_cov_inv == inp.obstacles["cov_inv"].ctypes.data
            assert u.obs_npred == inp.obstacles["npred"].ctypes.data and u.obs_hull == inp.obstacles["hull"].ctypes.data
            assert u.obs_nhull == inp.obstacles["nhull"].ctypes.data
            assert abs(u.x0_orientation - inp.x0_orientation) == 0.0

    def fill(n, yaw, res, pkg, blocks):
        for a in range(n):
            res[a].n_candidates, res[a].best_index, res[a].best_cost = 18, 7 + a, 1.5 * (a + 1)
            res[a].n_returned, res[a].n_feasible, res[a].n_infeasible, res[a].n_collisions = 18, 12, 6, a
            for k in range(_abi.FX_NUM_REASONS):
                res[a].reason_hist[k] = 10 * a + k
            res[a].feasible_percentage, res[a].kernel_ms = 66.5, -1.0
            pkg[a].found, pkg[a].index, pkg[a].cost = 1, 7 + a, 1.5 * (a + 1)
            for i in range(_abi.FX_PKG_ROWS * S):
                blocks[a][i] = 1000.0 * a + i + yaw[a]

    @C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, PU, PD, C.POINTER(_abi.FxResult), C.POINTER(_abi.FxPackage), PPD)
    def whole(ctx, n, upd, yaw, res, pkg, blocks):
        try:
            seen["ctx"], seen["n"], seen["upd"] = ctx, n, bool(upd)
            if upd:
                check_updates(n, upd)
            fill(n, yaw, res, pkg, blocks)
            return 0
        except Exception as ex:   # (an exception cannot cross the C frame: reported through the return code)
            seen["error"] = repr(ex)
            return -99

    @C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, PU)
    def begin(ctx, n, upd):
        try:
            check_updates(n, upd)
            seen["begun"] = n
            return 0
        except Exception as ex:
            seen["error"] = repr(ex)
            return -99

    @C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, PD, C.POINTER(_abi.FxResult), C.POINTER(_abi.FxPackage), PPD)
    def end(ctx, n, yaw, res, pkg, blocks):
        fill(n, yaw, res, pkg, blocks)
        return 0

    @C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, PU, PD, C.POINTER(_abi.FxResult), C.POINTER(_abi.FxPackage), PPD)
    def failing(ctx, n, upd, yaw, res, pkg, blocks):
        return 3

    addr = lambda f: C.cast(f, C.c_void_p).value
    yaw = [0.25, 0.5, 0.75]

    def verify(out, pkgs, blocks):
        assert "error" not in seen, seen.get("error")
        assert len(out) == 3
        for a, d in enumerate(out):
            assert set(d) == set(_abi.FxResult().as_dict())
            assert d["best_index"] == 7 + a and d["best_cost"] == 1.5 * (a + 1) and d["n_collisions"] == a
            assert d["reason_hist"] == [10 * a + k for k in range(_abi.FX_NUM_REASONS)] and d["kernel_ms"] == -1.0
            assert pkgs[a].found == 1 and pkgs[a].index == 7 + a
            assert blocks[a, 0, 0] == 1000.0 * a + yaw[a] and blocks[a, -1, -1] == 1000.0 * a + _abi.FX_PKG_ROWS * S - 1 + yaw[a]

    pkgs, blocks = (_abi.FxPackage * 3)(), np.zeros((3, _abi.FX_PKG_ROWS, S))
    out = _fxhost.plan_batch(addr(whole), 0x1234, agents, yaw, blocks, C.addressof(pkgs), True)
    assert seen["ctx"] == 0x1234 and seen["n"] == 3 and seen["upd"] is True
    verify(out, pkgs, blocks)
    out = _fxhost.plan_batch(addr(whole), 0x1234, agents, yaw, blocks, C.addressof(pkgs), False)   # resident inputs: no updates
    assert seen["upd"] is False
    verify(out, pkgs, blocks)
    pkgs, blocks = (_abi.FxPackage * 3)(), np.zeros((3, _abi.FX_PKG_ROWS, S))
    assert _fxhost.plan_batch_begin(addr(begin), 0x1234, agents, True) is None and seen["begun"] == 3
    verify(_fxhost.plan_batch_end(addr(end), 0x1234, 3, yaw, blocks, C.addressof(pkgs)), pkgs, blocks)
    # a library error comes back as its code; malformed arguments are refused before anything is called
    assert _fxhost.plan_batch(addr(failing), 0x1234, agents, yaw, blocks, C.addressof(pkgs), True) == 3
    with pytest.raises(ValueError):
        _fxhost.plan_batch(addr(whole), 0x1234, agents, yaw[:2], blocks, C.addressof(pkgs), True)
    with pytest.raises(ValueError):
        _fxhost.plan_batch(addr(whole), 0x1234, agents, yaw, np.zeros(7), C.addressof(pkgs), True)
    with pytest.raises(ValueError):
        _fxhost.plan_batch(0, 0x1234, agents, yaw, blocks, C.addressof(pkgs), True)
    with pytest.raises((TypeError, ValueError, AttributeError)):
        _fxhost.plan_batch(addr(whole), 0x1234, [object()], [0.0], np.zeros((1, _abi.FX_PKG_ROWS, S)), C.addressof(pkgs), True)


def test_state_update_states_the_array_shapes_and_refuses_arrays_that_do_not_fit_together():
    """FxStateUpdate carries nT / nV / nD / K / P of the caller's arrays (ABI 8): fx_update_state copies with the UPLOAD's counts out of
    borrowed pointers, so a stale structure key must end in FX_ERR_INVALID_ARGUMENT, not in a read past the arrays' end.
    Here: the counts are filled by both host paths, and obstacle arrays whose lengths contradict each other are refused in C."""
    import ctypes as C
    from frenetix_motion_planner_amd import _abi, engine, synthetic
    inp = synthetic.make_inputs(ref_kind="arc", v0=8.0, grid=(3, 5, 7), n_obstacles=3)
    o = inp.obstacles
    for use_c in (True, False):
        saved = engine._FXHOST
        if not use_c:
            engine._FXHOST = False
        try:
            u = engine.FrenetEngine._state_update_of(inp)
        finally:
            engine._FXHOST = saved
        assert (u.nT, u.nV, u.nD) == (len(inp.t_samp), len(inp.v_samp), len(inp.d_samp))
        assert (u.K, u.P) == (o["K"], o["P"]) == (3, o["pos"].shape[1])
    m = engine.FrenetEngine.make_state_update(t_samp=inp.t_samp, d_samp=inp.d_samp, obstacles=o)
    assert (m.nT, m.nV, m.nD, m.K, m.P) == (len(inp.t_samp), 0, len(inp.d_samp), o["K"], o["P"])
    h = engine._fxhost()
    if not h:
        pytest.skip("_fxhost extension not built")
    a, t = np.arange(3.0), np.arange(4.0)
    u = _abi.FxStateUpdate()
    with pytest.raises(ValueError):   # cov_inv for another (K, P) than pos / npred say
        h[0].state_update(C.addressof(u), a, a, 0.0, 1.0, 0, t, t, t, o["pos"], o["cov_inv"][:2], o["npred"], None, None)
    with pytest.raises(ValueError):   # hulls for another P
        h[0].state_update(C.addressof(u), a, a, 0.0, 1.0, 0, t, t, t, o["pos"], o["cov_inv"], o["npred"], o["hull"][:, :-1].copy(), o["nhull"])
    with pytest.raises(ValueError):   # a state of two values
        h[0].state_update(C.addressof(u), a[:2], a, 0.0, 1.0, 0, t, t, t, None, None, None, None, None)


class _ToyOcclusionModule:
    """An occlusion module by its two call points only (the real one is outside the reference tree): calc_costs adds a visibility
    cost that grows with the lateral end offset to the LEFT, trajectory_safety_assessment vetoes end velocities above a bound."""

    def __init__(self, v_veto=None, weight=0.0, invalidate=()):
        self.v_veto, self.weight, self.calc_calls, self.assessed, self.invalidate = v_veto, weight, [], [], set(invalidate)

    def calc_costs(self, trajectories):
        self.calc_calls.append([t.uniqueId for t in trajectories])
        for t in trajectories:
            t.cost = t.cost + self.weight * max(0.0, t.sampling_parameters[10])
            if t.uniqueId in self.invalidate:      # the real module: harm above the maximum harm -> valid = False
                t.valid = False

    def trajectory_safety_assessment(self, trajectory):
        self.assessed.append(trajectory.uniqueId)
        ok = self.v_veto is None or trajectory.sampling_parameters[5] <= self.v_veto
        return 0.5, ok


def _open_road_planner(engine="oracle", **cfg):
    rp = ReactivePlannerHip(PlannerConfig(**cfg), VehicleParams(), engine=OracleEngine() if engine == "oracle" else engine)
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = synthetic.CoordinateSystem(ref)
    xy = cs.convert_to_cartesian_coords(float(cs.ref_pos[40]) + 0.1, 0.2)
    x0 = ReactivePlannerState(time_step=0, position=np.array(xy), orientation=float(cs.ref_theta[40]), velocity=9.0)
    n = 31
    car = dict(pos_list=np.stack([xy[0] + 12.0 + 0.4 * np.arange(n), np.full(n, xy[1] + 0.3)], axis=1), cov_list=np.tile(np.eye(2) * 0.2, (n, 1, 1)),
               orientation_list=np.full(n, float(cs.ref_theta[40])), shape=dict(length=4.5, width=1.9))
    rp.update_externals(reference_path=ref, x_0=x0, desired_velocity=9.0, predictions={3: car})
    return rp


def test_occlusion_module_call_points():
    """planner.py:271-273, 384-388 and trajectories.py:557-560: with an occlusion module set, its calc_costs sees the step's feasible
    trajectories in creation order before they are sorted, and its safety assessment is asked for the collision-free candidates in
    cost order until one passes.  Restated over the step's arrays."""
    base = _open_road_planner()
    assert base.plan() is not None
    want_plain = base.optimal_trajectory.uniqueId

    # a module that changes nothing and vetoes nothing: the device's own winner, asked exactly once
    occ = _ToyOcclusionModule()
    rp = _open_road_planner()
    rp.set_occlusion_module(occ)
    assert rp.use_occ_model and rp.plan() is not None
    step = rp.last_step
    costed = np.nonzero(step.mask(_abi.FX_FLAG_COSTED))[0]
    assert occ.calc_calls == [list(costed)] and rp.optimal_trajectory.uniqueId == want_plain == step.result["best_index"]
    assert occ.assessed == [want_plain] and rp._collision_counter == step.result["n_collisions"]

    # added costs re-order the list, a veto skips candidates: the walk restated with NumPy on the device's arrays
    occ = _ToyOcclusionModule(v_veto=float(np.sort(np.unique(base.last_step.inputs.v_samp))[-2]), weight=3.0)
    rp = _open_road_planner()
    rp.set_occlusion_module(occ)
    assert rp.plan() is not None
    step = rp.last_step
    inp = step.inputs
    ids = np.nonzero(step.mask(_abi.FX_FLAG_COSTED))[0]
    sp = np.array([step.sample(int(g)).sampling_parameters for g in ids])
    cost = step.cost[ids] + 3.0 * np.maximum(0.0, sp[:, 10])
    order = ids[np.argsort(cost, kind="stable")]
    sel, col = step.mask(_abi.FX_FLAG_SELECTABLE), step.mask(_abi.FX_FLAG_COLLISION)
    expect_asked, n_col, want = [], 0, None
    for g in order:
        if not sel[g]:
            continue
        if col[g]:
            n_col += 1
            continue
        expect_asked.append(int(g))
        if step.sample(int(g)).sampling_parameters[5] <= occ.v_veto:
            want = int(g)
            break
    assert want is not None and occ.assessed == expect_asked and rp.optimal_trajectory.uniqueId == want
    assert rp._collision_counter == n_col and rp.optimal_trajectory.cost == pytest.approx(float(cost[list(ids).index(want)]))
    # the batched path's phases take the same decision
    occ2 = _ToyOcclusionModule(v_veto=occ.v_veto, weight=3.0)
    rp2 = _open_road_planner()
    rp2.set_occlusion_module(occ2)
    inp2 = rp2.plan_begin()
    res = rp2.engine.plan_batch([inp2])[0]
    assert rp2.plan_consume(inp2, res, rp2.engine, 0).uniqueId == want
    # candidates the module invalidated in calc_costs never reach the collision walk (planner.py:338-339): neither counted
    # as collisions, nor assessed, nor returned -- here the plain winner and the first colliding candidate of the cost order
    order0 = step.sorted_ids()
    first_col = next(int(g) for g in base.last_step.sorted_ids() if base.last_step.mask(_abi.FX_FLAG_COLLISION)[g])
    occ4 = _ToyOcclusionModule(invalidate=(want_plain, first_col))
    rp4 = _open_road_planner()
    rp4.set_occlusion_module(occ4)
    assert rp4.plan() is not None
    st4 = rp4.last_step
    walk4 = [int(g) for g in st4.sorted_ids() if st4.mask(_abi.FX_FLAG_SELECTABLE)[g] and int(g) not in occ4.invalidate]
    n_col4 = 0
    for g in walk4:
        if st4.mask(_abi.FX_FLAG_COLLISION)[g]:
            n_col4 += 1
            continue
        break
    assert rp4.optimal_trajectory.uniqueId == g != want_plain and want_plain not in occ4.assessed
    assert rp4._collision_counter == n_col4
    # a module that vetoes everything leaves the step without a trajectory (the planner escalates, then stands still)
    occ3 = _ToyOcclusionModule(v_veto=-1.0)
    rp3 = _open_road_planner(sampling_min=2, sampling_max=3)
    rp3.set_occlusion_module(occ3)
    rp3.plan()
    assert len(occ3.assessed) > 0 and (rp3.optimal_trajectory is None or not hasattr(rp3.optimal_trajectory, "uniqueId") or rp3.optimal_trajectory.uniqueId not in occ3.assessed)
    rp3.set_occlusion_module(None)
    assert not rp3.use_occ_model


def test_level_inputs_in_one_extension_call_equal_the_python_path(monkeypatch):
    """`_fxhost.next_inputs_level` (the closed loop's usual step at a sampling level of the reference: set(np.linspace(v_min, v_max, n)) and
    d_level.union({d}) in the sets' OWN iteration order, built with the same set operations in C) against the Python path -- ordered
    ranges + PlanInputs.next_step -- step by step: same arrays, same scalars, same structure key; levels 1 and 2, lateral positions on
    and off the level's values, a changed time sampling and a changed level fall back to the Python path and come out equal too."""
    from frenetix_motion_planner_amd import reactive_planner as rpm
    fn = rpm._NEXT_INPUTS_LEVEL
    if fn is None:
        pytest.skip("_fxhost is not built")
    ref = synthetic.reference_polyline("arc", 400, 0.5, 0.01)
    cs = synthetic.CoordinateSystem(ref)
    s0 = float(cs.ref_pos[40] + 0.1)
    preds = synthetic.synthetic_predictions(cs, 5, 30, 0.1, s0, np.random.default_rng(1))
    pa = ReactivePlannerHip(PlannerConfig(sampling_min=1, sampling_max=3), VehicleParams(), engine=OracleEngine())
    pb = ReactivePlannerHip(PlannerConfig(sampling_min=1, sampling_max=3), VehicleParams(), engine=OracleEngine())
    rng = np.random.default_rng(5)
    calls = []
    counted = lambda *a: (calls.append(1), fn(*a))[1]
    for k in range(120):
        v = float(rng.uniform(0.5, 20.0))
        d = float(rng.choice([rng.uniform(-1.5, 1.5), 0.0, -3.0, 0.75, 1.5]))
        x0 = ReactivePlannerState(k, np.asarray(cs.convert_to_cartesian_coords(s0 + 0.3 * k, d)), float(cs.ref_theta[40] + 0.005 * k), v)
        for p in (pa, pb):
            p.update_externals(reference_path=ref if k == 0 else None, x_0=x0, desired_velocity=12.0 + 0.01 * k, predictions=preds)
            if k == 60:
                p.set_sampling_parameters(1.3, 3.0, -2.5, 2.5)    # new time / lateral sampling objects
        lvl = 1 if k % 25 == 24 else 2
        monkeypatch.setattr(rpm, "_NEXT_INPUTS_LEVEL", counted)
        ia = pa._inputs_for_level(lvl)
        monkeypatch.setattr(rpm, "_NEXT_INPUTS_LEVEL", None)
        ib = pb._inputs_for_level(lvl)
        for f in ("t_samp", "v_samp", "d_samp", "x0_lon", "x0_lat"):
            assert np.array_equal(getattr(ia, f), getattr(ib, f)), (k, f)
        ka, kb = ia.structure_key(), ib.structure_key()
        assert ka[:5] + ka[6:] == kb[:5] + kb[6:]      # (entry 5 is the coordinate system's identity: one per planner)
        assert (ia.low_vel_mode, ia.x0_orientation, ia.v_des, ia.n_candidates) == (ib.low_vel_mode, ib.x0_orientation, ib.v_des, ib.n_candidates)
        assert ia.obstacles is pa._packed_predictions and ia.cost_weights == ib.cost_weights
    assert 100 <= len(calls) < 120      # the first step, the level changes and the new sampling objects took the long way
