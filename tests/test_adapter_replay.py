"""The call sequence the reference's `ReactivePlannerCpp.plan()` makes on the `frenetix` module (reactive_planner_cpp.py:96-178,
228-256, 292-441), recorded as data in the build container (tests/dropin/run_reference_cpp_planner.py --record: functor
keyword arguments, the 800 x 13 sampling matrix, PoseWithCovariance / PredictedObject payloads, compute_initial_state) and
replayed here against this package's frenetix_compat -- on the oracle stand-in in the CPU suite (checks the recorder), on the
HIP engine on the GPU box (checks the engine behind the reference's own calling convention)."""
import os

import numpy as np
import pytest

from tests.dropin.trace_recorder import CLASSES, replay

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACES = ["cpp_adapter_trace", "cpp_adapter_trace_blocked"]


def namespace():
    from frenetix_motion_planner_amd import frenetix_compat as fc
    ns = {n: getattr(fc, n) for names in CLASSES.values() for n in names}
    ns["compute_initial_state"] = fc.compute_initial_state
    return ns


def check(name, on_new=None):
    from frenetix_motion_planner_amd import frenetix_compat as fc
    objs, captured, expected = replay(os.path.join(GOLD, name + ".json"), namespace(), on_new)
    handler = next(o for o in objs.values() if isinstance(o, fc.TrajectoryHandler))
    n_sorted = 0
    for ev, got in captured:
        if ev["op"] == "func":  # compute_initial_state
            assert np.allclose(got["x0_lon"], ev["result"]["x0_lon"], rtol=1e-12, atol=1e-12)
            assert np.allclose(got["x0_lat"], ev["result"]["x0_lat"], rtol=1e-12, atol=1e-12)
            continue
        ref = ev["sorted"]
        n_sorted += 1
        assert got["n"] == ref["n"] and got["n_feasible"] == ref["n_feasible"]
        assert np.allclose(got["costs"], ref["costs"], rtol=1e-9, atol=0)
        if got["ids"] != ref["ids"]:  # only neighbours whose costs agree to the tolerance may swap
            for a, b, ca, cb in zip(got["ids"], ref["ids"], got["costs"], ref["costs"]):
                assert a == b or abs(ca - cb) <= 1e-9 * max(1.0, abs(cb))
        else:
            assert got["feasible"] == ref["feasible"]
    assert n_sorted >= 1
    res = handler.last_result
    assert res["n_candidates"] == expected["n_matrix"] == 800
    assert res["n_collisions"] == expected["collisions"]
    if expected["collisions"] == 0:
        # nothing collides: the reference's walk takes the head of the feasible list = the engine's winner
        assert res["best_index"] == expected["optimal_id"] and res["best_cost"] == pytest.approx(expected["optimal_cost"], rel=1e-9)
    else:
        # every feasible candidate collides: no winner; the reference then picks a stopping trajectory from the sorted list
        assert res["best_index"] == -1
        step = handler._step
        tr = step.sample(expected["optimal_id"])
        assert tr.cost == pytest.approx(expected["optimal_cost"], rel=1e-9)
        assert np.allclose(tr.sampling_parameters, expected["sampling_parameters"])
    assert res["feasible_percentage"] == pytest.approx(expected["feasible_percentage"])
    return handler


@pytest.mark.parametrize("name", TRACES)
def test_trace_replays_on_the_oracle_stand_in(name):
    from frenetix_motion_planner_amd import frenetix_compat as fc
    from tests.oracle_engine import OracleEngine

    def on_new(o):
        if isinstance(o, fc.TrajectoryHandler):
            o._engine = OracleEngine()
    check(name, on_new)


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRACES)
def test_reference_adapter_trace_replays_on_the_hip_engine(name):
    handler = check(name)
    handler.engine.close()
