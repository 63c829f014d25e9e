#!/usr/bin/env python3
"""Host time of the closed-loop multi-agent planning step WITHOUT a GPU (run by hand: python tests/host_step_profile.py [--prof]).

The agents' first sampling level runs on the oracle-backed engine with the winner-package surface
(tests/oracle_engine.PackagingOracleEngine) on a tiny dense grid; the time inside the engine calls is subtracted, what is left is
the Python / C host work around the batched launch -- the part of BASELINE config 4 that does not depend on the device.
Test infrastructure (imports the oracle); the measured numbers of the real engine are tools/seg_config4.py's."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frenetix_motion_planner_amd import commonroad_xml as crx   # noqa: E402
from frenetix_motion_planner_amd.multiagent import MultiAgentSimulation   # noqa: E402
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig   # noqa: E402
from tests.oracle_engine import PackagingOracleEngine   # noqa: E402


ACC = {}
CALLS = {}


def wrap(cls, name):
    f = getattr(cls, name)
    label = f"{cls.__name__}.{name}"

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            ACC[label] = ACC.get(label, 0.0) + dt
            CALLS.setdefault(label, []).append(dt)
    setattr(cls, name, g)


def instrument():
    from frenetix_motion_planner_amd import frenet_interface, multiagent, reactive_planner
    for cls, names in ((frenet_interface.FrenetPlannerInterfaceHip, ("update_planner", "begin_step", "finish_step", "needs_plan")),
                       (frenet_interface.VelocityPlanner, ("calculate_desired_velocity",)),
                       (reactive_planner.ReactivePlannerHip, ("plan_consume", "plan_finish", "_inputs_for_level", "update_externals",
                                                              "_compute_trajectory_pair", "_consume_result", "record_state_and_input")),
                       (multiagent.MultiAgentSimulation, ("predictions_for", "packed_predictions_for", "_shared_predictions", "_exchange")),
                       (multiagent.AgentBatchHip, ("step",))):
        for n in names:
            wrap(cls, n)


def main():
    if "--seg" in sys.argv:
        instrument()
    sc = crx.read_scenario_json(os.path.join(ROOT, "tests", "golden", "ZAM_Tjunction-1_42_T-1.scenario.json"))
    cfg = PlannerConfig(sampling_min=0, sampling_max=1, dense_grid=(3, 5, 5))
    sim = MultiAgentSimulation(sc, config=cfg, engine_factory=PackagingOracleEngine)
    eng = sim.batch.engine
    for _ in range(6):
        sim.step()
    ACC.clear(); CALLS.clear()
    n = int(os.environ.get("STEPS", "90"))
    host = []
    for _ in range(n):
        e0 = eng.engine_s
        t0 = time.perf_counter()
        sim.step()
        host.append((time.perf_counter() - t0) - (eng.engine_s - e0))
    host = np.array(host) * 1e6
    plan = host[host > np.median(host) * 2]
    print(f"host time outside the engine: planning steps ({len(plan)}) p50 {np.median(plan):.1f} us min {plan.min():.1f} us, "
          f"others p50 {np.median(host[host <= np.median(host) * 2]):.1f} us")
    if "--seg" in sys.argv:
        for k, v in sorted(ACC.items(), key=lambda kv: -kv[1]):
            q = np.array(CALLS[k]) * 1e6
            print(f"  {k:48s} {v / max(len(plan), 1) * 1e6:9.1f} us per planning step (inclusive)   per call: n {len(q):5d} p50 {np.median(q):7.1f} "
                  f"min {q.min():7.1f}")
    if "--prof" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(30):
            sim.step()
        pr.disable()
        st = pstats.Stats(pr)
        rows = sorted(((tt, ct, nc, f"{os.path.basename(fn)}:{ln}({name})") for (fn, ln, name), (cc, nc, tt, ct, _) in st.stats.items()
                       if "oracle" not in fn), reverse=True)
        print("own time per planning step under cProfile, us")
        for tt, ct, nc, where in rows[:45]:
            print(f"  {tt * 1e5:8.1f} own  {ct * 1e5:8.1f} incl  {nc / 10:7.1f} calls  {where}")
    sim.close()


if __name__ == "__main__":
    main()
