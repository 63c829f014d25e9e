#!/usr/bin/env python3
"""Wall-clock segments of a planning step of the closed-loop multi-agent simulation (BASELINE config 4): where the host time
of the five agents goes (accumulated perf_counter around the calls of AgentBatchHip.step)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import commonroad_xml as crx
from frenetix_motion_planner_amd import multiagent, frenet_interface, reactive_planner, engine
from frenetix_motion_planner_amd.multiagent import MultiAgentSimulation
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig

ACC = collections.defaultdict(float)
CALLS = collections.defaultdict(list)


def wrap(cls, name, label=None):
    f = getattr(cls, name)
    label = label or f"{cls.__name__}.{name}"

    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            ACC[label] += dt
            CALLS[label].append(dt)
    setattr(cls, name, g)


for cls, names in ((frenet_interface.FrenetPlannerInterfaceHip, ("update_planner", "begin_step", "finish_step", "needs_plan")),
                   (reactive_planner.ReactivePlannerHip, ("plan_consume", "plan_finish", "_inputs_for_level", "update_externals",
                                                          "_compute_trajectory_pair", "_consume_result")),
                   (engine.FrenetEngine, ("plan_batch", "plan_batch_packaged", "package", "evaluate", "finish", "update_state")),
                   (multiagent.MultiAgentSimulation, ("predictions_for", "packed_predictions_for", "_shared_predictions", "_pack_shared_arrays", "_exchange")),
                   (multiagent.AgentBatchHip, ("step",))):
    for n in names:
        if hasattr(cls, n):
            wrap(cls, n)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = crx.read_scenario_json(os.path.join(ROOT, "tests", "golden", "ZAM_Tjunction-1_42_T-1.scenario.json"))
sim = MultiAgentSimulation(sc, config=PlannerConfig(sampling_min=0, sampling_max=1, dense_grid=(19, 23, 23)), device=0, freeze_gc=True)
for _ in range(6):
    sim.step()
ACC.clear(); CALLS.clear()
t = []
n = 60
for _ in range(n):
    t0 = time.perf_counter(); sim.step(); t.append(time.perf_counter() - t0)
t = np.array(t)
plan = t[t > np.median(t) * 3]
print(f"simulation step: mean {t.mean()*1e3:.3f} ms, planning steps ({len(plan)}) p50 {np.median(plan)*1e3:.3f} ms, others p50 {np.median(t[t <= np.median(t)*3])*1e3:.3f} ms")
for k, v in sorted(ACC.items(), key=lambda kv: -kv[1]):
    q = np.array(CALLS[k]) * 1e6
    print(f"  {k:48s} {v / max(len(plan), 1) * 1e6:9.1f} us per planning step (inclusive)   per call: n {len(q):5d} p50 {np.median(q):7.1f} "
          f"mean {q.mean():7.1f} max {q.max():8.1f}")
sim.close()
