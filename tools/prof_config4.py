#!/usr/bin/env python3
"""cProfile of the closed-loop multi-agent simulation step (BASELINE config 4) on the engine."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frenetix_motion_planner_amd import commonroad_xml as crx
from frenetix_motion_planner_amd.multiagent import MultiAgentSimulation
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = crx.read_scenario_json(os.path.join(ROOT, "tests", "golden", "ZAM_Tjunction-1_42_T-1.scenario.json"))
cfg = PlannerConfig(sampling_min=0, sampling_max=1, dense_grid=(19, 23, 23))
sim = MultiAgentSimulation(sc, config=cfg, device=0, freeze_gc=True)
for _ in range(6): sim.step()
t = []
for _ in range(30):
    t0 = time.perf_counter(); sim.step(); t.append(time.perf_counter() - t0)
print("sim step ms: planning steps", np.round(np.array(t[::3]) * 1e3, 3), "others", np.round(np.median(np.array(t)[np.arange(30) % 3 != 0]) * 1e3, 3), "last batch ms", sim.batch.last_batch_ms)
pr = cProfile.Profile(); pr.enable()
for _ in range(30): sim.step()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(38)
if "--tottime" in sys.argv:   # own time per PLANNING step (10 of the 30 profiled steps plan), microseconds
    rows = sorted(((tt, ct, nc, f"{os.path.basename(fn)}:{ln}({name})") for (fn, ln, name), (cc, nc, tt, ct, _) in st.stats.items()), reverse=True)
    print(f"own time per planning step, us (total {sum(r[0] for r in rows) * 1e5:.0f})")
    for tt, ct, nc, where in rows[:80]:
        print(f"  {tt * 1e5:8.1f} own  {ct * 1e5:8.1f} incl  {nc / 10:7.1f} calls  {where}")
sim.close()
