#!/usr/bin/env python3
"""Memory growth of the batched packaged calls (GPU box): plan_batch_packaged and its two halves over many steps on the same
agents -- the result dicts, packages and blocks of a step are garbage once the next one is taken."""
import gc, os, resource, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

agents = [synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a, grid=(3, 5, 7), n_obstacles=1 + a % 3,
                                seed=a) for a in range(4)]
yaw = [0.0] * 4
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
with FrenetEngine(max_candidates=4096, max_agents=4) as eng:
    for _ in range(2000):
        eng.plan_batch_packaged(agents, yaw)
    gc.collect(); r0 = rss()
    for rep in range(3):
        for _ in range(20000):
            res, pk = eng.plan_batch_packaged(agents, yaw)
            tok = eng.plan_batch_begin(agents)
            res, pk = eng.plan_batch_end(tok, yaw)
        gc.collect()
        print(f"after {(rep + 1) * 40000} calls: rss growth {rss() - r0} KB", flush=True)
