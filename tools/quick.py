#!/usr/bin/env python3
"""Quick A/B of the evaluation kernel on the GPU box: config2 (50k) Mode A/B, config3, 1M; selected tunings."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

def run(label, tunings, steps=60, **kw):
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
    out = {}
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        for tn in tunings:
            G, w, var, blk = tn[:4]; mp = tn[4] if len(tn) > 4 else 0
            eng.set_timing("kernel"); eng.set_tuning(G, w, var, blk, mp); eng.upload(inp)
            for _ in range(5): eng.evaluate(); eng.finish()
            ts = []
            for _ in range(steps):
                eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
            out[f"G{G}w{w}v{var}b{blk}m{mp}"] = round(float(np.median(ts)) * 1e3, 1)
    print(label, inp.n_candidates, json.dumps(out), flush=True)

T = [(0, 0, 0, 0), (1, 2, 2, 256), (2, 2, 2, 256, 1), (2, 2, 2, 256, 2), (4, 2, 2, 256, 2), (4, 3, 2, 256, 2), (4, 4, 2, 256, 2), (1, 4, 2, 256)]
run("config2_modeB", T, ref_kind="arc", v0=10.0, grid=(19, 51, 51))
run("config2_modeA", T, ref_kind="arc", v0=10.0, grid=(19, 51, 51), write_bundle=False, write_costmap=False)
run("config3_modeB", T, ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20)
run("config3_modeA", T, ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, write_bundle=False, write_costmap=False)
run("1M_modeA_obs20", T, steps=10, ref_kind="arc", v0=10.0, grid=(19, 230, 229), n_obstacles=20, write_bundle=False, write_costmap=False)
run("1M_modeA", T, steps=10, ref_kind="arc", v0=10.0, grid=(19, 230, 229), write_bundle=False, write_costmap=False)
run("1M_modeB", T, steps=10, ref_kind="arc", v0=10.0, grid=(19, 230, 229))
run("200k_modeB", T, steps=20, ref_kind="arc", v0=10.0, grid=(19, 102, 102))
