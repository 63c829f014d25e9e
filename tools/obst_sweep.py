#!/usr/bin/env python3
"""Obstacle-stage sweep on the GPU box: evaluation-kernel time of config 3 (Mode A / B) and the 1 M x 20 obstacles
case for (lanes per candidate, waves per SIMD, part mapping)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls


def run(label, tunings, steps=30, **kw):
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
    out = {}
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        for tn in tunings:
            G, w, var, blk = tn[:4]; mp = tn[4] if len(tn) > 4 else 0
            eng.set_timing("kernel"); eng.set_tuning(G, w, var, blk, mp); eng.upload(inp)
            for _ in range(3): eng.evaluate(); eng.finish()
            ts = []
            for _ in range(steps):
                eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
            out[f"G{G}w{w}b{blk}m{mp}"] = round(float(np.median(ts)) * 1e3, 1)
    print(label, inp.n_candidates, json.dumps(out), flush=True)


T = [(0, 0, 0, 0), (1, 2, 2, 256), (1, 3, 2, 256), (1, 4, 2, 256), (2, 2, 2, 256, 2), (2, 3, 2, 256, 2), (2, 4, 2, 256, 2),
     (4, 2, 2, 256, 2), (4, 4, 2, 256, 2), (2, 2, 2, 256, 1)]
run("config3_modeB", T, ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20)
run("config3_modeA", T, ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, write_bundle=False, write_costmap=False)
run("1M_modeA_obs20", T, steps=8, ref_kind="arc", v0=10.0, grid=(19, 230, 229), n_obstacles=20, write_bundle=False, write_costmap=False)
run("200k_modeA_obs20_N50", T, steps=8, ref_kind="arc", v0=10.0, horizon=5.0, n_pred=50, grid=(39, 72, 71), n_obstacles=20,
    write_bundle=False, write_costmap=False)
