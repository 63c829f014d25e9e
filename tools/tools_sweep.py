#!/usr/bin/env python3
"""Tuning sweep on the GPU box: evaluation-kernel time for every (lanes per candidate, waves per SIMD)."""
import json
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic  # noqa: E402
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls  # noqa: E402


def run(label, steps=60, **kw):
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
    out = {}
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        eng.set_timing("kernel")
        for blk in (64, 128, 256):
          for G in (1, 2, 4):
            for w in (2, 3, 4):
                variant = 2
                eng.set_tuning(G, w, variant, blk)
                eng.upload(inp)
                for _ in range(5):
                    eng.evaluate(); eng.finish()
                ts = []
                for _ in range(steps):
                    eng.evaluate(); eng.finish()
                    ts.append(eng.last_eval_kernel_ms)
                out[f"b{blk}_G{G}_w{w}"] = round(float(np.median(ts)) * 1e3, 1)
    print(label, inp.n_candidates, json.dumps(out))


if __name__ == "__main__":
    run("config2_modeB", ref_kind="arc", v0=10.0, grid=(19, 51, 51))
    run("config2_modeA", ref_kind="arc", v0=10.0, grid=(19, 51, 51), write_bundle=False, write_costmap=False)
    run("config3_modeB", ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20)
    run("config3_modeA", ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, write_bundle=False, write_costmap=False)
    run("1M_modeA_obs20", steps=10, ref_kind="arc", v0=10.0, grid=(19, 230, 229), n_obstacles=20, write_bundle=False, write_costmap=False)
    run("1M_modeA", steps=10, ref_kind="arc", v0=10.0, grid=(19, 230, 229), write_bundle=False, write_costmap=False)
