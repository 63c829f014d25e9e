#!/usr/bin/env python3
"""Tuning sweep on the GPU box: evaluation-kernel time for every (lanes per candidate, waves per SIMD)."""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from frenetix_motion_planner_amd import synthetic  # noqa: E402
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls  # noqa: E402


def run(label, steps=60, **kw):
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
    out = {}
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        for variant in (1, 2):
          for G in (1, 2, 4, 8):
            for w in (2, 3, 4):
                eng.set_tuning(G, w, variant)
                eng.upload(inp)
                for _ in range(5):
                    eng.evaluate(); eng.finish()
                ts = []
                for _ in range(steps):
                    eng.evaluate(); eng.finish()
                    ts.append(eng.last_eval_kernel_ms)
                out[f"{'gen' if variant == 1 else 'grid'}_G{G}_w{w}"] = round(float(np.median(ts)) * 1e3, 1)
    print(label, inp.n_candidates, json.dumps(out))


if __name__ == "__main__":
    run("config2_modeB", ref_kind="arc", v0=10.0, grid=(19, 51, 51))
    run("config2_modeA", ref_kind="arc", v0=10.0, grid=(19, 51, 51), write_bundle=False, write_costmap=False)
    run("config3_modeB", ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20)
    run("config3_modeA", ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, write_bundle=False, write_costmap=False)
    run("1M_modeA_obs20", steps=10, ref_kind="arc", v0=10.0, grid=(19, 230, 229), n_obstacles=20, write_bundle=False, write_costmap=False)
    run("1M_modeA", steps=10, ref_kind="arc", v0=10.0, grid=(19, 230, 229), write_bundle=False, write_costmap=False)
