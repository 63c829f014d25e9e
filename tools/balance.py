#!/usr/bin/env python3
"""Does workgroup/CU balance matter?  Evaluation-kernel time per candidate for grids that fill 256 CUs evenly or not."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine

for label, kw in (("B", {}), ("A", dict(write_bundle=False, write_costmap=False))):
    for grid in ((16, 32, 63), (19, 51, 51), (16, 64, 63), (16, 80, 63), (16, 96, 63), (16, 128, 63)):
        inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=grid, **kw)
        C = inp.n_candidates
        out = {}
        with FrenetEngine(max_candidates=C + 64, max_steps=inp.N) as eng:
            eng.set_timing("kernel")
            for G, w, blk, mp in ((1, 2, 256, 0), (2, 2, 256, 2), (2, 2, 128, 2), (1, 2, 128, 0), (1, 2, 64, 0), (2, 2, 64, 1)):
                eng.set_tuning(G, w, 2, blk, mp); eng.upload(inp)
                for _ in range(3): eng.evaluate(); eng.finish()
                ts = []
                for _ in range(30):
                    eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
                t = float(np.median(ts)) * 1e3
                out[f"G{G}b{blk}m{mp}"] = (round(t, 1), round(t / C * 1e3, 3))
        print(label, C, "WG256:", round(C / 256, 1), json.dumps(out), flush=True)
