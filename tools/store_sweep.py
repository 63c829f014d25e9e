#!/usr/bin/env python3
"""Write-back vs write-through plane stores by bundle size (Mode B, no obstacles): evaluation-kernel time."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine

for nv, nd in ((26, 25), (51, 51), (62, 61), (72, 71), (88, 87), (102, 102), (144, 144), (230, 229)):
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, nv, nd))
    out = {}
    with FrenetEngine(max_candidates=inp.n_candidates + 64) as eng:
        eng.set_timing("kernel")
        for mode in (1, 2):
            eng.set_store_mode(mode); eng.upload(inp)
            for _ in range(5): eng.evaluate(); eng.finish()
            ts = []
            for _ in range(40):
                eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
            out["wb" if mode == 1 else "wt"] = round(float(np.median(ts)) * 1e3, 1)
    mb = inp.n_candidates * 3472 / 1e6
    print(f"{inp.n_candidates:8d} candidates {mb:8.0f} MB  {json.dumps(out)}", flush=True)
