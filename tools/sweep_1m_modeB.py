import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine
inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 230, 229))
with FrenetEngine(max_candidates=inp.n_candidates + 64) as eng:
    eng.set_timing("kernel")
    for tn in ((0,0,0,0),(1,2,2,256),(1,3,2,256),(1,4,2,256),(1,2,2,128),(1,4,2,128)):
        out = {}
        for mode in (1, 2):
            eng.set_store_mode(mode); eng.set_tuning(*tn); eng.upload(inp)
            for _ in range(3): eng.evaluate(); eng.finish()
            ts = []
            for _ in range(15):
                eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
            out["wb" if mode == 1 else "wt"] = round(float(np.median(ts)) * 1e3, 1)
        print(tn, out, flush=True)
