#!/usr/bin/env python3
"""Where a plan step that takes its inputs from host buffers spends its time (configs 2 and 3): resident step, the
in-place update alone (host packing), update + step; p50 over 400 steps each."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from frenetix_motion_planner_amd.engine import FrenetEngine


class A:  # the arguments make_workload / perturbed_updates read
    select_only = False
    workload = "config3"


def p50(f, n=400, warm=40):
    for _ in range(warm): f()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts) * 1e6)


out = {}
for wl in ("config3", "config2"):
    A.workload = wl
    inp = bench.make_workload(A, 1)
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_ref_knots=1024, max_obstacles=32, max_pred_steps=64) as eng:
        eng.upload(inp)
        ups = bench.perturbed_updates(eng, inp, 8)
        j = [0]
        def upd_only():
            eng.update_state(ups[j[0] % 8]); j[0] += 1
        def upd_step():
            eng.update_step_raw(ups[j[0] % 8]); j[0] += 1
        rec = {"resident_us": p50(lambda: eng.step_raw())}
        rec["update_plus_step_us"] = p50(upd_step)
        eng.step_raw()
        # the update alone (no evaluation between updates: nothing is in flight, nothing is copied)
        rec["update_only_host_us"] = p50(upd_only)
        rec["full_upload_plus_step_us"] = p50(lambda: eng.plan_step(inp), n=200)
        out[wl] = rec
print(json.dumps(out, indent=1))
