#!/usr/bin/env python3
"""Config 3 with the obstacle stage FUSED into the walk under other decompositions (lanes per candidate x workgroup size x waves per
SIMD x part mapping) against the split step (walk + obstacle kernel): evaluation kernel time and the step's wall time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0)
combos = [(2, 0, 0, 0, 0)] + [(1, g, w, b, m) for g in (2, 4, 8) for w in (2, 3) for b in (256, 128) for m in (0,)] + [(1, 1, 3, 256, 0), (1, 1, 2, 256, 0)]
with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
    base = None
    for stage, g, w, b, m in combos:
        try:
            eng.set_tuning(g, w, 0, b, m)
            eng.set_obstacle_stage(stage, 3 if stage == 2 else 0)
            eng.upload(inp)
        except ValueError as e:
            print(stage, g, w, b, m, "not applicable:", str(e)[:80], flush=True)
            continue
        eng.set_timing("kernel")
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            eng.evaluate(); eng.finish()
        n = 60
        for _ in range(n):
            eng.evaluate(); r = eng.finish()[0]
        ev, st = eng.kernel_times(n); ob = eng.obstacle_kernel_times(n)
        eng.set_timing("off")
        tw = []
        for _ in range(n):
            a = time.perf_counter(); eng.evaluate(); eng.finish(); tw.append(time.perf_counter() - a)
        info = eng.step_info()
        if base is None:
            base = (r["best_index"], r["n_collisions"])
        print(f"stage {'split' if stage == 2 else 'fused'} lanes {g} wpe {w} block {b}: ran G={info['lanes_per_candidate']} wpe={info['waves_per_simd']} block={info['block']} ws={info['wave_split']} "
              f"blocks={info['blocks']} obstacle_kernel={info['obstacle_kernel']}  evaluation {np.median(ev) * 1e3:6.1f} us  obstacle kernel {np.median(ob) * 1e3:5.1f} us  "
              f"step wall p50 {np.median(tw) * 1e6:6.1f} us  same result {(r['best_index'], r['n_collisions']) == base}", flush=True)
