#!/usr/bin/env python3
"""How the large-grid kernel's time grows with the number of workgroups (is the last, partly filled round of workgroups a visible
tail?) and how the alternatives compare at the north-star size: lanes per candidate, workgroup size.
usage: ns_tail.py [select|bundle|both] ; FX_NS_ND="200,210,...": lateral sample counts (19 x 230 x nD candidates)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

mode = sys.argv[1] if len(sys.argv) > 1 else "select"
nds = [int(v) for v in os.environ.get("FX_NS_ND", "200,210,220,224,225,226,228,229,232,240").split(",")]
tunings = [tuple(int(x) for x in t.split(":")) for t in os.environ.get("FX_NS_TUNE", "0:0:0").split(",")]   # lanes:wpe:block


def run(eng, inp, n=12):
    eng.upload(inp)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        eng.step_raw()
    for _ in range(n):
        eng.step_raw()
    ev, _ = eng.kernel_times(n)
    return float(np.median(ev)) * 1e3, float(np.min(ev)) * 1e3


for select_only in ([True] if mode == "select" else [False] if mode == "bundle" else [True, False]):
    with FrenetEngine(max_candidates=19 * 230 * 260 + 64, max_steps=30, max_ref_knots=1024, max_obstacles=32, max_pred_steps=64) as eng:
        eng.set_timing("kernel", every=1)
        for lanes, wpe, block in tunings:
            eng.set_tuning(lanes, wpe, 0, block, 0)
            for nd in nds:
                inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 230, nd - 1), n_obstacles=20, n_pred=30, lead_gap=25.0,
                                            write_bundle=not select_only, write_costmap=not select_only, draw_traj_set=False,
                                            kinematic_debug=False, hull_builder=build_obstacle_hulls)
                try:
                    med, mn = run(eng, inp)
                except ValueError as e:
                    print("not applicable", lanes, wpe, block, nd, e, flush=True)
                    continue
                info = eng.step_info()
                C = inp.n_candidates
                print(f"{'select' if select_only else 'bundle'} tune {lanes}:{wpe}:{block} nD={nd} C={C} blocks={info['blocks']} block={info['block']} "
                      f"G={info['lanes_per_candidate']} wpe={info['waves_per_simd']} rounds={info['blocks'] * info['block'] / 64 / (1024 * info['waves_per_simd']):.2f} "
                      f"kernel {med:.1f} us (min {mn:.1f})  {med * 1e3 / C:.4f} ns/cand", flush=True)
