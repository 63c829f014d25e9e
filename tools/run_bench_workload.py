#!/usr/bin/env python3
"""One of bench.py's workloads, a few resident steps with the automatic tuning (driver for rocprofv3 passes; the inputs are built
by bench.make_workload, so the kernels are exactly those of the bench line).
usage: run_bench_workload.py <config3|config2|north_star_obstacles|north_star_bundle|north_star_bundle_obstacles> [steps]"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from frenetix_motion_planner_amd.engine import FrenetEngine
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
args = types.SimpleNamespace(workload="config3" if name == "config3" else "config2", select_only=False)
if name in ("config3", "config2"):
    inp = bench.make_workload(args, 1)
elif name == "north_star_obstacles":
    inp = bench.make_workload(args, 1, grid=bench.NORTH_STAR_GRID, n_obst=20, select_only=True)
elif name == "north_star_bundle_obstacles":
    inp = bench.make_workload(args, 1, grid=bench.NORTH_STAR_GRID, n_obst=20, select_only=False)
else:
    inp = bench.make_workload(args, 1, grid=bench.NORTH_STAR_GRID, n_obst=0, select_only=False)
with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_ref_knots=1024, max_obstacles=32, max_pred_steps=64) as eng:
    eng.upload(inp)
    for _ in range(steps):
        r = eng.step_raw()[0]
    print(name, inp.n_candidates, int(r.best_index), int(r.n_collisions), eng.step_info())
