#!/usr/bin/env python3
"""How often the collision broad phase of the fused obstacle stage lets something through (probe build -DFX_CULL_STATS ->
tools/probe_build/libfxplan_cstat.so): per (wave, step) with obstacle hulls -- share with survivors of the wave-level cull,
survivors per such step, share that reaches the exact axis test, obstacles per such step.
usage: cull_stats.py [m1o c5 c3A ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "probe_build", "libfxplan_cstat.so")
CSRC = os.path.join(ROOT, "frenetix-motion-planner_amd", "csrc")
srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [os.path.join(ROOT, "include", "fxplan.h")]
if not os.path.exists(PROBE) or os.path.getmtime(PROBE) < max(os.path.getmtime(f) for f in srcs):
    # the probe build of the CURRENT sources (hipcc is on the GPU box too; about a minute)
    import subprocess
    os.makedirs(os.path.dirname(PROBE), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                    "-Wno-unused-function", "-DFX_CULL_STATS", "-shared", "-o", PROBE, "fx_kernels.hip", "fx_api.hip"], cwd=CSRC, check=True)
os.environ["FXPLAN_SO"] = PROBE
sys.path.insert(0, ROOT)
import numpy as np
from frenetix_motion_planner_amd import synthetic, _lib
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

W = dict(m1o=dict(grid=(19, 230, 229), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
         c3A=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
         sparse=dict(grid=(19, 230, 229), n_obstacles=5, lead_gap=25.0, write_bundle=False, write_costmap=False))
lib = _lib.lib()
buf = np.zeros(16, np.uint64)
for name in sys.argv[1:] or ["m1o", "c5"]:
    if name == "c5":
        inp = synthetic.stress_agents(1, grid=(39, 51, 51), hull_builder=build_obstacle_hulls)[0]
    else:
        inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, n_pred=30, **W[name])
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_pred_steps=64) as eng:
        eng.set_tuning(1, 3, 2)
        eng.upload(inp)
        eng.evaluate(); r = eng.finish()[0]
        lib.fx_cull_stats_read(C.c_void_p(buf.ctypes.data), 1)
        lib.fx_probe_bound_set.argtypes = [C.c_double]
        lib.fx_probe_bound_set(float(r["best_cost"]))
        eng.evaluate(); r = eng.finish()[0]
        lib.fx_probe_bound_set(1e300)
        lib.fx_cull_stats_read(C.c_void_p(buf.ctypes.data), 1)
    n, nc, sc, ne, se = (int(x) for x in buf[:5])
    print(f"{name}: wave-steps {int(buf[6])}; every lane infeasible: {int(buf[5]) / max(int(buf[6]), 1):.3f} (lane-steps "
          f"{int(buf[7]) / max(64 * int(buf[6]), 1):.3f}); every lane infeasible or over the winner's cost {r['best_cost']:.3f}: "
          f"{int(buf[8]) / max(int(buf[6]), 1):.3f} (lane-steps {int(buf[9]) / max(64 * int(buf[6]), 1):.3f}); feasible {r['n_feasible']} of {inp.n_candidates}")
    print(f"{name}: {n} (wave, step) pairs with hulls; wave cull lets something through in {nc / max(n, 1):.3f} "
          f"({sc / max(nc, 1):.2f} obstacles each); exact test reached in {ne / max(n, 1):.3f} ({se / max(ne, 1):.2f} obstacles each); "
          f"collisions {r['n_collisions']}")
