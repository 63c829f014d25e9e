import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("FX_PROBE_LIB", "")
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
sys.argv = ["x"]
import importlib.util

def kernel_ms(inp, tag):
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_ref_knots=4096) as eng:
        eng.set_timing("kernel"); eng.upload(inp)
        for _ in range(30): eng.evaluate(); eng.finish()
        ts = []
        for _ in range(30):
            eng.evaluate(); eng.finish(); ts.append(eng.last_eval_kernel_ms)
        info = eng.step_info()
        print(f"{tag:50s} C={inp.n_candidates} kernel {np.median(ts)*1e3:6.1f} us  G{info['lanes_per_candidate']} grid={info['grid_kernel']} tail={info['tail']} lds={info['lds_bytes']}", flush=True)
for dbg in (False, True):
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, hull_builder=build_obstacle_hulls, as_matrix=True, level=2, cpp_style=True, n_obstacles=5,
                                draw_traj_set=dbg, kinematic_debug=dbg)
    kernel_ms(inp, f"synthetic arc matrix dbg={dbg}")
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, hull_builder=build_obstacle_hulls, as_matrix=True, level=2, cpp_style=True, n_obstacles=5,
                                draw_traj_set=dbg, kinematic_debug=dbg, knot_jitter=0.3)
    kernel_ms(inp, f"synthetic arc matrix jittered knots dbg={dbg}")
    inp = synthetic.make_inputs(ref_kind="arc", v0=5.6, v_des=8.0, hull_builder=build_obstacle_hulls, as_matrix=True, level=2, cpp_style=True, n_obstacles=5,
                                draw_traj_set=dbg, kinematic_debug=dbg)
    kernel_ms(inp, f"synthetic arc matrix v0=5.6 dbg={dbg}")
# the ZAM inputs

src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_timeline.py")).read().split("def run(")[0].split("SL = 16")[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exec(src)
z = zam_inputs(True)
kernel_ms(z, "ZAM 800 matrix (debug flags)")
z2 = copy.copy(z); z2.draw_traj_set = False; z2.kinematic_debug = False; z2.__post_init__()
kernel_ms(z2, "ZAM 800 matrix (production flags)")
z3 = copy.copy(z); z3.obstacles = synthetic.make_inputs(ref_kind="arc", v0=10.0).obstacles; z3.collision = False; z3.__post_init__()
try: kernel_ms(z3, "ZAM 800 matrix, no obstacles")
except Exception as e: print("no-obstacle variant failed", e)
print("weights", z.cost_weights, "low_vel", z.low_vel_mode, "x0", z.x0_lon, z.x0_lat, "M", len(z.coordinate_system.ref_pos))
