#!/usr/bin/env python3
"""In-step timeline of the large-grid kernel with the fused obstacle stage (probe build -DFX_PROBE=2 -DFX_PROBE_STEP=15): core-clock
stamps at the entry of step 15, in front of its obstacle stage, behind the prediction loop, behind the collision part and at the entry
of step 16, per wave; printed as medians over the waves for launches that put 1 / 2 / 3 waves on a SIMD and for the 1 M grid.
  hipcc ... -DFX_PROBE=2 -DFX_PROBE_STEP=15 -shared -o tools/probe_build/libfxplan_ps.so fx_kernels.hip fx_api.hip"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FXPLAN_SO"] = os.path.join(ROOT, "tools", "probe_build", os.environ.get("FX_PROBE_LIB", "libfxplan_ps.so"))
sys.path.insert(0, ROOT)
import numpy as np
from frenetix_motion_planner_amd import synthetic, _lib
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

SL = 16
lib = _lib.lib()
lib.fx_probe_read_obs.argtypes = [C.c_void_p, C.c_size_t]
with FrenetEngine(max_candidates=19 * 230 * 232 + 64, max_steps=30, max_ref_knots=1024, max_obstacles=32, max_pred_steps=64) as eng:
    eng.set_timing("kernel")
    eng.set_tuning(1, 3, 0, 256, 0)
    for nd in (15, 30, 45, 230):
        inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 230, nd - 1), n_obstacles=20, n_pred=30, lead_gap=25.0,
                                    write_bundle=False, write_costmap=False, draw_traj_set=False, kinematic_debug=False,
                                    hull_builder=build_obstacle_hulls)
        eng.upload(inp)
        for _ in range(8):
            eng.evaluate(); eng.finish()
        n = 1 << 16
        buf = np.zeros(n * SL, dtype=np.uint64)
        assert lib.fx_probe_read_obs(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(n, SL).astype(np.int64)
        nw = (inp.n_candidates + 63) // 64
        st = st[:min(nw, n)]
        ok = (st[:, 1] > 0) & (st[:, 5] > st[:, 1])
        d = st[ok]
        seg = [("walk part", d[:, 2] - d[:, 1]), ("prediction", d[:, 3] - d[:, 2]), ("collision", d[:, 4] - d[:, 3]),
               ("to next step", d[:, 5] - d[:, 4]), ("whole step", d[:, 5] - d[:, 1])]
        print(f"nD={nd} C={inp.n_candidates} waves={nw} (probed {int(ok.sum())}) kernel {eng.last_eval_kernel_ms * 1e3:.1f} us; core-clock cycles of step 15, median [p10 .. p90]:")
        for name, v in seg:
            print(f"   {name:14s} {np.median(v):8.0f}  [{np.percentile(v, 10):.0f} .. {np.percentile(v, 90):.0f}]")
