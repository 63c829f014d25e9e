#!/usr/bin/env python3
"""Wall-clock timeline of the obstacle kernel (probe build -DFX_PROBE=3: every stamp of the obstacle kernel is the 100 MHz wall
clock): per wave, relative to the first wave's entry: entry (dispatch ramp) | 1 problem + list + flags known | 2 tables in LDS, rows
there | 4 visits done | 15 end.   hipcc ... -DFX_PROBE=3 -shared -o tools/probe_build/libfxplan_p3.so fx_kernels.hip fx_api.hip"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FXPLAN_SO"] = os.path.join(ROOT, "tools", "probe_build", os.environ.get("FX_PROBE_LIB", "libfxplan_p3.so"))
sys.path.insert(0, ROOT)
import numpy as np
from frenetix_motion_planner_amd import synthetic, _lib
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

SL = 16
lib = _lib.lib()
lib.fx_probe_read_obs.argtypes = [C.c_void_p, C.c_size_t]
lib.fx_probe_read.argtypes = [C.c_void_p, C.c_size_t]
inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, n_pred=30, lead_gap=25.0, hull_builder=build_obstacle_hulls)
with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=30) as eng:
    eng.set_timing("kernel")
    eng.upload(inp)
    for _ in range(30):
        eng.evaluate(); eng.finish()
    n = 1 << 16
    for rep in range(3):
        eng.evaluate(); eng.finish()
        buf = np.zeros(n * SL, dtype=np.uint64); assert lib.fx_probe_read_obs(buf.ctypes.data, buf.size) == 0
        wb = np.zeros(n * SL, dtype=np.uint64); assert lib.fx_probe_read(wb.ctypes.data, wb.size) == 0
        st = buf.reshape(n, SL).astype(np.int64); wk = wb.reshape(n, SL).astype(np.int64)
        live = st[:, 0] > 0
        st = st[live]
        t0 = st[:, 0].min()
        walk_end = wk[wk[:, 15] > 0, 15].max() if (wk[:, 15] > 0).any() else t0
        ev, _ = eng.kernel_times(1); ob = eng.obstacle_kernel_times(1)
        print(f"rep {rep}: walk {ev[-1] * 1e3:.1f} us, obstacle kernel {ob[-1] * 1e3:.1f} us (HIP events); walk's last wave end -> obstacle kernel's first entry "
              f"{(t0 - walk_end) * 1e-2:.2f} us; waves stamped {len(st)} ({eng.step_info()['obstacle_items']} items)")
        def rel(col, sel=None):
            v = st[:, col] if sel is None else st[sel, col]
            v = v[v > 0]
            return "-" if not len(v) else f"min {(v.min() - t0) * 1e-2:6.2f}  p50 {(np.median(v) - t0) * 1e-2:6.2f}  p90 {(np.percentile(v, 90) - t0) * 1e-2:6.2f}  max {(v.max() - t0) * 1e-2:6.2f}  (n {len(v)})"
        for col, name in ((0, "entry"), (1, "problem, list, flags"), (2, "tables + rows there"), (4, "visits done"), (15, "end")):
            print(f"   {name:24s} {rel(col)}")
        worked = st[:, 2] > 0
        d = st[worked]
        print(f"   per working wave (us): entry->1 {np.median(d[:,1]-d[:,0])*1e-2:.2f}  1->2 {np.median(d[:,2]-d[:,1])*1e-2:.2f}  2->4 (visits) {np.median(d[:,4]-d[:,2])*1e-2:.2f} "
              f"[p90 {np.percentile(d[:,4]-d[:,2],90)*1e-2:.2f}]  4->end {np.median(d[d[:,15]>0,15]-d[d[:,15]>0,4])*1e-2:.2f}")
