#!/usr/bin/env python3
"""Wall-clock timeline of the one-launch step (csrc/fx_step_kernel.h; probe build -DFX_PROBE=3 -> tools/probe_build/libfxplan_p3.so):
per wave, relative to the first entry: 6 entry | 7 walk done | 8 behind barrier 1 | 9 obstacle items done | 10 behind barrier 2 |
11 selection done."""
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
inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 51, 51), n_obstacles=20, n_pred=30, lead_gap=25.0, hull_builder=build_obstacle_hulls,
                            collision=os.environ.get("FX_NO_COLLISION") is None)
for ch in [int(a) for a in sys.argv[1:]] or [5]:
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=30) as eng:
        eng.set_timing("kernel"); eng.set_step_kernel(2, ch)
        eng.upload(inp)
        for _ in range(10):
            eng.evaluate(); eng.finish()
        n = 1 << 16
        buf = np.zeros(n * SL, dtype=np.uint64); assert lib.fx_probe_read_obs(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(n, SL).astype(np.int64)
        st = st[st[:, 6] > 0]
        t0 = st[:, 6].min()
        info = eng.step_info()
        print(f"steps/item {info['obstacle_steps_per_item']} waves {len(st)} (launch {info['obstacle_items']}), kernel {eng.last_kernel_ms * 1e3:.1f} us, walk blocks {info['blocks']}")
        walkers = np.arange(len(st)) < info["blocks"] * 4
        h = st[~walkers]
        if len(h):   # helpers run ONE item each: the item's own stamps (obstacle_item, FX_OSTAMP)
            d = lambda a, b: f"{np.median(h[:, b] - h[:, a]) * 1e-2:.2f} [p90 {np.percentile(h[:, b] - h[:, a], 90) * 1e-2:.2f}]"
            print(f"   first step of the item: prediction {d(2, 12)}  collision {d(12, 13)}  second step {d(13, 14)}")
            closers = h[h[:, 15] > h[:, 4]]
            print(f"   helper item (us): barrier->entry {d(8, 0)}  entry->list,flags {d(0, 1)}  ->rows,tables {d(1, 2)}  visits {d(2, 4)}  hand-off {d(4, 5)}"
                  f"  closing (n {len(closers)}) {np.median(closers[:, 15] - closers[:, 5]) * 1e-2 if len(closers) else 0:.2f}")
        for col, name in ((6, "entry"), (7, "walk done"), (8, "behind barrier 1"), (9, "items done"), (10, "behind barrier 2"), (11, "selection done")):
            for lab, sel in (("walkers", walkers), ("helpers", ~walkers)):
                v = st[sel, col]; v = v[v > 0]
                if len(v):
                    print(f"   {name:18s} {lab}: min {(v.min() - t0) * 1e-2:7.2f}  p50 {(np.median(v) - t0) * 1e-2:7.2f}  max {(v.max() - t0) * 1e-2:7.2f}  (n {len(v)})")
