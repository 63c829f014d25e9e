#!/usr/bin/env python3
"""What the reference's C++ adapter (ReactivePlannerCpp.plan(), reactive_planner_cpp.py:292-441) spends INSIDE this package's
`frenetix` module per plan step: generate_trajectories + evaluate_all_current_functions + get_sorted_trajectories + the
feasible / infeasible split the adapter makes over the returned objects (:353-358).  Recorded call trace (tests/golden), HIP engine."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from frenetix_motion_planner_amd import frenetix_compat as fc
from tests.dropin.trace_recorder import CLASSES, replay

ns = {n: getattr(fc, n) for names in CLASSES.values() for n in names}
ns["compute_initial_state"] = fc.compute_initial_state
objs, captured, expected = replay(os.path.join(ROOT, "tests", "golden", "cpp_adapter_trace.json"), ns, None)
h = next(o for o in objs.values() if isinstance(o, fc.TrajectoryHandler))
m, low = h._matrix.copy(), h._low_vel
seg = {k: [] for k in ("generate", "evaluate", "sorted", "split", "total")}
for it in range(130):
    t0 = time.perf_counter()
    h.reset_Trajectories() if hasattr(h, "reset_Trajectories") else None
    h.generate_trajectories(m, low)
    t1 = time.perf_counter()
    h.evaluate_all_current_functions_concurrent(True)
    t2 = time.perf_counter()
    srt = h.get_sorted_trajectories()
    t3 = time.perf_counter()
    feas = [t for t in srt if t.feasible]
    infeas = [t for t in srt if not t.feasible]
    best = feas[0] if feas else None
    _ = best.cost if best is not None else None
    t4 = time.perf_counter()
    if it >= 30:
        for k, v in zip(seg, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)):
            seg[k].append(v)
print(f"{len(srt)} trajectories returned, {len(feas)} feasible; grid recognised: {fc.product_grid_of(m) is not None}")
for k, v in seg.items():
    print(f"  {k:9s} p50 {np.median(v) * 1e6:9.1f} us")
h.engine.close()
