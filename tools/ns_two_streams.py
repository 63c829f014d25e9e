#!/usr/bin/env python3
"""Does a second, lower-priority launch fill the drain of the large-grid kernel?  The million-candidate step as ONE launch against
the same candidates split over two engine contexts (two streams): part A = the first nD_a lateral samples' worth at one lane per
candidate, part B = the rest, launched right behind it (optionally on a low-priority stream / with more lanes per candidate).
usage: ns_two_streams.py ; FX_SPLITS="207:23,..."  (nD of part A : nD of part B)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls


def inputs(nd):
    return synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 230, nd - 1), n_obstacles=20, n_pred=30, lead_gap=25.0,
                                 write_bundle=False, write_costmap=False, draw_traj_set=False, kinematic_debug=False,
                                 hull_builder=build_obstacle_hulls)


def engine(prio, cap):
    if prio:
        os.environ["FX_STREAM_PRIORITY"] = prio
    else:
        os.environ.pop("FX_STREAM_PRIORITY", None)
    return FrenetEngine(max_candidates=cap + 64, max_steps=30, max_ref_knots=1024, max_obstacles=32, max_pred_steps=64)


def wall(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        fn()
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return float(np.median(ts)) * 1e6


whole = inputs(230)
with engine(None, whole.n_candidates) as e:
    e.upload(whole)
    print(f"one launch: C={whole.n_candidates} wall {wall(lambda: (e.evaluate(), e.finish())):.1f} us", flush=True)
for spec in os.environ.get("FX_SPLITS", "207:23,200:30,190:40,215:15").split(","):
    na, nb = (int(v) for v in spec.split(":"))
    A, B = inputs(na), inputs(nb)
    for prio_b, lanes_b in ((None, 0), ("low", 0), ("low", 4), ("low", 8), (None, 4)):
        with engine("high" if prio_b else None, A.n_candidates) as ea, engine(prio_b, B.n_candidates) as eb:
            ea.upload(A)
            eb.set_tuning(lanes_b, 0, 0, 0, 0)
            eb.upload(B)
            wa = wall(lambda: (ea.evaluate(), ea.finish()))
            wb = wall(lambda: (eb.evaluate(), eb.finish()))
            both = wall(lambda: (ea.evaluate(), eb.evaluate(), ea.finish(), eb.finish()))
            print(f"split {na}:{nb} B prio={prio_b} lanes={lanes_b} G_b={eb.step_info()['lanes_per_candidate']}: A alone {wa:.1f} us, B alone {wb:.1f} us, "
                  f"together {both:.1f} us", flush=True)
