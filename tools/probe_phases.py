#!/usr/bin/env python3
"""Phase timing inside the evaluation kernel from per-wave cycle stamps (probe build, -DFX_PROBE):
    hipcc ... -DFX_PROBE=9 -shared -o tools/probe_build/libfxplan_p9.so fx_kernels.hip fx_api.hip
Stamps: 0 kernel entry (wall clock, 100 MHz) | 1 after phase 1 | 2 rows done | 3 walk start | 4 walk end |
5 parts combined | 6 flags | 7 costs | 8 histogram | 9 wave arg-min | 10 partial + counters | 11 stores drained |
12 ticket | 13 tail: winner known | 14 tail: collisions counted | 15 end (wall clock; behind the tail's publication where it runs)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FXPLAN_SO"] = os.path.join(ROOT, "tools", "probe_build", "libfxplan_p9.so")
sys.path.insert(0, ROOT)
import numpy as np
from frenetix_motion_planner_amd import synthetic, _lib
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

SL = 16
def run(label, G, mp, fused, wpe=2, blk=256, package=False, stage=0, **kw):
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, hull_builder=build_obstacle_hulls, **kw)
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N) as eng:
        eng.set_timing("kernel"); eng.set_fused_selection(fused); eng.set_tuning(G, wpe, 2, blk, mp); eng.set_package(package)
        eng.set_obstacle_stage(stage); eng.upload(inp)
        for _ in range(5): eng.evaluate(); eng.finish()
        ms = eng.last_eval_kernel_ms
        n_waves = -(-inp.n_candidates // (blk // G)) * (blk // 64)
        buf = np.zeros(n_waves * SL, dtype=np.uint64)
        lib = _lib.lib()
        lib.fx_probe_read.argtypes = [C.c_void_p, C.c_size_t]
        assert lib.fx_probe_read(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(n_waves, SL).astype(np.int64)
    t0 = st[:, 0].min()
    names = ["rows", "lat_setup", "walk", "combine", "flags", "costs", "hist", "argmin", "partial", "drain", "ticket", "tail_winner", "tail_count"]
    print("   step info:", {k: v for k, v in eng.step_info().items() if k in ("lanes_per_candidate", "block", "fused_selection", "tail", "blocks")}) if False else None
    def phase(i, q):   # only the first wave of a workgroup reaches the stamps behind the reductions, and only on fused steps
        ok = (st[:, i + 1] > 0) & (st[:, i + 2] > 0)
        return int(np.percentile(st[ok, i + 2] - st[ok, i + 1], q)) if ok.any() else None
    done = st[:, 15] > 0
    wall = (st[done, 15].max() - t0) * 10e-3 if done.any() else float("nan")  # us (100 MHz)
    print(label, f"G{G}m{mp} fused={fused} kernel {ms*1e3:.1f} us; first entry -> last end {wall:.1f} us; entry spread "
          f"{(st[:,0].max()-t0)*10e-3:.1f} us")
    print("   median cycles per phase:", {n: phase(i, 50) for i, n in enumerate(names)})
    print("   p95    cycles per phase:", {n: phase(i, 95) for i, n in enumerate(names)})
    print("   wave life (stamp 1 -> stamp 9, cycles): median", int(np.median(st[:, 9] - st[:, 1])), "max", int((st[:, 9] - st[:, 1]).max()))
    # the publishing wave: cycles from its ticket to the end of the tail, and when the first wave / the last ticket happened
    pub = np.nonzero(st[:, 13] > 0)[0]
    if len(pub):
        w = pub[0]
        print("   tail (cycles): ticket -> winner", int(st[w, 13] - st[w, 12]), " -> counted", int(st[w, 14] - st[w, 13]),
              "; kernel entry -> this wave's entry", f"{(st[w, 0] - t0) * 10e-3:.1f} us, -> its end {(st[w, 15] - t0) * 10e-3:.1f} us")
    first = st[np.argmin(st[:, 0])]
    print("   first wave: stamps 1..12 relative to stamp 1 (cycles):", [int(first[k] - first[1]) if first[k] > 0 else None for k in range(1, 13)])

RUNS = dict(
    c2B=lambda: run("50k_B", 2, 2, True, grid=(19, 51, 51)),
    c2B_sel=lambda: run("50k_B", 2, 2, False, grid=(19, 51, 51)),
    c2A=lambda: run("50k_A", 2, 2, True, grid=(19, 51, 51), write_bundle=False, write_costmap=False),
    c2B_g1=lambda: run("50k_B", 1, 0, True, grid=(19, 51, 51)),
    m1A=lambda: run("1M_A", 1, 0, True, grid=(19, 230, 229), write_bundle=False, write_costmap=False),
    c3B=lambda: run("c3_B", 2, 2, True, grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
    c3B_g4=lambda: run("c3_B", 4, 2, True, grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
    c3B_g4w3=lambda: run("c3_B", 4, 2, True, wpe=3, grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
    c3A=lambda: run("c3_A", 2, 2, True, grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
    c1_g8=lambda: run("c1", 8, 0, True, blk=128, level=2, n_obstacles=5),
    c1_g32=lambda: run("c1", 32, 0, True, blk=128, level=2, n_obstacles=5),
    c1_g32_64=lambda: run("c1", 32, 0, True, blk=64, level=2, n_obstacles=5),
    c1_g4=lambda: run("c1", 4, 2, True, blk=256, level=2, n_obstacles=5),
    c1_g8_sel=lambda: run("c1", 8, 0, False, blk=128, level=2, n_obstacles=5),
    c1_tail=lambda: run("c1_tail", 32, 0, True, blk=128, package=True, stage=1, level=2, n_obstacles=5, draw_traj_set=True, kinematic_debug=True),
    c1_tail16=lambda: run("c1_tail", 16, 0, True, blk=128, package=True, stage=1, level=2, n_obstacles=5, draw_traj_set=True, kinematic_debug=True),
    c1_notail=lambda: run("c1_notail", 32, 0, False, blk=128, package=True, stage=1, level=2, n_obstacles=5, draw_traj_set=True, kinematic_debug=True),
    c2A_g4=lambda: run("50k_A", 4, 2, True, grid=(19, 51, 51), write_bundle=False, write_costmap=False),
)
for name in sys.argv[1:] or list(RUNS):
    RUNS[name]()
