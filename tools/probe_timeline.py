#!/usr/bin/env python3
"""Wall-clock timeline of a planner-sized step inside the evaluation kernel (probe build -DFX_PROBE=2: every stamp is the 100 MHz
wall clock):  hipcc ... -DFX_PROBE=2 -shared -o tools/probe_build/libfxplan_p2.so fx_kernels.hip fx_api.hip
Prints, relative to the first wave's entry, when each phase boundary was passed by the first wave of the first workgroup, by the
median wave and by the workgroup that ran the tail.  Stamps: 0 entry | 1 phase 1 (tables in LDS) | 2 rows | 3 walk start | 4 walk end
| 5 parts combined | 6 flags | 7 costs | 8 histogram | 9 wave arg-min | 10 partial + counters | 11 stores drained | 12 ticket |
13 tail: winner | 14 tail: collisions counted | 15 end."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["FXPLAN_SO"] = os.path.join(ROOT, "tools", "probe_build", os.environ.get("FX_PROBE_LIB", "libfxplan_p2.so"))
sys.path.insert(0, ROOT)
import numpy as np
from frenetix_motion_planner_amd import synthetic, _lib
from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls

SL = 16


def zam_inputs(matrix800):
    """BASELINE config 1 as bench.py builds it: the ZAM_Tjunction ego's level-2 step, optionally as the C++-style 800-row matrix"""
    import copy
    from frenetix_motion_planner_amd import commonroad_xml as crx
    from frenetix_motion_planner_amd.frenet_interface import FrenetPlannerInterfaceHip
    from frenetix_motion_planner_amd.sampling import generate_sampling_matrix
    sc = crx.read_scenario_json(os.path.join(ROOT, "tests", "golden", "ZAM_Tjunction-1_42_T-1.scenario.json"))
    itf = FrenetPlannerInterfaceHip(60000, sc, sc.planning_problems[60000], device=0)
    itf.update_planner(None, sc.ground_truth_predictions(0, 30))
    inp = itf.begin_step()
    if not matrix800:
        return inp
    t = np.union1d(inp.t_samp, [inp.N * inp.dt]); v = np.union1d(inp.v_samp, [inp.x0_lon[1]]); d = np.union1d(inp.d_samp, [inp.x0_lat[0]])
    m = copy.copy(inp)
    m.t_samp = m.v_samp = m.d_samp = None
    m.sampling_matrix = generate_sampling_matrix(
        t0_range=0.0, t1_range=t, s0_range=inp.x0_lon[0], ss0_range=inp.x0_lon[1], sss0_range=inp.x0_lon[2], ss1_range=v, sss1_range=0.0,
        d0_range=inp.x0_lat[0], dd0_range=inp.x0_lat[1], ddd0_range=inp.x0_lat[2], d1_range=d, dd1_range=0.0, ddd1_range=0.0)
    m.__post_init__()
    return m


def run(label, G=0, blk=0, fused=True, package=True, stage=1, matrix=False, zam=None, **kw):
    inp = zam_inputs(zam == 800) if zam else synthetic.make_inputs(ref_kind="arc", v0=10.0, hull_builder=build_obstacle_hulls, as_matrix=matrix, **kw)
    lib = _lib.lib()
    lib.fx_probe_read.argtypes = [C.c_void_p, C.c_size_t]
    with FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N, max_ref_knots=4096) as eng:
        eng.set_timing("kernel"); eng.set_fused_selection(fused); eng.set_tuning(G, 0, 0, blk, 0); eng.set_package(package)
        eng.set_obstacle_stage(stage); eng.upload(inp)
        for _ in range(20): eng.evaluate(); eng.finish()
        n = 1 << 14
        zero = np.zeros(n * SL, dtype=np.uint64)
        rows = []
        for rep in range(5):
            eng.evaluate(); eng.finish()
            buf = np.zeros(n * SL, dtype=np.uint64)
            assert lib.fx_probe_read(buf.ctypes.data, buf.size) == 0
            rows.append(buf.reshape(n, SL).astype(np.int64))
        ms = eng.last_eval_kernel_ms
        info = eng.step_info()
    st = rows[-1]
    live = st[:, 0] > 0
    # stamps of earlier runs stay in the table: keep the waves whose entry stamp belongs to the last run
    t_last = st[live, 0].max()
    cur = live & (st[:, 0] > t_last - 5000)
    st = st[cur]
    t0 = st[:, 0].min()
    st = np.where(st >= st[:, :1], st, 0)   # stamps older than the wave's own entry are leftovers of earlier launches
    us = lambda x: (x - t0) * 0.01
    print(f"{label}: {inp.n_candidates} candidates, G{info['lanes_per_candidate']} block {info['block']} blocks {info['blocks']} grid={info['grid_kernel']} "
          f"fused={info['fused_selection']} tail={info['tail']}; kernel {ms*1e3:.1f} us; waves {len(st)}; entry spread {us(st[:,0].max()):.2f} us")
    first = st[np.argmin(st[:, 0])]
    def fmt(v): return " ".join(f"{k}:{us(x):5.2f}" if x >= t0 else f"{k}:  -  " for k, x in enumerate(v))
    print("   first wave      ", fmt(first))
    lead = st[(st[:, 12] >= t0)]
    if len(lead):
        print("   median lead wave", fmt(np.array([np.median(c[c >= t0]) if (c >= t0).any() else 0 for c in lead.T])))
        print("   last  lead wave ", fmt(lead[np.argmax(lead[:, 12])]))
    tail = st[(st[:, 14] >= t0)]
    if len(tail):
        print("   tail waves      ", fmt(np.max(tail, axis=0)))
    print(f"   last end {us(st[:, 15].max()):.2f} us")


RUNS = dict(
    c1=lambda: run("config-1 sized, tail", level=2, n_obstacles=5, draw_traj_set=True, kinematic_debug=True),
    c1_sel=lambda: run("config-1 sized, selection kernel", fused=False, level=2, n_obstacles=5, draw_traj_set=True, kinematic_debug=True),
    c1_noobs=lambda: run("config-1 sized, no obstacles", level=2, draw_traj_set=True, kinematic_debug=True),
    c1_g16=lambda: run("config-1 sized, G16", G=16, level=2, n_obstacles=5, draw_traj_set=True, kinematic_debug=True),
    c1_g8=lambda: run("config-1 sized, G8", G=8, level=2, n_obstacles=5, draw_traj_set=True, kinematic_debug=True),
    cpp800=lambda: run("800-row sampling matrix (generic kernel)", matrix=True, level=2, cpp_style=True, n_obstacles=5),
    c4agent=lambda: run("10 488 candidates", grid=(19, 23, 24), n_obstacles=9),
    zam630=lambda: run("BASELINE config 1 (ZAM_Tjunction ego, level 2)", zam=630, package=False),
    zam800=lambda: run("BASELINE config 1 as the 800-row C x 13 matrix", zam=800, package=False),
    c4agent_sel=lambda: run("10 488 candidates, selection kernel", fused=False, grid=(19, 23, 24), n_obstacles=9),
    c3=lambda: run("config 3 (walk of the split step)", grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0, package=False),
    c2=lambda: run("config 2 (bundle, no obstacles)", grid=(19, 51, 51), package=False),
    c2g4=lambda: run("config 2, four lanes per candidate", G=4, grid=(19, 51, 51), package=False),
    c2g8=lambda: run("config 2, eight lanes per candidate", G=8, grid=(19, 51, 51), package=False),
    c2g1=lambda: run("config 2, one lane per candidate", G=1, grid=(19, 51, 51), package=False),
    l4=lambda: run("level 4 (11 220 candidates), 5 obstacles", level=4, n_obstacles=5),
    l4_sel=lambda: run("level 4 (11 220 candidates), 5 obstacles, selection kernel", fused=False, level=4, n_obstacles=5),
)
for name in sys.argv[1:] or ["c1", "c1_sel", "c1_noobs", "cpp800"]:
    RUNS[name]()
