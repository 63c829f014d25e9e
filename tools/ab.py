#!/usr/bin/env python3
"""Same-box A/B of library builds on the workloads the round works on.  Each build runs in its own process (FXPLAN_SO), builds
alternate so that clock drift between the arms shows.

usage: ab.py [--builds cur,prev,...] [--rounds 2] [workload ...]
  builds:    cur = the in-tree library, anything else = tools/probe_build/libfxplan_<name>.so
  workloads: m1o  (1 005 100 x 31 x 20 obstacles, select only: the north-star kernel)
             c5   (config 5: --agents agents x 103 428 x 51 x 20 obstacles in one batched launch)
             c3   (config 3: walk + obstacle kernel + selection; kernel times and wall step)
             c2B / m1B (bundle materialised, no obstacles)
With --child the process measures and prints one JSON line (internal)."""
import json, os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(which, agents):
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    out = {}
    for name in which:
        if name == "c5":
            inp = synthetic.stress_agents(agents, grid=(39, 51, 51), hull_builder=build_obstacle_hulls)
            C = sum(a.n_candidates for a in inp)
            eng = FrenetEngine(max_candidates=C + 64 * agents, max_steps=inp[0].N, max_ref_knots=1024, max_obstacles=32,
                               max_pred_steps=64, max_agents=agents)
        else:
            W = dict(m1o=dict(grid=(19, 230, 229), n_obstacles=20, lead_gap=25.0, write_bundle=False, write_costmap=False),
                     c3=dict(grid=(19, 51, 51), n_obstacles=20, lead_gap=25.0),
                     c2B=dict(grid=(19, 51, 51)), m1B=dict(grid=(19, 230, 229)))[name]
            inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=10.0, n_pred=30, **W)
            eng = FrenetEngine(max_candidates=inp.n_candidates + 64, max_steps=inp.N)
        with eng:
            eng.set_timing("kernel")
            if os.environ.get("FX_AB_TUNING"):   # "G,wpe,variant,block,mapping" (0 = automatic)
                eng.set_tuning(*[int(x) for x in os.environ["FX_AB_TUNING"].split(",")])
            if os.environ.get("FX_AB_OBST"):     # "stage,steps_per_item" (fx_set_obstacle_stage)
                eng.set_obstacle_stage(*[int(x) for x in os.environ["FX_AB_OBST"].split(",")])
            eng.upload(inp)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.5:   # warm clocks
                eng.evaluate(); eng.finish()
            n = 40 if name in ("c3", "c2B") else 12
            ev, ob, wall = [], [], []
            for _ in range(n):
                ts = time.perf_counter()
                eng.evaluate(); r = eng.finish()
                wall.append(time.perf_counter() - ts)
                ev.append(eng.last_eval_kernel_ms); ob.append(eng.last_obstacle_kernel_ms)
            info = eng.step_info()
            r0 = r[0]
            out[name] = dict(eval_us=round(float(np.median(ev)) * 1e3, 1), obst_us=round(float(np.median(ob)) * 1e3, 1),
                             wall_us=round(float(np.median(wall)) * 1e6, 1), winner=int(r0["best_index"]), coll=int(r0["n_collisions"]),
                             cost=float(r0["best_cost"]), G=info["lanes_per_candidate"], wpe=info["waves_per_simd"], block=info.get("block"),
                             lds=info.get("lds_bytes"))
    print("ABJSON " + json.dumps(out), flush=True)


def main():
    a = sys.argv[1:]
    if a and a[0] == "--child":
        return child(a[2:], int(a[1]))
    builds, rounds, agents, which = ["cur", "prev"], 2, 32, []
    i = 0
    while i < len(a):
        if a[i] == "--builds": builds = a[i + 1].split(","); i += 2
        elif a[i] == "--rounds": rounds = int(a[i + 1]); i += 2
        elif a[i] == "--agents": agents = int(a[i + 1]); i += 2
        else: which.append(a[i]); i += 1
    which = which or ["m1o", "c5", "c3"]
    res = {b: [] for b in builds}
    for r in range(rounds):
        for b in builds:
            env = dict(os.environ)
            if b != "cur":
                env["FXPLAN_SO"] = os.path.join(ROOT, "tools", "probe_build", f"libfxplan_{b}.so")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(agents)] + which, env=env, capture_output=True,
                               text=True, timeout=900)
            line = [l for l in p.stdout.split("\n") if l.startswith("ABJSON ")]
            if not line:
                print(f"[{b}] FAILED rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-3000:]}", flush=True)
                continue
            d = json.loads(line[0][7:])
            res[b].append(d)
            print(f"[{b} #{r}] " + "  ".join(f"{k}: eval {v['eval_us']} obst {v['obst_us']} wall {v['wall_us']} (G{v['G']} w{v['wpe']} b{v['block']} "
                                             f"lds {v['lds']}; win {v['winner']} coll {v['coll']})" for k, v in d.items()), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
