#!/usr/bin/env python3
"""Per-step cost of the survivor exchange with a one-rank RCCL group (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from frenetix_motion_planner_amd import synthetic
from frenetix_motion_planner_amd.distributed import ShardedEvaluator
from frenetix_motion_planner_amd.engine import FrenetEngine
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29517"
torch.cuda.set_device(0); dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 51, 51))
with FrenetEngine(max_candidates=inp.n_candidates + 64) as eng:
    for mode in ("lib", "lib-direct", "torch"):
        os.environ["FX_EXCHANGE"] = "lib" if mode.startswith("lib") else mode
        ev = ShardedEvaluator(eng, k=int(os.environ.get('FXK', '1')), force_exchange=True); eng.upload(inp)
        eng.set_exchange_mode(1 if mode == "lib-direct" else 0)   # receive straight in the pinned block / device buffer + publication kernel
        ref = None
        for _ in range(20): ev.step_enqueued()
        t = []
        for _ in range(400):
            t0 = time.perf_counter(); ev.step_enqueued(); t.append(time.perf_counter() - t0)
        r = ev.step_enqueued()
        print("step + 1-rank RCCL all-gather of the winner, exchange driven by %s (lib_exchange=%s): p50 %.1f us; winner %d cost %.6f" %
              (mode, ev.lib_exchange, np.median(t) * 1e6, r["global_best_index"], r["global_best_cost"]), flush=True)
        eng.set_exchange_mode(0)
        if ev.lib_exchange: eng.comm_destroy()
        eng.set_winner_buffer(0)
    ev2 = ShardedEvaluator(eng, k=8); eng.upload(inp)
    t = []
    for _ in range(20): ev2.step_enqueued()
    for _ in range(400):
        t0 = time.perf_counter(); ev2.step_enqueued(); t.append(time.perf_counter() - t0)
    print("plain step p50 %.1f us" % (np.median(t) * 1e6), flush=True)
dist.destroy_process_group()
