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
    ev = ShardedEvaluator(eng, k=int(os.environ.get('FXK', '1')), force_exchange=True); eng.upload(inp)
    for _ in range(20): ev.step_enqueued()
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); ev.step_enqueued(); t.append(time.perf_counter() - t0)
    print("step with top-k + 1-rank RCCL all-gather + D2H: p50 %.1f us" % (np.median(t) * 1e6), flush=True)
    ev2 = ShardedEvaluator(eng, k=8); eng.upload(inp)
    t = []
    for _ in range(200):
        t0 = time.perf_counter(); ev2.step_enqueued(); t.append(time.perf_counter() - t0)
    print("plain step p50 %.1f us" % (np.median(t) * 1e6), flush=True)
dist.destroy_process_group()
