"""world_size-2 `gloo` test of the multi-GPU path (candidate sharding + survivor all-gather + agent sharding).

There is no GPU in the build container, so the per-rank evaluation is stood in for by the CPU oracle (test code
may use the oracle; the product's ShardedEvaluator only sees an object with the engine's methods).  What is
under test is the distributed logic: contiguous shards, global indices, the (cost, index) tie-break across
ranks, identical winners on every rank, agent round-robin.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Duck-typed stand-in for FrenetEngine backed by the oracle (tests only)."""

    def __init__(self):
        self.last = None

    def set_stream(self, _):
        pass

    def plan_step(self, inp):
        from oracle import oracle
        out = oracle.plan_step(inp, want_planes=False)
        self.last = [(inp, out)]
        return dict(out["result"])

    def plan_batch(self, inps):
        from oracle import oracle
        self.last = [(i, oracle.plan_step(i, want_planes=False)) for i in inps]
        return [dict(o["result"]) for _, o in self.last]

    def topk(self, k):
        cost = np.full((len(self.last), k), np.inf)
        idx = np.full((len(self.last), k), -1, np.int64)
        for a, (inp, out) in enumerate(self.last):
            ok = out["selectable"] & ~out["collision"]
            ids = np.nonzero(ok)[0]
            order = ids[np.lexsort((ids, out["cost"][ids]))][:k]
            cost[a, :len(order)] = out["cost"][order]
            idx[a, :len(order)] = order + inp.shard_begin
        return cost, idx


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = dict(ref_kind="arc", v0=10.0, grid=(5, 9, 11), n_obstacles=6)
        ev = ShardedEvaluator(OracleEngine(), k=8)
        inp = synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw)
        res = ev.plan_step(inp)
        # agent sharding: 5 agents over 2 ranks
        agents = [synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a,
                                        grid=(3, 5, 7), n_obstacles=a % 3, seed=a) for a in range(5)]
        ares = ev.plan_agents(agents)
        q.put((rank, res["global_best_index"], res["global_best_cost"], list(res["survivors"][:8]), inp.shard,
               [(r["best_index"], r["best_cost"]) for r in ares]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_candidate_and_agent_sharding_world2():
    from frenetix_motion_planner_amd import synthetic
    from oracle import oracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    # single-process truth
    kw = dict(ref_kind="arc", v0=10.0, grid=(5, 9, 11), n_obstacles=6)
    full = oracle.plan_step(synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, **kw), want_planes=False)
    ok = full["selectable"] & ~full["collision"]
    ids = np.nonzero(ok)[0]
    order = ids[np.lexsort((ids, full["cost"][ids]))]
    for rank, bi, bc, surv, shard, ares in got:
        assert bi == full["result"]["best_index"] and bc == full["result"]["best_cost"]
        assert surv[:4] == list(order[:4])        # merged survivors are the global cost order
    assert got[0][4] == (0, 270) and got[1][4] == (270, 270)  # 5 x 9 x 12 = 540 candidates, contiguous shards
    assert got[0][5] == got[1][5]                 # every rank knows every agent's winner
    for a, (bi, bc) in enumerate(got[0][5]):
        inp = synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a, grid=(3, 5, 7),
                                    n_obstacles=a % 3, seed=a)
        ref = oracle.plan_step(inp, want_planes=False)["result"]
        assert bi == ref["best_index"] and bc == ref["best_cost"]


# ---------------------------------------------------------------------------------------------------------
# Same sharded path with the REAL engine: two ranks share cuda:0 (RCCL refuses two ranks on one device, so the
# exchange runs over gloo on host tensors -- the ShardedEvaluator's non-NCCL branch); GPU box only.
# ---------------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = dict(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=8)
        with FrenetEngine(max_candidates=8192, max_agents=4, device=0) as eng:
            ev = ShardedEvaluator(eng, k=8)
            inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
            res = ev.plan_step(inp)
            agents = [synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a, grid=(3, 5, 7),
                                            n_obstacles=a % 3, seed=a) for a in range(5)]
            ares = ev.plan_agents(agents)
            q.put((rank, res["global_best_index"], res["global_best_cost"], list(res["survivors"][:8]), inp.shard,
                   [(r["best_index"], r["best_cost"]) for r in ares]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_sharded_path_with_real_engine_world2():
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    kw = dict(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=8)
    inp = synthetic.make_inputs(hull_builder=build_obstacle_hulls, **kw)
    with FrenetEngine(max_candidates=8192, max_agents=4, device=0) as eng:
        full = eng.plan_step(inp)
        tc, ti = eng.topk(8)
        singles = [eng.plan_step(synthetic.make_inputs(hull_builder=build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a,
                                                       grid=(3, 5, 7), n_obstacles=a % 3, seed=a)) for a in range(5)]
    for rank, bi, bc, surv, shard, ares in got:
        assert bi == full["best_index"] and bc == full["best_cost"]
        assert surv[:4] == list(ti[0][:4])   # merged survivors == single-GPU top-k
        for a, (abi, abc) in enumerate(ares):
            assert abi == singles[a]["best_index"] and abc == singles[a]["best_cost"]
    C = inp.n_candidates_global
    assert got[0][4] == (0, (C + 1) // 2) and got[1][4] == ((C + 1) // 2, C // 2)


@pytest.mark.gpu
def test_rccl_exchange_path_single_rank():
    """The device-side exchange (top-k written straight into torch tensors on torch's stream + RCCL all-gather)
    with a one-rank nccl group: everything bench.py --gpus N does per step except having N > 1 peers."""
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=8, hull_builder=build_obstacle_hulls)
        with FrenetEngine(max_candidates=8192, device=0) as eng:
            ref = eng.plan_step(inp)
            tc, ti = eng.topk(8)
            ev = ShardedEvaluator(eng, k=8, force_exchange=True)
            assert ev.on_device
            res = ev.plan_step(inp)
            assert res["global_best_index"] == ref["best_index"] and res["global_best_cost"] == ref["best_cost"]
            assert list(res["survivors"]) == [int(x) for x in ti[0] if x >= 0]
            eng.upload(inp)
            for _ in range(3):
                res2 = ev.step_enqueued()
            assert res2["global_best_index"] == ref["best_index"]
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_exchange_k1_uses_selection_result():
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=8, hull_builder=build_obstacle_hulls)
        with FrenetEngine(max_candidates=8192, device=0) as eng:
            ref = eng.plan_step(inp)
            ev = ShardedEvaluator(eng, k=1, force_exchange=True)
            res = ev.plan_step(inp)
            assert res["global_best_index"] == ref["best_index"] and res["global_best_cost"] == ref["best_cost"]
            assert list(res["survivors"]) == [ref["best_index"]]
            eng.set_winner_buffer(0)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_agent_sharded_topk_gather_single_rank():
    """BASELINE config 5's exchange: per-agent top-k written into one torch buffer, ONE all-gather, published to the
    host -- with a one-rank nccl group, compared with the engine's own top-k read-back."""
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        agents = synthetic.stress_agents(5, grid=(7, 9, 9), n_obstacles=6, hull_builder=build_obstacle_hulls)
        with FrenetEngine(max_candidates=sum(a.n_candidates for a in agents) + 5 * 64, max_steps=50, max_agents=5, device=0) as eng:
            ref = eng.plan_batch(agents)
            tc, ti = eng.topk(32)
            ev = ShardedEvaluator(eng, k=32, force_exchange=True)
            assert ev.on_device
            ev.setup_agents(5)
            eng.upload(agents)
            for _ in range(2):
                res, (sc, si) = ev.step_agents_enqueued()
            assert sc.shape == (1, 5, 32) and si.shape == (1, 5, 32)
            assert np.array_equal(si[0], ti) and np.array_equal(sc[0][ti >= 0], tc[ti >= 0])
            for a in range(5):
                assert res[a]["best_index"] == ref[a]["best_index"] == (ti[a, 0] if ti[a, 0] >= 0 else -1)
    finally:
        dist.destroy_process_group()
