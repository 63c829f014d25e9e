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


def _single_rank_nccl_group():
    """a one-rank nccl (= RCCL) group on cuda:0; the port is probed first and may be taken by another test process before the store
    binds it (pytest -n 2): try a few"""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    torch.cuda.set_device(0)
    last = None
    for _ in range(5):
        os.environ["MASTER_PORT"] = str(_free_port())
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            return dist
        except Exception as e:   # (DistNetworkError: EADDRINUSE)
            last = e
    raise last


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
            hyb = ev.plan_agents(agents[3:4])   # one agent on two ranks: its candidates are split (hybrid_assignment)
            q.put((rank, res["global_best_index"], res["global_best_cost"], list(res["survivors"][:8]), inp.shard,
                   [(r["best_index"], r["best_cost"]) for r in ares], (hyb[0]["best_index"], hyb[0]["best_cost"], hyb[0]["part"])))
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
    for rank, bi, bc, surv, shard, ares, hyb in got:
        assert bi == full["best_index"] and bc == full["best_cost"]
        assert hyb == (singles[3]["best_index"], singles[3]["best_cost"], (rank, 2))
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
    _single_rank_nccl_group()
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
    _single_rank_nccl_group()
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
def test_library_side_exchange_single_rank():
    """fx_comm_init / fx_step_exchange: the all-gather of the winners issued by the library on a communicator of its own (one
    rank here): same winner as the plain step and as the torch.distributed exchange, collisions counted, repeated steps"""
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    _single_rank_nccl_group()
    try:
        for kw in (dict(n_obstacles=8), dict(n_obstacles=0), dict(n_obstacles=6, lead_gap=15.0)):
            inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(9, 21, 21), hull_builder=build_obstacle_hulls, **kw)
            with FrenetEngine(max_candidates=8192, device=0) as eng:
                ref = eng.plan_step(inp)
                ev = ShardedEvaluator(eng, k=1, force_exchange=True)
                assert ev.lib_exchange
                res = ev.plan_step(inp)
                assert res["global_best_index"] == ref["best_index"] and res["global_best_cost"] == ref["best_cost"]
                assert res["n_collisions"] == ref["n_collisions"] and res["n_feasible"] == ref["n_feasible"]
                assert list(res["survivors"]) == ([ref["best_index"]] if ref["best_index"] >= 0 else [])
                for _ in range(3):
                    r2 = ev.step_enqueued()
                assert r2["global_best_index"] == ref["best_index"] and r2["best_index"] == ref["best_index"]
                # what bench.py does before it times anything: library exchange against the torch.distributed exchange
                assert ev.crosscheck_exchange() == 1 and ev.lib_exchange
                # ... which tries the direct mode first (all-gather received in the pinned block, stream-ordered sequence write)
                assert ev.exchange_mode == 1
                assert ev.step_enqueued()["global_best_index"] == ref["best_index"]
                eng.set_exchange_mode(0)
                assert ev.step_enqueued()["global_best_index"] == ref["best_index"]
                eng.set_winner_buffer(0)
        os.environ["FX_EXCHANGE"] = "torch"
        try:
            with FrenetEngine(max_candidates=8192, device=0) as eng:
                ev = ShardedEvaluator(eng, k=1, force_exchange=True)
                assert not ev.lib_exchange
                assert ev.plan_step(inp)["global_best_index"] == ref["best_index"]
                eng.set_winner_buffer(0)
        finally:
            del os.environ["FX_EXCHANGE"]
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
    _single_rank_nccl_group()
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


@pytest.mark.gpu
def test_library_side_topk_exchange_equals_the_torch_path():
    """fx_step_exchange_topk (evaluation, selection, top-k, all-gather, publication enqueued by the library) against the
    torch.distributed path (FX_EXCHANGE=torch) on a one-rank nccl group: same survivors, same results."""
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    _single_rank_nccl_group()
    try:
        agents = synthetic.stress_agents(4, grid=(7, 9, 9), n_obstacles=6, hull_builder=build_obstacle_hulls)
        got = {}
        for mode in ("lib", "torch"):
            os.environ["FX_EXCHANGE"] = mode
            with FrenetEngine(max_candidates=sum(a.n_candidates for a in agents) + 4 * 64, max_steps=50, max_agents=4, device=0) as eng:
                ev = ShardedEvaluator(eng, k=16, force_exchange=True)
                ev.setup_agents(4)
                assert ev.lib_exchange_agents == (mode == "lib")
                eng.upload(agents)
                for _ in range(3):
                    res, (sc, si) = ev.step_agents_enqueued()
                got[mode] = ([dict(r) if isinstance(r, dict) else r.as_dict() for r in res], sc.copy(), si.copy())
        a, b = got["lib"], got["torch"]
        assert a[1].shape == (1, 4, 16) and np.array_equal(a[2], b[2]) and np.array_equal(a[1][a[2] >= 0], b[1][b[2] >= 0])
        for ra, rb in zip(a[0], b[0]):
            for k in ("best_index", "best_cost", "n_feasible", "n_collisions", "n_returned"):
                assert ra[k] == rb[k], k
    finally:
        os.environ.pop("FX_EXCHANGE", None)
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------
# Watchdog: every wait on a hand-off is bounded in time (the reference: TIMEOUT = 20 s, simulation.py:637,655)
# ---------------------------------------------------------------------------------------------------------
def test_wait_word_times_out_and_returns_when_the_word_arrives():
    """fx_wait_word is the wait behind fx_finish / the exchanges (pure host code): FX_ERR_TIMEOUT after the bound when nothing
    arrives, FX_OK as soon as the word does."""
    import ctypes as C
    import threading
    import time
    from frenetix_motion_planner_amd import _abi, _lib
    L = _lib.lib()
    word = C.c_uint64(0)
    t0 = time.perf_counter()
    rc = L.fx_wait_word(C.addressof(word), 7, 150)
    dt = time.perf_counter() - t0
    assert rc == _abi.FX_ERR_TIMEOUT and 0.14 < dt < 1.5
    assert b"no answer from the device within 150 ms" in L.fx_last_error()
    with pytest.raises(_lib.FxTimeoutError):
        _lib.check(rc)
    threading.Timer(0.05, lambda: setattr(word, "value", 7)).start()
    t0 = time.perf_counter()
    assert L.fx_wait_word(C.addressof(word), 7, 5000) == 0 and time.perf_counter() - t0 < 2.0


def _absent_peer_worker(rank, port, q):
    """rank 0 runs a sharded plan step; rank 1 joins the group and then never enters the exchange"""
    sys.path.insert(0, ROOT)
    import datetime
    import time
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=2, timeout=datetime.timedelta(seconds=3))
    if rank == 1:
        time.sleep(12)      # alive, but not in the collective
        os._exit(0)
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator, exit_on_timeout
    from tests.oracle_engine import OracleEngine
    from oracle import oracle
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, level=1, hull_builder=oracle.build_obstacle_hulls)
    ev = ShardedEvaluator(OracleEngine(), k=4)
    q.put("started")
    exit_on_timeout(ev.plan_step, inp, code=7)
    q.put("returned")      # not reached: the exchange times out and the process ends with code 7


@pytest.mark.timeout(120)
def test_a_rank_that_never_joins_ends_its_peer_with_an_error_not_a_hang():
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_absent_peer_worker, args=(r, port, q)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    assert q.get(timeout=60) == "started"
    procs[0].join(timeout=60)
    assert procs[0].exitcode == 7, procs[0].exitcode        # gave up after the group's 3 s bound, non-zero exit
    assert time.time() - t0 < 60 and q.empty()
    procs[1].join(timeout=30)


# ---------------------------------------------------------------------------------------------------------
# Hybrid agent x candidate sharding (fewer agents than ranks: BASELINE config 4 has 5 agents on 8 GPUs)
# ---------------------------------------------------------------------------------------------------------
def test_hybrid_assignment_gives_every_rank_work():
    from frenetix_motion_planner_amd.distributed import hybrid_assignment, shard_range
    items = hybrid_assignment(5, 8)                       # config 4
    assert len(items) == 8 and all(len(it) == 1 for it in items)
    parts = {}
    for (a, p, n), in items:
        parts.setdefault(a, []).append((p, n))
    assert sorted(parts) == [0, 1, 2, 3, 4]
    assert [len(parts[a]) for a in range(5)] == [2, 2, 2, 1, 1]   # 8 = 3 x 2 + 2 x 1
    for a, pl in parts.items():
        assert sorted(p for p, _ in pl) == list(range(len(pl))) and all(n == len(pl) for _, n in pl)
    # the parts of an agent tile its candidates
    assert [shard_range(10051, p, 2) for p in range(2)] == [(0, 5026), (5026, 5025)]
    # at least as many agents as ranks: plain round-robin
    assert hybrid_assignment(5, 2) == [[(0, 0, 1), (2, 0, 1), (4, 0, 1)], [(1, 0, 1), (3, 0, 1)]]
    assert hybrid_assignment(1, 1) == [[(0, 0, 1)]]
    with pytest.raises(ValueError):
        hybrid_assignment(0, 4)


def _agents(n):
    from frenetix_motion_planner_amd import synthetic
    from oracle import oracle
    return [synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a, grid=(4, 7, 9),
                                  n_obstacles=1 + a % 3, seed=a, lead_gap=18.0 if a % 2 else 0.0) for a in range(n)]


def test_five_agents_on_eight_emulated_ranks_match_single_rank():
    """config 4's shape without eight processes: every 'rank' evaluates its item with the oracle stand-in, the gathered
    rows are merged the way every rank merges them -- the winners equal the single-rank winners of the whole agents."""
    import copy
    from frenetix_motion_planner_amd.distributed import hybrid_assignment, merge_agent_parts, shard_range
    from oracle import oracle
    agents = _agents(5)
    rows = []
    for rank_items in hybrid_assignment(5, 8):
        for a, part, n_parts in rank_items:
            inp = copy.copy(agents[a])
            if n_parts > 1:
                inp.shard = shard_range(agents[a].n_candidates_global, part, n_parts)
            res = oracle.plan_step(inp, want_planes=False)["result"]
            rows.append((a, res["best_cost"], res["best_index"]))
    winners = merge_agent_parts(5, rows)
    for a in range(5):
        ref = oracle.plan_step(agents[a], want_planes=False)["result"]
        assert winners[a] == (ref["best_cost"], ref["best_index"]) or (ref["best_index"] < 0 and winners[a][1] < 0)
    assert any(w[1] >= 0 for w in winners)


def _hybrid_worker(rank, world, port, q, n_agents):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ev = ShardedEvaluator(OracleEngine(), k=1)
        ares = ev.plan_agents(_agents(n_agents))
        q.put((rank, [(r["best_index"], r["best_cost"]) for r in ares], [r.get("part") for r in ares]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,n_agents", [(2, 1), (3, 2)])
def test_hybrid_sharding_gloo(world, n_agents):
    """1 agent on 2 ranks, 2 agents on 3 ranks: the candidates of an agent are split over its ranks, one all-gather, the
    same winners on every rank, equal to the single-rank winners"""
    from oracle import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hybrid_worker, args=(r, world, port, q, n_agents)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    agents = _agents(n_agents)
    for rank, winners, parts in got:
        assert winners == got[0][1]
        for a, (bi, bc) in enumerate(winners):
            ref = oracle.plan_step(agents[a], want_planes=False)["result"]
            assert (bi, bc) == (ref["best_index"], ref["best_cost"])
    # every rank evaluated a part: with fewer agents than ranks nobody idles
    assert all(any(p is not None for p in parts) for _, _, parts in got)


class _CommStubEngine:
    """Only what ShardedEvaluator._init_library_exchange touches; `fail` = which call raises on this rank."""

    def __init__(self, fail=None):
        self.fail, self.inits, self.destroys, self.uid_seen = fail, 0, 0, None

    def comm_check(self, world):
        return self.fail != "check"     # local preconditions (RCCL present, capacity): nothing collective

    def comm_unique_id(self):
        if self.fail == "uid":
            raise RuntimeError("no communicator library")
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        if self.fail == "init":
            raise RuntimeError("communicator initialisation failed")
        self.inits += 1
        self.uid_seen = bytes(uid)

    def comm_set_agents(self, n):
        if self.fail == "rows":
            raise RuntimeError("agent rows refused")
        self.rows = n

    def comm_destroy(self):
        self.destroys += 1


def _exchange_agreement_worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fail = {"ok": None, "uid_fails_on_rank0": "uid" if rank == 0 else None,
                "init_fails_on_rank1": "init" if rank == 1 else None,
                "precondition_fails_on_rank1": "check" if rank == 1 else None,
                "rows_refused_on_rank1": "rows" if rank == 1 else None, "agent_rows_differ": None}[scenario]
        ev = ShardedEvaluator.__new__(ShardedEvaluator)   # the agreement logic alone (the constructor needs a GPU for this path)
        ev.torch, ev.dist, ev.group, ev.rank, ev.world, ev.on_device = torch, dist, None, rank, world, False
        ev.engine = _CommStubEngine(fail)
        ok = ev._init_library_exchange(torch.device("cpu"))
        if scenario == "agent_rows_differ":   # setup_agents with different agents per rank: nobody may use the library exchange
            assert ok and ev.engine.rows == 1
            ok = ev._agree_agent_rows(2 + rank)
        q.put((rank, ok, ev.engine.inits, ev.engine.destroys, ev.engine.uid_seen == bytes(range(128))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario", ["ok", "uid_fails_on_rank0", "init_fails_on_rank1", "precondition_fails_on_rank1",
                                      "rows_refused_on_rank1", "agent_rows_differ"])
def test_library_exchange_setup_is_agreed_by_all_ranks(scenario):
    """The in-library exchange is used only if EVERY rank could set it up: rank 0 draws the id (or says it could not), the id
    travels over the torch group, and one failing rank sends all of them to the torch.distributed path (communicators that
    were created are destroyed) -- otherwise half the ranks would wait in an all-gather the others never enter."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_agreement_worker, args=(r, world, port, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    oks = [g[1] for g in got]
    assert oks == ([True, True] if scenario == "ok" else [False, False])
    if scenario == "ok":
        assert all(g[2] == 1 and g[3] == 0 and g[4] for g in got)
    elif scenario in ("uid_fails_on_rank0", "precondition_fails_on_rank1"):
        # nobody initialises without an id -- and nobody enters the (blocking, collective) initialisation when one rank's
        # local preconditions fail: its peers would wait inside ncclCommInitRank for a rank that never comes
        assert all(g[2] == 0 and g[3] == 0 for g in got)
    elif scenario in ("rows_refused_on_rank1", "agent_rows_differ"):
        # the element count of the library's all-gathers must be the same everywhere: one rank that cannot take the agreed agent
        # rows -- or ranks that ask for different numbers -- and every communicator is destroyed again
        assert all(g[2] == 1 and g[3] == 1 for g in got)
    else:
        assert got[0][2] == 1 and got[0][3] == 1                    # rank 0 had a communicator: destroyed again
        assert got[1][2] == 0 and got[1][3] == 0


def _crosscheck_worker(rank, world, port, q, scenario):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd._lib import FxError
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        calls = []

        class Ev(ShardedEvaluator):
            def step_enqueued(self):   # what the two exchanges report as the global winner
                if self.lib_exchange:
                    # the library's own all-gather is not the torch group's: entered by every rank, local errors reported after it
                    calls.append("lib")
                    if scenario == "library_raises_on_rank1" and self.rank == 1:
                        raise FxError("FX_ERR_CAPACITY (reported behind the all-gather)")
                    wrong = scenario == "library_wrong_on_rank1" and self.rank == 1
                    return {"global_best_index": 41 if wrong else 42, "global_best_cost": 1.5}
                # the torch.distributed exchange IS a collective on the group the agreement rounds use: a rank that skipped it
                # while its peer entered it would pair this all_gather with the peer's all_reduce (a hang or garbage)
                calls.append("torch")
                got = [None] * self.world
                self.dist.all_gather_object(got, ("torch-half", self.rank))
                assert got == [("torch-half", r) for r in range(self.world)], got
                return {"global_best_index": 42, "global_best_cost": 1.5}

        class Eng(_CommStubEngine):
            def set_exchange_mode(self, mode):
                if scenario == "direct_mode_refused_on_rank1" and rank == 1 and mode == 1:
                    raise ValueError("stream-ordered write refused")
                self.mode = mode

        ev = Ev.__new__(Ev)
        ev.torch, ev.dist, ev.group, ev.rank, ev.world, ev.on_device = torch, dist, None, rank, world, False
        ev.engine = Eng()
        ev.lib_exchange, ev.lib_exchange_agents = True, False
        state = ev.crosscheck_exchange()
        q.put((rank, state, ev.lib_exchange, ev.engine.destroys, getattr(ev, "exchange_mode", None), calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scenario", ["agree", "library_wrong_on_rank1", "library_raises_on_rank1", "direct_mode_refused_on_rank1"])
def test_library_exchange_is_cross_checked_against_the_torch_exchange(scenario):
    """Before anything is timed the library-side exchange has to report the winner the torch.distributed exchange reports; one
    rank that sees a difference switches it off on every rank (bench.py: `exchange` in the line says which one ran).  Whatever
    fails on ONE rank, every rank issues the same collectives in the same order: a rank whose library half raises keeps its
    peers out of the torch half (mismatched collectives on one group), a rank that cannot set the direct mode keeps
    them out of the library's all-gather."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_crosscheck_worker, args=(r, world, port, q, scenario)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if scenario == "agree":
        assert [g[1:] for g in got] == [(1, True, 0, 1, ["lib", "torch"])] * 2
    elif scenario == "library_wrong_on_rank1":   # both modes are tried, both disagree on rank 1
        assert [g[1:] for g in got] == [(0, False, 1, None, ["lib", "torch", "lib", "torch"])] * 2
    elif scenario == "library_raises_on_rank1":   # nobody enters the torch half in either mode
        assert [g[1:] for g in got] == [(0, False, 1, None, ["lib", "lib"])] * 2
    else:   # mode 1 is skipped by BOTH ranks before anything collective runs in it; mode 0 agrees
        assert [g[1:] for g in got] == [(1, True, 0, 0, ["lib", "torch"])] * 2


# ---------------------------------------------------------------------------------------------------------
# The RCCL path with TWO ranks on two devices: the in-library exchanges (fx_step_exchange, fx_step_exchange_topk) and the
# torch.distributed exchange over nccl.  Needs >= 2 visible GPUs; skipped on the one-GPU boxes of the pool, there for the day
# the suite runs on a node.
# ---------------------------------------------------------------------------------------------------------
def _nccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator, exit_on_timeout
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        out = {}
        # candidate split, k = 1: the winner exchange inside the library, cross-checked against torch.distributed
        inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=8, hull_builder=build_obstacle_hulls)
        with FrenetEngine(max_candidates=8192, device=rank) as eng:
            eng.set_timeout_ms(60000)
            ev = ShardedEvaluator(eng, k=1)
            out["lib_exchange"] = bool(ev.lib_exchange)
            out["comm"] = eng.comm_info() if ev.lib_exchange else None
            res = exit_on_timeout(ev.plan_step, inp)
            out["winner"] = (int(res["global_best_index"]), float(res["global_best_cost"]))
            out["crosscheck"] = exit_on_timeout(ev.crosscheck_exchange)
            r2 = exit_on_timeout(ev.step_enqueued)
            out["winner_again"] = (int(r2["global_best_index"]), float(r2["global_best_cost"]))
            eng.set_winner_buffer(0)
        # agent sharding with the per-agent top-k gather (config 5's exchange): library path and torch path give the same survivors
        agents = synthetic.stress_agents(3, grid=(7, 9, 9), n_obstacles=6, first_agent=3 * rank, hull_builder=build_obstacle_hulls)
        with FrenetEngine(max_candidates=sum(a.n_candidates for a in agents) + 3 * 64, max_steps=50, max_agents=3, device=rank) as eng:
            eng.set_timeout_ms(60000)
            ev = ShardedEvaluator(eng, k=8)
            ev.setup_agents(3)
            eng.upload(agents)
            out["lib_exchange_agents"] = bool(ev.lib_exchange_agents)
            _, (sc, si) = exit_on_timeout(ev.step_agents_enqueued)
            out["survivors"] = (sc.tolist(), si.tolist())
            out["crosscheck_agents"] = exit_on_timeout(ev.crosscheck_exchange)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_world2_library_and_torch_exchange():
    """Two ranks on two GPUs over RCCL: candidate split with the in-library winner exchange (every rank reports the single-GPU
    winner, ncclCommCount = 2, library exchange == torch.distributed exchange) and agent sharding with the in-library per-agent
    top-k gather (every rank holds every rank's survivors).  Skipped where fewer than two devices are visible."""
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls, device_count
    if device_count() < 2:
        pytest.skip("needs two GPUs (the pool's boxes have one): runs on a multi-GPU node")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=800) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(9, 21, 21), n_obstacles=8, hull_builder=build_obstacle_hulls)
    with FrenetEngine(max_candidates=8192, device=0) as eng:
        full = eng.plan_step(inp)
    for rank in range(world):
        o = got[rank]
        assert o["winner"] == (full["best_index"], full["best_cost"]) == o["winner_again"]
        assert o["crosscheck"] == 1 and o["crosscheck_agents"] == 1
        if o["lib_exchange"]:
            assert o["comm"]["world"] == 2 and o["comm"]["rank"] == rank and o["comm"]["rccl_ranks"] in (2, -1) and o["comm"]["agent_rows"] == 1
    assert got[0]["survivors"] == got[1]["survivors"]   # [world][agents][k] on every rank
    sc, si = (np.asarray(x) for x in got[0]["survivors"])
    assert sc.shape == (2, 3, 8)
    for r in range(world):   # rank r's rows are its own agents' top-k
        agents = synthetic.stress_agents(3, grid=(7, 9, 9), n_obstacles=6, first_agent=3 * r, hull_builder=build_obstacle_hulls)
        with FrenetEngine(max_candidates=sum(a.n_candidates for a in agents) + 3 * 64, max_steps=50, max_agents=3, device=0) as eng:
            eng.plan_batch(agents)
            tc, ti = eng.topk(8)
        assert np.array_equal(si[r], ti) and np.array_equal(sc[r], tc)


# ---------------------------------------------------------------------------------------------------------
# Eight real processes over gloo (the node's world size): BASELINE config 5's agent round-robin with the per-agent top-k gather,
# and BASELINE config 4's five agents on eight ranks (hybrid agent x candidate split).  Reduced grids, the oracle-backed engine.
# ---------------------------------------------------------------------------------------------------------
class _BatchOracleEngine(OracleEngine):
    """the stand-in with the resident-batch surface ShardedEvaluator.step_agents_enqueued drives (upload / evaluate / finish)"""

    def upload(self, inputs):
        self._resident = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]

    def evaluate(self):
        self._res = self.plan_batch(self._resident)

    def finish(self):
        return self._res


def _world8_worker(rank, world, port, q, n_agents_total, k):
    sys.path.insert(0, ROOT)
    import datetime
    import torch.distributed as dist
    from frenetix_motion_planner_amd import synthetic
    from frenetix_motion_planner_amd.distributed import ShardedEvaluator, hybrid_assignment
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        # config 5: agent a of the global list lives on rank a // n_local (bench.py: first_agent = rank * n_local)
        n_local = n_agents_total // world
        agents = synthetic.stress_agents(n_local, grid=(3, 5, 5), n_obstacles=3, first_agent=rank * n_local,
                                         hull_builder=oracle.build_obstacle_hulls)
        eng = _BatchOracleEngine()
        ev = ShardedEvaluator(eng, k=k)
        ev.setup_agents(n_local)
        eng.upload(agents)
        res, (sc, si) = ev.step_agents_enqueued()
        # config 4: five agents on eight ranks, every rank one item, split agents merged by the (cost, index) minimum
        five = [synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a, grid=(4, 7, 9),
                                      n_obstacles=a % 3, seed=a) for a in range(5)]
        ev2 = ShardedEvaluator(OracleEngine(), k=1)
        hyb = ev2.plan_agents(five)
        q.put((rank, sc.shape, sc.tolist(), si.tolist(), [r["best_index"] for r in res],
               [(h["best_index"], h["best_cost"]) for h in hyb], hybrid_assignment(5, world)[rank]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_world8_agent_round_robin_and_hybrid_split():
    """WORLD_SIZE = 8 with real processes (gloo): (a) BASELINE config 5's exchange -- 64 agents (8 per rank here; 32 per rank in the
    bench), one per-agent top-k all-gather -- leaves EVERY rank with every agent's survivors, equal to a single process planning
    each agent on its own; (b) BASELINE config 4's 5 agents on 8 ranks: three agents split over two ranks each, two whole, nobody
    idles, every rank ends with every agent's global winner == the unsplit agent's."""
    from frenetix_motion_planner_amd import synthetic
    from oracle import oracle
    world, n_total, k = 8, 64, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q, n_total, k)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # (a) every rank holds the same [world, n_local, k] survivor block, and it is what one process finds agent by agent
    n_local = n_total // world
    assert all(g[1] == (world, n_local, k) for g in got)
    assert all(g[2] == got[0][2] and g[3] == got[0][3] for g in got)
    cost, idx = np.array(got[0][2]), np.array(got[0][3])
    single = _BatchOracleEngine()
    agents = synthetic.stress_agents(n_total, grid=(3, 5, 5), n_obstacles=3, hull_builder=oracle.build_obstacle_hulls)
    want = single.plan_batch(agents)
    tc, ti = single.topk(k)
    assert np.array_equal(idx.reshape(n_total, k), ti) and np.array_equal(cost.reshape(n_total, k), tc)
    for r, g in enumerate(got):   # each rank's local winners are its slice of the global list
        assert g[4] == [w["best_index"] for w in want[r * n_local:(r + 1) * n_local]]
    assert sum(w["best_index"] >= 0 for w in want) >= n_total // 2
    # (b) the hybrid split: 8 = 3 x 2 + 2 x 1, one item per rank, the same winners everywhere
    items = [g[6] for g in got]
    assert all(len(it) == 1 for it in items)
    assert sorted((a, p, n) for ((a, p, n),) in items) == [(0, 0, 2), (0, 1, 2), (1, 0, 2), (1, 1, 2), (2, 0, 2), (2, 1, 2), (3, 0, 1), (4, 0, 1)]
    assert all(g[5] == got[0][5] for g in got)
    for a, (bi, bc) in enumerate(got[0][5]):
        inp = synthetic.make_inputs(hull_builder=oracle.build_obstacle_hulls, ref_kind="arc", v0=6.0 + 2 * a, grid=(4, 7, 9),
                                    n_obstacles=a % 3, seed=a)
        ref = oracle.plan_step(inp, want_planes=False)["result"]
        assert bi == ref["best_index"] and bc == ref["best_cost"]
