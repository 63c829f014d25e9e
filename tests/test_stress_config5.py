"""BASELINE config 5 ("synthetic stress": many agents x 100 k candidates x 51 samples x 20 obstacles) on the GPU:
a reduced batch against the oracle, and the full per-agent size through size-independent properties plus an oracle
pass over every candidate of two agents."""
import numpy as np
import pytest

from frenetix_motion_planner_amd import _abi, synthetic

FRAGILE = 1e-9
COST_RTOL = 1e-9


def test_stress_agents_are_rank_independent():
    """Agent a's inputs depend on (seed, a) only: any rank can build any slice of the global agent list."""
    a = synthetic.stress_agents(4, grid=(3, 5, 5), n_obstacles=3)
    b = synthetic.stress_agents(2, grid=(3, 5, 5), n_obstacles=3, first_agent=2)
    for x, y in zip(a[2:], b):
        assert np.array_equal(x.x0_lon, y.x0_lon) and np.array_equal(x.x0_lat, y.x0_lat) and x.v_des == y.v_des
        assert np.array_equal(x.obstacles["pos"], y.obstacles["pos"]) and np.array_equal(x.v_samp, y.v_samp)
    assert len({round(float(x.x0_lon[1]), 6) for x in a}) == 4 and a[0].N == 50 and a[0].n_samples == 51
    full = synthetic.stress_agents(1, n_obstacles=0)[0]
    assert full.n_candidates == 39 * 51 * 52


@pytest.mark.gpu
def test_reduced_stress_batch_vs_oracle():
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    from oracle import oracle
    from tests.test_hip_parity import compare
    hip = synthetic.stress_agents(6, grid=(7, 9, 9), n_obstacles=20, hull_builder=build_obstacle_hulls, write_bundle=True)
    ora = synthetic.stress_agents(6, grid=(7, 9, 9), n_obstacles=20, hull_builder=oracle.build_obstacle_hulls, write_bundle=True)
    with FrenetEngine(max_candidates=sum(a.n_candidates for a in hip) + 6 * 64, max_steps=50, max_agents=6) as eng:
        res = eng.plan_batch(hip)
        winners = 0
        for a in range(6):
            out = oracle.plan_step(ora[a])
            compare(eng, hip[a], out, res[a], agent=a)
            if np.all(out["margin"] >= FRAGILE):
                assert res[a]["best_index"] == out["result"]["best_index"]
                assert res[a]["n_collisions"] == out["result"]["n_collisions"]
            winners += res[a]["best_index"] >= 0
        assert winners >= 3


@pytest.mark.gpu
def test_full_size_stress_properties():
    """Four agents of the full per-agent size (103 428 candidates x 51 samples, 20 obstacles) in one launch, select-only."""
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    from oracle import oracle
    n = 4
    agents = synthetic.stress_agents(n, hull_builder=build_obstacle_hulls)
    assert all(a.n_candidates == 103428 and a.n_samples == 51 and a.obstacles["K"] == 20 for a in agents)
    with FrenetEngine(max_candidates=n * 103428 + n * 64, max_steps=50, max_agents=n) as eng:
        res = eng.plan_batch(agents)
        per = [eng.costs(a) for a in range(n)]
        tc, ti = eng.topk(32)
        for a in range(n):
            cost, flags = per[a]
            ok = ((flags & _abi.FX_FLAG_SELECTABLE) != 0) & ((flags & _abi.FX_FLAG_COLLISION) == 0)
            ids = np.nonzero(ok)[0]
            order = ids[np.lexsort((ids, cost[ids]))]
            # winner = first collision-free entry of the stable cost order; top-32 = its sorted prefix
            assert res[a]["best_index"] == (order[0] if len(order) else -1)
            k = min(32, len(order))
            assert np.array_equal(ti[a][:k], order[:k]) and np.array_equal(tc[a][:k], cost[order[:k]])
            assert np.all(ti[a][k:] == -1)
            # counters agree with the flag words
            ret = (flags & _abi.FX_FLAG_RETURNED) != 0
            assert res[a]["n_returned"] == int(ret.sum())
            assert res[a]["n_feasible"] == int((((flags & 3) == 3) & ret).sum())
            if len(order):   # collisions counted = colliding candidates ordered before the winner (planner.py:336-357)
                sel = np.nonzero((flags & _abi.FX_FLAG_SELECTABLE) != 0)[0]
                so = sel[np.lexsort((sel, cost[sel]))]
                pos = int(np.nonzero(so == order[0])[0][0])
                assert res[a]["n_collisions"] == pos
        # idempotence and batch == single
        res2 = eng.plan_batch(agents)
        for a in range(n):
            c2, f2 = eng.costs(a)
            assert np.array_equal(c2, per[a][0]) and np.array_equal(f2, per[a][1]) and res2[a]["best_index"] == res[a]["best_index"]
        one = eng.plan_step(agents[2])
        c1, f1 = eng.costs(0)
        # a single-agent launch may split the horizon differently (auto-tuning by wave count): sums agree to rounding
        assert np.allclose(c1, per[2][0], rtol=1e-12, atol=0) and np.array_equal(f1, per[2][1])
        assert one["best_index"] == res[2]["best_index"]
    # the oracle over every candidate of two agents
    ora = synthetic.stress_agents(2, hull_builder=oracle.build_obstacle_hulls)
    for a in range(2):
        out = oracle.plan_step(ora[a], want_planes=False)
        robust = out["margin"] >= FRAGILE
        cost, flags = per[a]
        assert np.array_equal(flags[robust], out["flags"][robust])
        c = out["costed"] & robust
        assert (np.abs(cost[c] - out["cost"][c]) / np.maximum(np.abs(out["cost"][c]), 1e-12)).max() < COST_RTOL
        assert res[a]["best_index"] == out["result"]["best_index"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_full_size_batch_of_32_agents():
    """The launch bench.py --workload config5 (and every rank of SCALE) times: 32 agents x 103 428 candidates x 51 samples x 20
    obstacles in ONE batched launch (grid.y = 32), select-only, with the per-agent top-32 behind it.  Size-independent properties on
    all 32 agents (winner = first collision-free entry of the stable cost order, top-k = its sorted prefix, counters = the flag
    words' counts, collisions counted in front of the winner), batch == single launch for two of them, and the oracle over EVERY
    candidate of two agents (multi-threaded range evaluation)."""
    from frenetix_motion_planner_amd.engine import FrenetEngine, build_obstacle_hulls
    from oracle import oracle
    n = 32
    agents = synthetic.stress_agents(n, hull_builder=build_obstacle_hulls)
    assert all(a.n_candidates == 103428 and a.n_samples == 51 and a.obstacles["K"] == 20 for a in agents)
    with FrenetEngine(max_candidates=n * 103428 + n * 64, max_steps=50, max_agents=n) as eng:
        res = eng.plan_batch(agents)
        info = eng.step_info()
        assert info["agents"] == 32 and info["lanes_per_candidate"] == 1
        tc, ti = eng.topk(32)
        keep = {}
        for a in range(n):
            cost, flags = eng.costs(a)
            if a in (5, 29):
                keep[a] = (cost, flags)
            sel = (flags & _abi.FX_FLAG_SELECTABLE) != 0
            ok = sel & ((flags & (_abi.FX_FLAG_COLLISION | _abi.FX_FLAG_BOUNDARY)) == 0)
            ids = np.nonzero(ok)[0]
            order = ids[np.lexsort((ids, cost[ids]))]
            assert res[a]["best_index"] == (order[0] if len(order) else -1), a
            k = min(32, len(order))
            assert np.array_equal(ti[a][:k], order[:k]) and np.array_equal(tc[a][:k], cost[order[:k]]) and np.all(ti[a][k:] == -1)
            ret = (flags & _abi.FX_FLAG_RETURNED) != 0
            assert res[a]["n_returned"] == int(ret.sum()) and res[a]["n_feasible"] == int((((flags & 3) == 3) & ret).sum())
            coll = sel & ((flags & _abi.FX_FLAG_COLLISION) != 0)
            if len(order):
                w = order[0]
                before = (cost[coll] < cost[w]) | ((cost[coll] == cost[w]) & (np.nonzero(coll)[0] < w))
                assert res[a]["n_collisions"] == int(before.sum()), a
            else:
                assert res[a]["n_collisions"] == int(coll.sum())
        assert sum(r["best_index"] >= 0 for r in res) >= 24
        for a in (5, 29):   # batch == single launch
            one = eng.plan_step(agents[a])
            c1, f1 = eng.costs(0)
            assert np.array_equal(f1, keep[a][1]) and np.allclose(c1, keep[a][0], rtol=1e-12, atol=0) and one["best_index"] == res[a]["best_index"]
    ora = synthetic.stress_agents(n, hull_builder=oracle.build_obstacle_hulls)
    for a in (5, 29):
        out = oracle.plan_step(ora[a], want_planes=False)
        robust = out["margin"] >= FRAGILE
        cost, flags = keep[a]
        assert np.array_equal(flags[robust], out["flags"][robust])
        c = out["costed"] & robust
        assert (np.abs(cost[c] - out["cost"][c]) / np.maximum(np.abs(out["cost"][c]), 1e-12)).max() < COST_RTOL
        assert res[a]["best_index"] == out["result"]["best_index"] and res[a]["n_collisions"] == out["result"]["n_collisions"]
