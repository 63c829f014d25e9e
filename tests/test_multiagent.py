"""Per-agent glue (FrenetPlannerInterfaceHip, VelocityPlanner) and batched multi-agent stepping (AgentBatchHip,
MultiAgentSimulation) -- BASELINE config 4 (multi-agent ZAM_Tjunction) on the data fixture
tests/golden/ZAM_Tjunction-1_42_T-1.scenario.json.

CPU tests drive the host logic with the oracle-backed stand-in engine (tests/oracle_engine.py); GPU tests run the same
closed loop on the HIP engine and compare it with the stand-in, and check batched == individually stepped.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "ZAM_Tjunction-1_42_T-1.scenario.json")
XML = "/root/reference/example_scenarios/ZAM_Tjunction-1_42_T-1.xml"

from frenetix_motion_planner_amd import commonroad_xml as crx  # noqa: E402
from frenetix_motion_planner_amd import multiagent  # noqa: E402
from frenetix_motion_planner_amd.frenet_interface import (FrenetPlannerInterfaceHip, VelocityPlanner,  # noqa: E402
                                                          create_from_initial_state)
from frenetix_motion_planner_amd.reactive_planner import PlannerConfig  # noqa: E402
from tests.oracle_engine import OracleEngine  # noqa: E402


@pytest.fixture(scope="module")
def scenario():
    return crx.read_scenario_json(FIXTURE)


@pytest.mark.skipif(not os.path.exists(XML), reason="reference example scenarios not present (GPU box)")
def test_fixture_is_what_the_reader_extracts_from_the_xml():
    assert crx.scenario_to_dict(crx.read_scenario(XML)) == crx.scenario_to_dict(crx.read_scenario_json(FIXTURE))


def test_agent_selection_and_goal_creation(scenario):
    ids = multiagent.select_agent_obstacles(scenario)
    assert ids == [1, 4, 5, 7]                       # obstacle 2 moves only 7.3 m (simulation.py:191-192)
    assert multiagent.select_agent_obstacles(scenario, 2) == [1, 4]
    for oid in ids:
        pp = multiagent.planning_problem_for_obstacle(scenario, oid)
        fin = scenario.obstacles[oid].state_list[-1]
        g = pp.goals[0]
        assert scenario.lanelets[g.lanelet_ids[0]].contains(fin.position)
        assert g.time_interval == (fin.time_step - 20, fin.time_step + 20)
        assert g.velocity_interval == (fin.velocity - 2, fin.velocity + 2)
        assert len(scenario.route_reference_path(pp)) >= 2


def test_initial_state_shift_and_velocity_planner(scenario):
    pp = scenario.planning_problems[60000]
    x0 = create_from_initial_state(pp.initial_state, 2.5789, 1.4227)
    o = pp.initial_state.orientation
    assert np.allclose(x0.position + 1.4227 * np.array([np.cos(o), np.sin(o)]), pp.initial_state.position)
    assert x0.steering_angle == 0.0 and x0.velocity == pp.initial_state.velocity
    itf = FrenetPlannerInterfaceHip(60000, scenario, pp, engine=OracleEngine())
    vp = itf.velocity_planner
    assert vp.used_goal_metric == "lanelets_of_goal_position" and vp.goal_s_position is not None
    s0 = itf.x_cl[0][0]
    steps = vp.calc_remaining_time_steps(0, 0.0)
    lo, hi = pp.goals[0].time_interval
    assert steps == int((lo + hi) / 2)
    want = (vp.goal_s_position - s0) / round(steps * scenario.dt, 3)
    v = vp.calculate_desired_velocity(itf.x_0, s0)
    assert v == pytest.approx(min(max(want, x0.velocity - 5), x0.velocity + 5))
    assert VelocityPlanner.clip_velocity(100.0, 10.0) == 15.0 and VelocityPlanner.clip_velocity(-3.0, 2.0) == 0


def test_replanning_counter_and_state_handover(scenario):
    """frenet_interface.py:231-287: a plan step every `replanning_frequency` steps, the steps between advance along the
    stored trajectory; x_cl is always (lon_list[k], lat_list[k]) of the stored pair."""
    pp = scenario.planning_problems[60000]
    eng = OracleEngine()
    calls = []
    plan_step = eng.plan_step
    eng.plan_step = lambda inp: (calls.append(1), plan_step(inp))[1]
    itf = FrenetPlannerInterfaceHip(60000, scenario, pp, engine=eng)
    preds = scenario.ground_truth_predictions(0, 30)
    pair = None
    for t in range(7):
        itf.update_planner(None, preds)
        planned = itf.needs_plan()
        sel, cnt = itf.step_interface(t)
        assert planned == (t % 3 == 0) and cnt == t % 3 and len(calls) == t // 3 + 1
        if planned:
            pair = itf.trajectory_pair
            assert sel is pair[0]
        k = 1 + cnt
        assert itf.x_cl == (pair[2][k], pair[3][k])
        assert itf.x_0.time_step == pair[0][k].time_step == t + 1
        assert np.array_equal(itf.x_0.position, pair[0][k].position)
    assert len(itf.record_state_list) == 8 and len(itf.record_input_list) == 8
    assert itf.record_input_list[2]["steering_angle_speed"] == pytest.approx(
        (itf.record_state_list[2].steering_angle - itf.record_state_list[1].steering_angle) / 0.1)


def _run_sim(sc, steps, **kw):
    sim = multiagent.MultiAgentSimulation(sc, **kw)
    winners = []
    for _ in range(steps):
        sim.step()
        winners.append([getattr(a.optimal_trajectory, "global_id", a.optimal_trajectory.uniqueId) if a.optimal_trajectory is not None else -1
                        for a in sim.batch.agents])
    return sim, winners


def test_batched_step_equals_individual_steps_cpu(scenario):
    """AgentBatchHip (one plan_batch per step) against each agent stepped on its own (agent_batch.py:186-189)."""
    sim, winners = _run_sim(scenario, 4, engine_factory=OracleEngine)
    assert sim.batch.launches == 2 and sim.agent_ids == [60000, 1, 4, 5, 7]
    solo = multiagent.MultiAgentSimulation(scenario, engine_factory=OracleEngine)
    for t in range(4):
        local = np.zeros((len(solo.batch.agents), solo.S, solo.FIELDS))
        for j, a in enumerate(solo.batch.agents):
            a.update_planner(None, solo.predictions_for(a.id))
            sel, _ = a.step_interface(t)
            cart = sel if isinstance(sel, list) else sel[0]
            for i, st in enumerate(cart[a.replanning_counter:][:solo.S]):
                local[j, i] = (st.position[0], st.position[1], st.orientation, st.velocity, 1.0)
        solo.plans = solo._exchange(local)
        solo.time_step += 1
    assert np.array_equal(solo.plans, sim.plans)
    # every agent sees the four others (and obstacle 2, which is no agent) as predictions
    p = sim.predictions_for(60000)
    assert set(p) == {1, 2, 4, 5, 7} and all(len(v["pos_list"]) >= 28 for v in p.values())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = crx.read_scenario_json(FIXTURE)
        sim, winners = _run_sim(sc, 4, engine_factory=OracleEngine)
        q.put((rank, [a.id for a in sim.batch.agents], sim.plans.copy(), winners))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_agent_sharding_world2_gloo(scenario):
    """Agents round-robin over two ranks, one all-gather of the planned trajectories per step: every rank ends with
    the plans a single process computes."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, _ = _run_sim(scenario, 4, engine_factory=OracleEngine)
    assert got[0][1] == [60000, 4, 7] and got[1][1] == [1, 5]
    assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][2], single.plans)


def _worker_split(rank, world, port, q, n_agents):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = crx.read_scenario_json(FIXTURE)
        sim, winners = _run_sim(sc, 4, engine_factory=OracleEngine, number_of_agents=n_agents - 1)
        q.put((rank, [a.id for a in sim.batch.agents], sim.batch.parts, sim.plans.copy(), winners, sim.batch.launches))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n_agents", [(2, 1), (3, 2)])
def test_fewer_agents_than_ranks_split_their_candidates_gloo(scenario, world, n_agents):
    """BASELINE config 4's shape (5 agents on 8 GPUs) in small: ONE agent over two ranks, two agents over three -- every rank
    carries a part of an agent's candidates in the closed loop (distributed.hybrid_assignment), the parts' winners meet in one
    all-gather, every replica re-evaluates the winning candidate: winners and plans are those of a single process, on every
    rank, and nobody idles."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_split, args=(r, world, port, q, n_agents)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=500) for _ in range(world)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single, w_single = _run_sim(scenario, 4, engine_factory=OracleEngine, number_of_agents=n_agents - 1)
    ids = single.agent_ids
    assert len(ids) == n_agents
    for r, agents, parts, plans, winners, launches in got:
        assert len(agents) == 1 and np.array_equal(plans, single.plans)     # every rank has exactly one item, same plans
        k = ids.index(agents[0])
        assert [w[0] for w in winners] == [w[k] for w in w_single]            # the replica chose the global winner every step
    split = [g for g in got if g[2][0][1] > 1]
    assert len(split) >= 2 and all(g[5] == 4 for g in split)                 # two plan steps: part launch + winner launch each
    assert sorted(g[2][0] for g in got if g[1][0] == ids[0]) == [(p, len([1 for h in got if h[1][0] == ids[0]])) for p in
                                                                 range(len([1 for h in got if h[1][0] == ids[0]]))]


def _worker_split_gpu(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # (RCCL refuses two ranks on one device)
    try:
        sc = crx.read_scenario_json(FIXTURE)
        sim, winners = _run_sim(sc, 6, number_of_agents=0)
        q.put((rank, sim.batch.parts, sim.plans.copy(), winners, sim.batch.launches))
        sim.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_one_agent_split_over_two_processes_on_the_engine(scenario):
    """The closed loop with ONE agent whose candidates are split over two ranks, on the real engine (two processes on cuda:0,
    exchange over gloo): the replicas stay bit-identical and equal a single process, plan step after plan step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_split_gpu, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=500) for _ in range(2)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single, w_single = _run_sim(scenario, 6, number_of_agents=0)
    try:
        assert [g[1] for g in got] == [[(0, 2)], [(1, 2)]]
        assert np.array_equal(got[0][2], got[1][2])                       # replicas bit-identical
        assert np.abs(got[0][2] - single.plans).max() < 1e-9              # and equal to the unsplit run
        assert got[0][3] == got[1][3] == w_single
        assert got[0][4] == got[1][4] == 4                                # two plan steps x (part launch + winner launch)
    finally:
        single.close()


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_closed_loop_on_the_engine_matches_the_oracle_engine(scenario):
    """Config 4 closed loop, 9 simulation steps (3 plan steps per agent): same winners, states within 1e-6."""
    ora, w_ora = _run_sim(scenario, 9, engine_factory=OracleEngine)
    for groups in (1, 2, None):   # one batched launch; two engine contexts, pipelined; the automatic choice (two, for five agents)
        hip, w_hip = _run_sim(scenario, 9, pipeline_groups=groups)
        try:
            assert w_hip == w_ora
            assert np.abs(hip.plans - ora.plans).max() < 1e-6
            n_groups = len(hip.batch.engines)
            assert n_groups == (groups or 2)
            assert hip.batch.launches == 3 * n_groups       # one batched launch per plan step and group
        finally:
            hip.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_closed_loop_at_the_bench_grid_matches_the_oracle_engine(scenario):
    """BASELINE config 4 as bench.py --workload config4 runs it: the dense 19 x 23 x 23 (+ d0) grid = 10 488 candidates per agent,
    five agents, nine simulation steps (three plan steps per agent) -- same winners as the closed loop on the oracle-backed engine
    at every step, states within 1e-6."""
    cfg = lambda: PlannerConfig(sampling_min=0, sampling_max=1, dense_grid=(19, 23, 23))
    ora, w_ora = _run_sim(scenario, 9, engine_factory=OracleEngine, config=cfg())
    hip, w_hip = _run_sim(scenario, 9, config=cfg())
    try:
        sizes = {int(a.planner.last_step.n_candidates) for a in hip.batch.agents if a.planner.last_step is not None}
        assert sizes and sizes <= {10488, 10488 - 19 * 23}, sizes   # (d0 may coincide with a lateral sample: 19 x 23 x 23)
        assert w_hip == w_ora
        assert np.abs(hip.plans - ora.plans).max() < 1e-6
        assert hip.batch.launches == 3 * len(hip.batch.engines)
    finally:
        hip.close()


@pytest.mark.gpu
def test_config4_sized_batch(scenario):
    """Sampling level 4 (10 x 33 x 34 = 11 220 candidates per agent), five agents in one launch vs one agent at a time."""
    cfg = PlannerConfig(sampling_min=4, sampling_max=5)
    sim = multiagent.MultiAgentSimulation(scenario, config=cfg, pipeline_groups=1)   # (all five on one engine context)
    try:
        preds = {a.id: sim.predictions_for(a.id) for a in sim.batch.agents}
        for a in sim.batch.agents:
            a.update_planner(None, preds[a.id])
        inputs = [a.begin_step() for a in sim.batch.agents]
        assert all(i.n_candidates == 11220 for i in inputs)
        batch = sim.batch.engine.plan_batch(inputs)
        costs = [sim.batch.engine.costs(j) for j in range(len(inputs))]
        for j, inp in enumerate(inputs):
            one = sim.batch.engine.plan_step(inp)
            c, f = sim.batch.engine.costs(0)
            assert one["best_index"] == batch[j]["best_index"] and one["best_cost"] == pytest.approx(batch[j]["best_cost"], rel=1e-12)
            assert one["n_feasible"] == batch[j]["n_feasible"] and one["n_collisions"] == batch[j]["n_collisions"]
            # the batched launch may split a candidate's horizon over a different number of lanes than the single
            # launch (auto-tuning by total wave count): cost sums agree to rounding, decisions exactly
            assert np.array_equal(f, costs[j][1]) and np.allclose(c, costs[j][0], rtol=1e-12, atol=0)
    finally:
        sim.close()


def test_shared_prediction_packing_equals_per_agent_packing(scenario):
    """MultiAgentSimulation packs the step's predictions ONCE (all non-agent obstacles and all agents) and hands every agent the
    rows without its own: the same arrays as packing the agent's own dict -- up to the padded length P -- and a dict view that is
    the reference's predictions dict when somebody (the logger) reads it."""
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls
    from frenetix_motion_planner_amd.problem import PackedPredictions, pack_predictions
    sim, _ = _run_sim(scenario, 4, engine_factory=OracleEngine)   # plans exist: the agents predict each other
    for a in sim.batch.agents:
        pp = sim.packed_predictions_for(a.id)
        want = pack_predictions(sim.predictions_for(a.id), sim.S, build_obstacle_hulls)
        got = pp.packed
        # the agent's own entry is still a row of the shared arrays, muted: zero predictions, zero hulls
        j = got["muted_row"]
        assert isinstance(pp, PackedPredictions) and got["K"] == want["K"] + 1 == len(pp) + 1 and got["P"] >= want["P"]
        assert got["npred"][j] == 0 and got["nhull"][j] == 0
        keep = np.arange(got["K"]) != j
        P = want["P"]
        assert np.array_equal(got["npred"][keep], want["npred"]) and np.array_equal(got["nhull"][keep], want["nhull"])
        assert np.array_equal(got["pos"][keep][:, :P], want["pos"]) and np.array_equal(got["cov_inv"][keep][:, :P], want["cov_inv"])
        assert np.array_equal(got["hull"][keep][:, :P - 1], want["hull"])
        for k in np.nonzero(keep)[0]:   # beyond an obstacle's own predictions nothing is stored
            assert not got["pos"][k, got["npred"][k]:].any() and not got["hull"][k, got["nhull"][k]:].any()
        d = sim.predictions_for(a.id)
        assert list(pp.keys()) == list(d) and a.id not in pp and all(np.array_equal(pp[k]["pos_list"], d[k]["pos_list"]) for k in d)
    # the closed loop with and without the shared packing takes the same decisions
    sim2 = multiagent.MultiAgentSimulation(scenario, engine_factory=OracleEngine)
    sim2.shared_packing = False
    for _ in range(4):
        sim2.step()
    assert np.array_equal(sim2.plans, sim.plans)


@pytest.mark.gpu
def test_shared_packing_on_the_engine_equals_per_agent_packing_within_tolerance(scenario):
    """On the device a muted row is never `present`, so the step's presence mask is never full and the prediction sum runs per
    obstacle instead of four terms per reciprocal (fx_walk.h): against packing every agent's own dict the cost sums differ in the
    last bits -- the decisions (winners) are the same, the planned states agree to the parity tolerance, and that is all
    `packed_predictions_for` promises (its docstring)."""
    a, w_a = _run_sim(scenario, 9)
    b = multiagent.MultiAgentSimulation(scenario)
    try:
        b.shared_packing = False
        w_b = []
        for _ in range(9):
            b.step()
            w_b.append([t.optimal_trajectory.global_id if t.optimal_trajectory is not None else -1 for t in b.batch.agents])
        assert w_a == w_b
        assert np.abs(a.plans - b.plans).max() < 1e-9
    finally:
        a.close()
        b.close()


def test_pipelined_groups_equal_one_batch_cpu(scenario):
    """AgentBatchHip over several engine contexts (pipeline_groups: prepare and launch group after group, consume in the same
    order) takes the decisions of the single batched launch: the agents are independent given the step's frozen predictions"""
    from tests.oracle_engine import PackagingOracleEngine
    one, w_one = _run_sim(scenario, 7, engine_factory=PackagingOracleEngine)
    for groups in (2, 3, 5):
        sim, w = _run_sim(scenario, 7, engine_factory=PackagingOracleEngine, pipeline_groups=groups)
        assert len(sim.batch.engines) == groups and sorted(k for ks in sim.batch.groups for k in ks) == list(range(len(sim.batch.agents)))
        assert w == w_one and np.array_equal(sim.plans, one.plans)
        for a, b in zip(sim.batch.agents, one.batch.agents):
            assert a.replanning_counter == b.replanning_counter and np.array_equal(a.x_0.position, b.x_0.position)


def test_shifted_plans_equal_windows_cpu(scenario):
    """steps in which nobody replans take the plans of the step before, shifted by one state: the same arrays (and history) as
    reading every agent's window off its stored trajectory"""
    from tests.oracle_engine import PackagingOracleEngine
    a = multiagent.MultiAgentSimulation(scenario, engine_factory=PackagingOracleEngine)
    b = multiagent.MultiAgentSimulation(scenario, engine_factory=PackagingOracleEngine)
    b.shift_plans = False
    for step in range(10):
        a.step(); b.step()
        assert np.array_equal(a.plans, b.plans), step
    for aid in a.agent_ids:
        assert len(a.history[aid]) == len(b.history[aid]) and all(np.array_equal(x, y) for x, y in zip(a.history[aid], b.history[aid]))
