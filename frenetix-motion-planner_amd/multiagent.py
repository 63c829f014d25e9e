"""Batched multi-agent stepping: every agent's plan step of a simulation step in ONE launch per GPU (SURVEY.md 8 f4,
BASELINE configs 4 and 5).

The reference runs agents in separate processes -- `AgentBatch.run / step_simulation / _step_agents`
(cr_scenario_handler/simulation/agent_batch.py:86-189) under `Simulation._step_parallel_simulation`
(simulation.py:621-663): each process loops over its agents and calls `agent.step_agent`, results come back pickled
through Queues, and the main process merges the new trajectories into the next step's scenario / predictions.  Agents
are independent given the predictions frozen at the start of a step, so here

    * `AgentBatchHip`        -- all agents of this rank share one FrenetEngine context with `max_agents` slots; their
                                first sampling level is evaluated in one batched launch (`fx_evaluate` over a grid of
                                (workgroups, agents)); an agent whose level finds nothing escalates on its own,
    * `MultiAgentSimulation` -- closed loop over a scenario: agent selection as simulation.py:168-211, predictions =
                                recorded futures of non-agent obstacles + the planned trajectories of the other agents
                                (what the reference obtains by writing the plans back into the scenario), agents
                                dealt to the ranks by `distributed.hybrid_assignment` -- round-robin when there are at
                                least as many agents as ranks (the reference's agent batches), otherwise (BASELINE
                                config 4: 5 agents on 8 GPUs) every rank still gets one item: the ranks of an agent
                                split its candidates into contiguous parts, their winners meet in one small
                                all-gather and every replica re-evaluates the one winning candidate, so no GPU idles
                                --, and ONE all-gather per simulation step of the planned trajectories
                                [items_per_rank][S][5] so every rank builds the same predictions for the next step.

Out of scope: goal/collision status bookkeeping of `Agent`, visualisation, evaluation, the prediction
network -- only the stepping that feeds the hot path.
"""
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from .commonroad_xml import GoalState, PlanningProblem, Scenario, State
from .distributed import hybrid_assignment, merge_agent_parts, shard_range
from .frenet_interface import FrenetPlannerInterfaceHip
from .problem import PackedPredictions, VehicleParams, pack_predictions
from .reactive_planner import PlannerConfig


def select_agent_obstacles(scenario: Scenario, number_of_agents: int = -1) -> List[int]:
    """simulation.py:168-211: dynamic cars that move more than 10 m, start on exactly one lanelet and end on one."""
    ids = []
    for oid, ob in scenario.obstacles.items():
        if ob.role != "dynamic" or ob.obstacle_type != "car" or not ob.state_list:
            continue
        if np.linalg.norm(ob.initial_state.position - ob.state_list[-1].position) <= 10:
            continue
        if len(scenario.lanelets_at(ob.initial_state.position)) != 1:
            continue
        if len(scenario.lanelets_at(ob.state_list[-1].position)) == 0:
            continue
        ids.append(oid)
    if -1 < number_of_agents < len(ids):
        ids = ids[:number_of_agents]
    return ids


def planning_problem_for_obstacle(scenario: Scenario, obstacle_id: int) -> PlanningProblem:
    """simulation.py:213-330 in short: initial state = the obstacle's, goal = the lanelet of its final state (the one
    whose direction fits the final orientation best), time +-20 steps, velocity +-2 m/s around the final state."""
    ob = scenario.obstacles[obstacle_id]
    fin = ob.state_list[-1]
    cands = scenario.lanelets_at(fin.position)

    def heading_error(lid):
        c = scenario.lanelets[lid].center_vertices
        k = int(np.argmin(np.linalg.norm(c - fin.position, axis=1)))
        k = min(max(k, 0), len(c) - 2)
        th = np.arctan2(c[k + 1, 1] - c[k, 1], c[k + 1, 0] - c[k, 0])
        return abs((th - fin.orientation + np.pi) % (2 * np.pi) - np.pi)

    lid = min(cands, key=heading_error)
    goal = GoalState(time_interval=(max(fin.time_step - 20, 0), fin.time_step + 20),
                     velocity_interval=(fin.velocity - 2, fin.velocity + 2), lanelet_ids=[lid])
    init = State(time_step=ob.initial_state.time_step, position=np.array(ob.initial_state.position, dtype=np.float64),
                 orientation=ob.initial_state.orientation, velocity=ob.initial_state.velocity,
                 acceleration=ob.initial_state.acceleration, yaw_rate=ob.initial_state.yaw_rate)
    return PlanningProblem(obstacle_id, init, [goal])


def trajectory_as_prediction(states: Sequence, shape: dict, wb_rear_axle: float, first: int = 0) -> dict:
    """A planned trajectory (rear-axle states) as a prediction of the other agents: centre positions
    (state.py:30-39), orientation, covariance 0.1 I (prediction_helpers.py:245)."""
    st = states[first:]
    th = np.array([s.orientation for s in st])
    pos = np.array([s.position for s in st]).reshape(-1, 2) + wb_rear_axle * np.stack([np.cos(th), np.sin(th)], axis=1)
    n = len(st)
    return dict(pos_list=pos, cov_list=np.tile(np.eye(2) * 0.1, (n, 1, 1)), orientation_list=th,
                v_list=np.array([s.velocity for s in st]), shape=dict(shape))


class AgentBatchHip:
    """The agents of one rank stepped together: one batched launch for every agent that plans in this step.

    parts: per agent (part, n_parts) -- n_parts > 1: this rank holds one of n_parts replicas of the agent and evaluates the
    contiguous part `part` of its candidates (distributed.hybrid_assignment); slots: the agent's index in the simulation;
    winner_exchange: collective callable([(slot, cost, global index), ...]) -> per-slot global (cost, index) -- called
    once per step by every rank when any agent of the simulation is split."""

    def __init__(self, agents: List[FrenetPlannerInterfaceHip], max_candidates: int, device: int = 0, max_ref_knots: int = 4096,
                 max_obstacles: int = 64, engine=None, parts=None, slots=None, winner_exchange=None, pipeline_groups: Optional[int] = None):
        """pipeline_groups: the agents are dealt to that many engine contexts (contiguous groups); a planning step prepares group
        0's inputs, launches it, prepares group 1's while group 0 evaluates, ... and consumes the results in the same order -- the
        host work of one group runs in the shadow of another group's kernels (planner-sized grids are host-bound: BASELINE config 4).
        None: 2 for four or more agents with planner-sized grids (<= 40 000 candidates each), else 1; an injected `engine`, a
        split agent (parts) or an engine without plan_batch_begin means one group."""
        import os
        self.agents = list(agents)
        self.parts = list(parts) if parts is not None else [(0, 1)] * len(self.agents)
        self.slots = list(slots) if slots is not None else list(range(len(self.agents)))
        self.winner_exchange = winner_exchange
        N = max(a.planner.N for a in self.agents) if self.agents else 30
        n = len(self.agents)
        if pipeline_groups is None:
            env = os.environ.get("FX_PIPELINE_GROUPS")
            pipeline_groups = int(env) if env else (2 if n >= 4 and max_candidates <= 40_000 else 1)
        if isinstance(engine, (list, tuple)):   # one engine object per group, handed in (tests)
            pipeline_groups = len(engine)
        elif engine is not None:
            pipeline_groups = 1
        if winner_exchange is not None or any(p[1] > 1 for p in self.parts):
            pipeline_groups = 1
        pipeline_groups = max(1, min(int(pipeline_groups), max(n, 1)))
        # contiguous groups of (almost) equal size: agent k belongs to group self.group_of[k] as its local agent self.local_of[k]
        # (the larger groups first: an early launch has the other groups' host work to hide behind)
        bounds = [-((-g * n) // pipeline_groups) for g in range(pipeline_groups + 1)]
        sizes = os.environ.get("FX_PIPELINE_SIZES")   # experiments: explicit group sizes, e.g. "1,2,2"
        if sizes and engine is None and pipeline_groups > 1 and sum(int(x) for x in sizes.split(",")) == n:
            bounds = [0]
            for x in sizes.split(","):
                bounds.append(bounds[-1] + int(x))
            pipeline_groups = len(bounds) - 1
        self.groups = [list(range(bounds[g], bounds[g + 1])) for g in range(pipeline_groups)]
        self.group_of = [g for g, ks in enumerate(self.groups) for _ in ks]
        self.local_of = [j for ks in self.groups for j in range(len(ks))]
        if engine is None:
            from .engine import FrenetEngine
            # a context's capacity is the TOTAL over its agent slots (fx_create_batch)
            self.engines = [FrenetEngine(max_candidates=max_candidates * max(len(ks), 1), max_steps=N, max_ref_knots=max_ref_knots,
                                         max_obstacles=max_obstacles, max_pred_steps=max(64, N + 2), device=device,
                                         max_agents=max(len(ks), 1)) for ks in self.groups]
        else:
            self.engines = list(engine)[:pipeline_groups] if isinstance(engine, (list, tuple)) else [engine]
        self.engine = self.engines[0]
        for e in self.engines:
            if hasattr(e, "set_package"):   # every agent's winner arrives packaged with the result block: no read-back per agent
                e.set_package(True)
        self.launches = 0
        self.escalations = 0
        self.last_batch_ms = 0.0

    def _step_pipelined(self, global_timestep: int, predictions: Dict[int, dict]) -> Dict[int, Optional[list]]:
        """step() over several engine contexts: prepare and launch group after group, then consume in the same order."""
        out: Dict[int, Optional[list]] = {}
        t0 = time.time()
        launched, passive = [], []
        for g, ks in enumerate(self.groups):
            planning = []
            for k in ks:
                a = self.agents[k]
                if a.id in predictions or a.needs_plan():
                    a.update_planner(None, predictions.get(a.id, {}))
                inp = a.begin_step()
                (planning if inp is not None else passive).append((a, inp))
            if planning:
                eng = self.engines[g]
                token = eng.plan_batch_begin([inp for _, inp in planning])
                launched.append((eng, planning, token))
                self.launches += 1
        for eng, planning, token in launched:
            yaws = [a.planner.x_0.yaw_rate for a, _ in planning]
            if token is not None:
                results, packages = eng.plan_batch_end(token, yaws)
            else:   # a batch the packaged path does not take (sampling matrix, mixed horizons): the general call
                results, packages = eng.plan_batch_packaged([inp for _, inp in planning], yaws)
            for j, (a, inp) in enumerate(planning):
                p = a.planner
                best = p.plan_consume(inp, results[j], eng, j, package=packages[j])
                if best is None and p._sampling_min + 1 < p._sampling_max:
                    self.escalations += 1
                    pair = p.plan_escalate(t0)
                else:
                    pair = p.plan_finish(best, t0)
                sel, _ = a.finish_step(pair, global_timestep)
                out[a.id] = sel
        if launched:
            self.last_batch_ms = (time.time() - t0) * 1e3
        for a, _ in passive:
            sel, _ = a.finish_step(None, global_timestep)
            out[a.id] = sel[0] if sel is not None else None
        return out

    def step(self, global_timestep: int, predictions: Dict[int, dict]) -> Dict[int, Optional[list]]:
        """agent_batch.py:140-189: update every agent with its predictions, plan (batched), finish the step.
        predictions: {agent id: predictions dict of that agent}.  Returns {agent id: selected Cartesian state list
        (None: no trajectory found)}."""
        import copy
        if len(self.engines) > 1:
            return self._step_pipelined(global_timestep, predictions)
        planning, passive = [], []
        for k, a in enumerate(self.agents):
            if a.id in predictions or a.needs_plan():
                a.update_planner(None, predictions.get(a.id, {}))
            inp = a.begin_step()
            if inp is not None and self.parts[k][1] > 1:   # this replica's contiguous part of the agent's candidates
                inp.shard = shard_range(inp.n_candidates_global, *self.parts[k])
            (planning if inp is not None else passive).append((a, inp, k))
        out: Dict[int, Optional[list]] = {}
        t0 = time.time()
        results = []
        packages = None
        if planning:
            if hasattr(self.engine, "plan_batch_packaged") and getattr(self.engine, "packaging", False):
                # one call across the boundary: states rewritten in place, evaluation, results, every winner packaged
                results, packages = self.engine.plan_batch_packaged([inp for _, inp, _ in planning],
                                                                    [a.planner.x_0.yaw_rate for a, _, _ in planning])
            else:
                results = self.engine.plan_batch([inp for _, inp, _ in planning])
            self.launches += 1
            self.last_batch_ms = (time.time() - t0) * 1e3

        def conclude(a, inp, res, j, packages=packages):
            p = a.planner
            best = (p.plan_consume(inp, res, self.engine, j, package=packages[j] if packages is not None else False)
                    if res is not None else None)
            if best is None and p._sampling_min + 1 < p._sampling_max:
                self.escalations += 1
                pair = p.plan_escalate(t0)   # the whole next level on the planner's own engine (identical on every replica)
            else:
                pair = p.plan_finish(best, t0)
            sel, _ = a.finish_step(pair, global_timestep)
            out[a.id] = sel

        split = [(j, a, inp, k) for j, (a, inp, k) in enumerate(planning) if self.parts[k][1] > 1]
        for j, (a, inp, k) in enumerate(planning):
            if self.parts[k][1] == 1:
                conclude(a, inp, results[j], j)
        if self.winner_exchange is not None:
            # the parts of a split agent meet: ONE all-gather of (slot, cost, global index) per rank item, every rank takes
            # the same lexicographic minimum; then every replica evaluates the one winning candidate (the same launch on
            # every replica, so their states stay bit-identical)
            winners = self.winner_exchange([(self.slots[k], results[j]["best_cost"], results[j]["best_index"]) for j, a, inp, k in split])
            again = []
            for j, a, inp, k in split:
                c, g = winners[self.slots[k]]
                if g >= 0:
                    one = copy.copy(inp)
                    one.shard = (int(g), 1)
                    again.append((a, one))
                else:
                    conclude(a, inp, None, j)
            if again:
                res1 = self.engine.plan_batch([one for _, one in again])
                self.launches += 1
                for j1, (a, one) in enumerate(again):
                    conclude(a, one, res1[j1], j1, None)
        for a, _, _ in passive:
            sel, _ = a.finish_step(None, global_timestep)
            out[a.id] = sel[0] if sel is not None else None
        return out

    def close(self):
        for e in self.engines:
            e.close()
        for a in self.agents:
            a.close()


class MultiAgentSimulation:
    """Closed-loop multi-agent run of a scenario on the batched engine (BASELINE config 4)."""

    FIELDS = 5  # x, y (rear axle), orientation, velocity, valid

    def __init__(self, scenario: Scenario, config: Optional[PlannerConfig] = None, vehicle: Optional[VehicleParams] = None,
                 number_of_agents: int = -1, sampling_level: Optional[int] = None, device: int = 0, group=None,
                 use_road_boundary: bool = False, max_candidates: Optional[int] = None, engine_factory=None,
                 pipeline_groups: Optional[int] = None, freeze_gc: bool = False):
        """engine_factory: callable returning an engine object (tests inject a stand-in); default = FrenetEngine.
        pipeline_groups: see AgentBatchHip (None = automatic; with an engine_factory: that many injected engines, default one).
        freeze_gc: take what exists once the simulation is set up out of the garbage collector's generations (gc.freeze(): a
        PROCESS-WIDE setting, so it is off unless asked for -- bench.py and the timing tools ask; close() undoes it)."""
        import torch.distributed as dist
        self.scenario = scenario
        self.config = config or PlannerConfig()
        if sampling_level is not None:
            self.config.sampling_min, self.config.sampling_max = sampling_level, sampling_level + 1
        self.vehicle = vehicle or VehicleParams()
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.group = group
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        # agents: the scenario's planning problems first, then the selected obstacles (simulation.py:131-166)
        self.problems: Dict[int, PlanningProblem] = dict(scenario.planning_problems)
        self.agent_obstacles = select_agent_obstacles(scenario, number_of_agents)
        for oid in self.agent_obstacles:
            self.problems[oid] = planning_problem_for_obstacle(scenario, oid)
        self.agent_ids = list(self.problems)
        self.shapes = {i: (dict(length=scenario.obstacles[i].length, width=scenario.obstacles[i].width)
                           if i in scenario.obstacles else dict(length=self.vehicle.length, width=self.vehicle.width))
                       for i in self.agent_ids}
        # work items of every rank: whole agents round-robin, or -- fewer agents than ranks -- one part of an agent's candidates
        self.items = hybrid_assignment(len(self.agent_ids), self.world)
        my_items = self.items[self.rank]
        self.my_slots = [k for k, _, _ in my_items]
        self.split = any(n_parts > 1 for it in self.items for _, _, n_parts in it)
        mine = [FrenetPlannerInterfaceHip(self.agent_ids[k], scenario, self.problems[self.agent_ids[k]], config=self._cfg(),
                                          vehicle=self.vehicle, device=device, use_road_boundary=use_road_boundary,
                                          engine=engine_factory() if engine_factory else None)
                for k in self.my_slots]
        # per-agent capacity from the sampling sets the planners can really reach (the time set is not bounded by 16 values)
        cap = max_candidates or max([4096] + [a.planner.max_candidates_per_step() for a in mine])
        self.batch = AgentBatchHip(mine, max_candidates=cap, device=device, pipeline_groups=pipeline_groups,
                                   engine=(([engine_factory() for _ in range(pipeline_groups)] if pipeline_groups and pipeline_groups > 1
                                            else engine_factory()) if engine_factory else None),
                                   parts=[(part, n_parts) for _, part, n_parts in my_items], slots=self.my_slots,
                                   winner_exchange=self._exchange_winners if self.split else None)
        self.S = self.batch.agents[0].planner.N + 1 if mine else int(self.config.planning_horizon / self.config.dt) + 1
        self.time_step = 0
        # planned trajectories of ALL agents, identical on every rank after the exchange: [agents][S][FIELDS]
        self.plans = np.zeros((len(self.agent_ids), self.S, self.FIELDS))
        self.history: Dict[int, list] = {i: [] for i in self.agent_ids}
        self._shared = None
        self._shared_packed = None
        self.shared_packing = True   # predictions packed once per step for all agents (packed_predictions_for)
        self.shift_plans = True      # steps in which nobody replans: the plans are the previous ones shifted by one state
        self._cov_tiles: Dict[int, np.ndarray] = {}
        # the scenario, the planners and their reference paths live as long as the simulation: taken out of the garbage
        # collector's generations, so that the full collections a closed loop triggers every few hundred steps walk the step's
        # own objects only (0.6 ms pauses inside a 0.35 ms planning step otherwise -- tools/seg_config4.py, per-call maxima)
        self._froze_gc = False
        if freeze_gc:
            import gc
            gc.collect()
            gc.freeze()
            self._froze_gc = True

    def _cfg(self) -> PlannerConfig:
        import copy
        return copy.deepcopy(self.config)

    # -- predictions of one agent: everything but itself ---------------------------------------------------------
    def _shared_predictions(self):
        """What every agent's predictions have in common at this time step, built ONCE per simulation step: the recorded
        futures of the non-agent obstacles and one entry per agent (its plan, or its recorded future before the first plan)."""
        t = self.time_step
        if self._shared is not None and self._shared[0] == t:
            return self._shared[1], self._shared[2]
        horizon = self.S - 1
        others = [o for o in self.scenario.obstacles if o not in self.problems]
        base = {k: v for k, v in self.scenario.ground_truth_predictions(t, horizon, obstacle_ids=others).items() if len(v["pos_list"])}
        own = {}
        wb = self.vehicle.wb_rear_axle
        # the box centres of ALL agents' plans in three array operations (state.py:30-39), windows handed out per agent
        th_all = np.ascontiguousarray(self.plans[:, :, 2])
        pos_all = np.empty(self.plans.shape[:2] + (2,))
        pos_all[:, :, 0] = self.plans[:, :, 0] + wb * np.cos(th_all)
        pos_all[:, :, 1] = self.plans[:, :, 1] + wb * np.sin(th_all)
        vel_all = np.ascontiguousarray(self.plans[:, :, 3])
        n_valid = np.count_nonzero(self.plans[:, :, 4] > 0, axis=1)
        for k, aid in enumerate(self.agent_ids):
            rows = self.plans[k]
            n = int(n_valid[k])
            if n and rows[n - 1, 4] > 0:   # the valid rows are a prefix (what lies ahead of the agent): views of the shared arrays
                th, pos, vel = th_all[k, :n], pos_all[k, :n], vel_all[k, :n]
            elif n:
                valid = rows[:, 4] > 0
                r = rows[valid]
                th = r[:, 2].copy()
                pos = np.empty((n, 2))
                pos[:, 0] = r[:, 0] + wb * np.cos(th)
                pos[:, 1] = r[:, 1] + wb * np.sin(th)
                vel = r[:, 3].copy()
            if n:
                cov = self._cov_tiles.get(n)
                if cov is None:   # covariance 0.1 I per step (prediction_helpers.py:245): one read-only block per length
                    cov = self._cov_tiles[n] = np.tile(np.eye(2) * 0.1, (n, 1, 1))
                    cov.setflags(write=False)
                own[aid] = dict(pos_list=pos, cov_list=cov, orientation_list=th, v_list=vel, shape=dict(self.shapes[aid]))
            elif aid in self.scenario.obstacles:  # no plan yet: the recorded future (first step)
                gt = self.scenario.ground_truth_predictions(t, horizon, obstacle_ids=[aid])[aid]
                if len(gt["pos_list"]):
                    own[aid] = gt
        self._shared = (t, base, own)
        return base, own

    def packed_predictions_for(self, agent_id: int):
        """The same predictions as `predictions_for`, packed: ALL entries of the step (non-agent obstacles, then the agents in
        order) are packed once -- covariance inverses and OBB-sum hulls included -- and every agent reads the SAME position,
        covariance and hull arrays; only the two count arrays are its own, with its own row set to zero predictions and zero
        hulls.  An obstacle without predictions contributes to no step of the prediction cost (collision_probability.py:287,
        `i < len(pos_list)`) and is skipped by the collision stage (collision_check.py:165-168), so the step's decisions are those
        of the dict without the agent; the dict view (`predictions_for`) is built when somebody reads it.
        Not bit for bit: a row that is never present keeps the device's per-step presence mask from ever being full, so the
        prediction sum is accumulated one obstacle per reciprocal instead of four (fx_walk.h) -- costs differ from packing the
        agent's own dict in the last bits (within the parity tolerance; the GPU test
        test_shared_packing_on_the_engine_equals_per_agent_packing_within_tolerance holds winners equal and states to 1e-9)."""
        t = self.time_step
        sh = getattr(self, "_shared_packed", None)
        if sh is None or sh[0] != t:
            base, own = self._shared_predictions()
            allp = dict(base)
            for aid in self.agent_ids:
                if aid in own:
                    allp[aid] = own[aid]
            packed = pack_predictions(allp, self.S, self._hull_builder()) if allp else None
            sh = self._shared_packed = (t, {k: j for j, k in enumerate(allp)}, packed)
        _, row_of, packed = sh
        if packed is None:
            return PackedPredictions(PackedPredictions.subset({}, []), lambda: {})
        j = row_of.get(agent_id)
        if j is None:
            return PackedPredictions(packed, lambda: self.predictions_for(agent_id))
        npred, nhull = packed["npred"].copy(), packed["nhull"].copy()
        npred[j] = nhull[j] = 0
        return PackedPredictions(dict(K=packed["K"], P=packed["P"], pos=packed["pos"], cov_inv=packed["cov_inv"], npred=npred,
                                      hull=packed["hull"], nhull=nhull, muted_row=j),
                                 lambda: self.predictions_for(agent_id))

    def _hull_builder(self):
        """what the agents' own planners pack with: the engine library's host geometry, or the engine object's own builder
        (tests inject an oracle-backed engine)"""
        hb = getattr(self.batch.engine, "hull_builder", None)
        if hb is not None:
            return hb
        from .engine import build_obstacle_hulls
        return build_obstacle_hulls

    def predictions_for(self, agent_id: int) -> dict:
        base, own = self._shared_predictions()
        preds = dict(base)
        for aid in self.agent_ids:   # agent order, as before
            if aid != agent_id and aid in own:
                preds[aid] = own[aid]
        return preds

    def _all_gather(self, buf: np.ndarray) -> np.ndarray:
        """[world] + buf.shape: every rank's buffer (one collective; nccl = RCCL on device tensors, gloo on the host)."""
        if self.world == 1:
            return buf[None]
        import torch
        t = torch.from_numpy(np.ascontiguousarray(buf))
        if self.dist.get_backend(self.group) == "nccl":
            t = t.cuda()
            g = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(g, t, group=self.group)
            return g.cpu().numpy()
        gl = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(gl, t, group=self.group)
        return torch.stack(gl).numpy()

    def _exchange_winners(self, rows) -> list:
        """Winners of the parts of split agents: ONE all-gather of (slot, cost, global index) per rank item; every rank takes
        the same lexicographic (cost, index) minimum per agent (distributed.merge_agent_parts)."""
        per = max(len(it) for it in self.items)
        buf = np.full((per, 3), -1.0)
        for j, row in enumerate(rows):
            buf[j] = row
        return merge_agent_parts(len(self.agent_ids), self._all_gather(buf).reshape(-1, 3))

    def _exchange(self, local_rows: np.ndarray) -> np.ndarray:
        """ONE all-gather per simulation step: [items_per_rank][S][FIELDS] from every rank -> plans of all agents (the replicas
        of a split agent hold the same plan: part 0's is taken)."""
        if self.world == 1 and len(local_rows) == len(self.agent_ids) and not self.split:
            return local_rows   # one rank holds every agent whole, in order: its rows are the plans
        per = max(len(it) for it in self.items)
        buf = np.zeros((per, self.S, self.FIELDS))
        buf[:len(local_rows)] = local_rows
        gathered = self._all_gather(buf)
        plans = np.zeros_like(self.plans)
        for r in range(self.world):
            for j, (k, part, _) in enumerate(self.items[r]):
                if part == 0:
                    plans[k] = gathered[r, j]
        return plans

    def step(self) -> Dict[int, Optional[list]]:
        """One simulation step of every agent (simulation.py:621-663 + agent_batch.py:140-189)."""
        # predictions are only read by agents that replan in this step (two of three steps just advance, :261-277)
        preds = {a.id: (self.packed_predictions_for(a.id) if self.shared_packing else self.predictions_for(a.id))
                 for a in self.batch.agents if a.needs_plan()}
        selected = self.batch.step(self.time_step, preds)
        if not preds and self.shift_plans and self.world == 1 and not self.split and len(self.batch.agents) == len(self.agent_ids):
            # nobody replanned: every agent moved one state along its stored trajectory, so what lies ahead of it is what lay
            # ahead a step ago without its first row -- one shift for all agents instead of a window per agent
            plans = np.zeros_like(self.plans)
            plans[:, :-1] = self.plans[:, 1:]
            self.plans = plans
            self._record_history()
            self.time_step += 1
            return selected
        local = np.zeros((len(self.batch.agents), self.S, self.FIELDS))
        for j, a in enumerate(self.batch.agents):
            sel = selected.get(a.id)
            if sel is None:
                continue
            # the part of the stored trajectory that lies ahead of the agent's new state (index 1 + counter - 1)
            rows = getattr(sel, "rows", None)
            if rows is not None:   # packaged trajectory: the block's columns, no state objects
                ahead = rows[a.replanning_counter:][:self.S]
                local[j, :len(ahead), :4] = ahead
                local[j, :len(ahead), 4] = 1.0
                continue
            ahead = sel[a.replanning_counter:]
            for i, st in enumerate(ahead[:self.S]):
                local[j, i] = (st.position[0], st.position[1], st.orientation, st.velocity, 1.0)
        self.plans = self._exchange(local)
        self._record_history()
        self.time_step += 1
        return selected

    def _record_history(self):
        """every agent's new state (x, y, orientation, velocity) of this step: rows of ONE copy of the plans' first states"""
        first = self.plans[:, 0, :].copy()
        for k, aid in enumerate(self.agent_ids):
            if first[k, 4] > 0:
                self.history[aid].append(first[k, :4])

    def run(self, n_steps: int):
        for _ in range(n_steps):
            self.step()
        return self.history

    def close(self):
        self.batch.close()
        if self._froze_gc:   # give the frozen objects back to the collector: a process that builds many simulations must not
            import gc        # accumulate their cyclic garbage in the permanent generation
            gc.unfreeze()
            self._froze_gc = False
