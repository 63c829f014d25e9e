"""Duck-typed stand-in for FrenetEngine backed by the CPU oracle -- TEST CODE ONLY.

The product's host layers (ShardedEvaluator, ReactivePlannerHip, AgentBatchHip, MultiAgentSimulation) only see an
object with the engine's methods, so on a machine without a GPU their logic (sharding, exchange, replanning, batching)
runs against the oracle; on the GPU box the same scripts run against the real engine and the two are compared.
"""
import numpy as np


class OracleEngine:
    def __init__(self):
        self.last = None
        self.max_agents = 1 << 20

    def set_stream(self, _):
        pass

    def close(self):
        pass

    def plan_step(self, inp):
        from oracle import oracle
        out = oracle.plan_step(inp, want_planes=True)
        self.last = [(inp, out)]
        return dict(out["result"])

    def plan_batch(self, inps):
        from oracle import oracle
        self.last = [(i, oracle.plan_step(i, want_planes=True)) for i in inps]
        return [dict(o["result"]) for _, o in self.last]

    def topk(self, k):
        cost = np.full((len(self.last), k), np.inf)
        idx = np.full((len(self.last), k), -1, np.int64)
        for a, (inp, out) in enumerate(self.last):
            ok = out["selectable"] & ~out["collision"] & ~out["boundary"]
            ids = np.nonzero(ok)[0]
            order = ids[np.lexsort((ids, out["cost"][ids]))][:k]
            cost[a, :len(order)] = out["cost"][order]
            idx[a, :len(order)] = order + inp.shard_begin
        return cost, idx

    # -- read-back surface of FrenetEngine --
    def costs(self, agent=0):
        out = self.last[agent][1]
        return out["cost"].copy(), out["flags"].copy()

    def boundary_steps(self, agent=0):
        return self.last[agent][1]["boundary_step"].copy()

    def costmap(self, agent=0):
        return self.last[agent][1]["costmap"].copy()

    def coeffs(self, index, agent=0):
        out = self.last[agent][1]
        return out["coeff_lon"][index].copy(), out["coeff_lat"][index].copy(), int(out["traj_len"][index])

    def sample(self, index, agent=0):
        return self.last[agent][1]["planes"][index].copy()

    def plane(self, name_or_index, agent=0):
        from frenetix_motion_planner_amd import _abi
        p = _abi.PLANE_INDEX[name_or_index] if isinstance(name_or_index, str) else int(name_or_index)
        return np.ascontiguousarray(self.last[agent][1]["planes"][:, p, :].T)
