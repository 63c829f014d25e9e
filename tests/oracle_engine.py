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
        return (out["coeff_lon"][index].copy(), out["coeff_lat"][index].copy(), int(out["traj_len"][index]),
                float(out["tau_lat"][index]))

    def sample(self, index, agent=0):
        return self.last[agent][1]["planes"][index].copy()

    def plane(self, name_or_index, agent=0):
        from frenetix_motion_planner_amd import _abi
        p = _abi.PLANE_INDEX[name_or_index] if isinstance(name_or_index, str) else int(name_or_index)
        return np.ascontiguousarray(self.last[agent][1]["planes"][:, p, :].T)


class _OraclePackage:
    """what engine.WinnerPackage carries, from the oracle's arrays (fx_read_package's derived rows restated in NumPy:
    planner.py:394-447)"""

    def __init__(self, inp, out, g, yaw_rate0):
        from frenetix_motion_planner_amd import _abi
        planes = out["planes"][g]
        S = planes.shape[1]
        block = np.empty((_abi.FX_PKG_ROWS, S))
        block[:_abi.FX_NUM_PLANES] = planes
        theta, kappa = planes[2], planes[5]
        block[_abi.PKG_ROW_YAW_RATE, 0] = yaw_rate0
        block[_abi.PKG_ROW_YAW_RATE, 1:] = (theta[1:] - theta[:-1]) / inp.dt
        block[_abi.PKG_ROW_STEERING] = np.arctan2(inp.vehicle.wheelbase * kappa, 1.0)
        lo, hi = inp.x0_orientation - np.pi, inp.x0_orientation + np.pi
        o = theta.copy()
        for _ in range(4):
            o = np.where(o < lo, o + 2 * np.pi, o)
            o = np.where(o > hi, o - 2 * np.pi, o)
        block[_abi.PKG_ROW_ORIENTATION] = o
        self.block, self.index = block, int(g) + inp.shard_begin
        self.cost, self.flags, self.traj_len = float(out["cost"][g]), int(out["flags"][g]), int(out["traj_len"][g])
        self.lon, self.lat = out["coeff_lon"][g].copy(), out["coeff_lat"][g].copy()
        self.tau_lat = float(out["tau_lat"][g])
        self._raw = out["costmap"][g].copy() if inp.write_costmap else None

    @property
    def raw_costs(self):
        return self._raw

    def raw_cost_list(self):
        return None if self._raw is None else self._raw.tolist()

    @property
    def planes(self):
        from frenetix_motion_planner_amd import _abi
        return self.block[:_abi.FX_NUM_PLANES]


class PackagingOracleEngine(OracleEngine):
    """OracleEngine with the winner-package surface of FrenetEngine (set_package / package / plan_batch_packaged /
    plan_step_packaged), so that the packaged paths of the planner and of the agent batch run on the CPU; `engine_s`
    accumulates the time spent inside the engine calls (host-overhead measurements subtract it)."""

    def __init__(self):
        super().__init__()
        self.packaging = False
        self.engine_s = 0.0

    def set_package(self, enabled):
        self.packaging = bool(enabled)

    def package(self, agent=0, yaw_rate0=0.0):
        inp, out = self.last[agent]
        g = out["result"]["best_index"]
        return _OraclePackage(inp, out, g - inp.shard_begin, yaw_rate0) if g >= 0 else None

    def plan_batch_packaged(self, inps, yaw_rates):
        import time
        t0 = time.perf_counter()
        res = self.plan_batch(inps)
        pk = [self.package(a, yaw_rates[a]) for a in range(len(inps))]
        self.engine_s += time.perf_counter() - t0
        return res, pk

    def plan_batch_begin(self, inps):
        import time
        t0 = time.perf_counter()
        res = self.plan_batch(list(inps))
        self.engine_s += time.perf_counter() - t0
        return (list(inps), res)

    def plan_batch_end(self, token, yaw_rates):
        inps, res = token
        return res, [self.package(a, yaw_rates[a]) for a in range(len(inps))]

    def plan_step_packaged(self, inp, yaw_rate0=0.0):
        import time
        t0 = time.perf_counter()
        res = self.plan_step(inp)
        pk = self.package(0, yaw_rate0)
        self.engine_s += time.perf_counter() - t0
        return res, pk
