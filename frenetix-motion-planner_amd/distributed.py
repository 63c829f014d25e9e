"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL (backend "nccl").

The reference has no collective at all -- its only parallelism is multiprocessing fan-out of candidate chunks
(reactive_planner.py:197-224) and of agent batches (simulation.py:449-470, agent_batch.py:186-189), with
results pickled back through Queues.  Candidates are independent given the shared inputs and agents are
independent given the frozen predictions, so the MI355X-native form is:

  * every rank receives the same (small) shared inputs,
  * candidate sharding: rank r evaluates the contiguous global range shard_range(C, r, W) -- contiguous so
    that the (cost, global index) tie-break of the stable sort (trajectories.py:560) is preserved,
  * agent sharding: rank r evaluates agents a with a % W == r in one batched launch,
  * ONE all-gather per plan step of each rank's k best collision-free survivors (cost f64, global index i64)
    -- 16*k bytes per rank per agent, latency-bound over xGMI --
  * every rank then takes the same lexicographic (cost, index) minimum.

k > 1 survivors are exchanged so the host-side road-boundary walk (planner.py:362-390) can step past a
rejected winner without another evaluation.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np


def exit_on_timeout(fn, *args, code: int = 3, **kwargs):
    """Run one collective step; a time-out of the exchange (FxTimeoutError from the library's bounded waits, or the process
    group's own time-out on the torch.distributed path) ends THIS PROCESS with a non-zero code after saying why -- the
    reference's processes give up after TIMEOUT = 20 s the same way (simulation.py:637,655, agent_batch.py:98).  The stuck
    stream cannot be recovered in-process; a supervisor may start a fresh child (never re-exec a process that touched the GPU)."""
    import os
    import sys
    from ._lib import FxTimeoutError
    try:
        return fn(*args, **kwargs)
    except FxTimeoutError as e:
        print(f"[fxplan] collective step timed out: {e}", file=sys.stderr, flush=True)
        os._exit(code)
    except RuntimeError as e:
        if "imed out" in str(e) or "timeout" in str(e).lower():
            print(f"[fxplan] collective step timed out: {e}", file=sys.stderr, flush=True)
            os._exit(code)
        raise


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """(begin, count) of rank's contiguous share of n items; the first n % world ranks get one more."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(int(n), world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


def agents_of_rank(n_agents: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_agents, world))


def hybrid_assignment(n_agents: int, world: int) -> List[List[Tuple[int, int, int]]]:
    """Work items of every rank: a list over ranks of (agent, part, n_parts) triples.

    n_agents >= world: agents round-robin over the ranks, every item the whole agent (part 0 of 1) -- the reference's agent
    batches (simulation.py:449-470).  n_agents < world (BASELINE config 4: 5 agents on 8 GPUs): every rank still gets exactly
    one item; the ranks are dealt to the agents as evenly as possible (the first `world % n_agents` agents get one more) and
    the ranks of an agent split its candidates into contiguous parts (shard_range), so no GPU idles."""
    if n_agents < 1 or world < 1:
        raise ValueError("bad n_agents/world")
    if n_agents >= world:
        return [[(a, 0, 1) for a in agents_of_rank(n_agents, r, world)] for r in range(world)]
    base, extra = divmod(world, n_agents)
    items: List[List[Tuple[int, int, int]]] = []
    for a in range(n_agents):
        n_parts = base + (1 if a < extra else 0)
        items.extend([[(a, p, n_parts)] for p in range(n_parts)])
    return items


def merge_agent_parts(n_agents: int, gathered) -> List[Tuple[float, int]]:
    """gathered: iterable of (agent, cost, global index) rows from every rank's items (index < 0: that part found nothing).
    Returns per agent the lexicographic (cost, index) minimum -- (0.0, -1) when no part of the agent has a survivor."""
    best: List[Optional[Tuple[float, int]]] = [None] * n_agents
    for a, c, i in gathered:
        a, i = int(a), int(i)
        if a < 0 or i < 0:
            continue
        if best[a] is None or (c, i) < best[a]:
            best[a] = (float(c), i)
    return [b if b is not None else (0.0, -1) for b in best]


def merge_survivors(cost: np.ndarray, index: np.ndarray):
    """Lexicographic (cost, index) minimum over gathered survivors; index < 0 marks an empty slot.
    Returns (best_cost, best_index, order) with order = all valid survivors sorted."""
    cost = np.asarray(cost, dtype=np.float64).reshape(-1)
    index = np.asarray(index, dtype=np.int64).reshape(-1)
    if cost.size <= 256:  # the per-step case (W x k entries): plain tuples beat array machinery by 4x
        pairs = sorted((c, i) for c, i in zip(cost.tolist(), index.tolist()) if i >= 0)
        if not pairs:
            return 0.0, -1, np.zeros(0, np.int64)
        return pairs[0][0], pairs[0][1], np.array([i for _, i in pairs], dtype=np.int64)
    ok = index >= 0
    if not ok.any():
        return 0.0, -1, np.zeros(0, np.int64)
    c, i = cost[ok], index[ok]
    order = np.lexsort((i, c))
    return float(c[order[0]]), int(i[order[0]]), i[order]


class _RawStepResult:
    """Dict-like view of one FxResult (winner and counters are already in host memory; nothing is converted until read)."""

    def __init__(self, raw):
        self._raw = raw

    def __getitem__(self, key):
        if key == "global_best_index":
            key = "best_index"
        elif key == "global_best_cost":
            key = "best_cost"
        v = getattr(self._raw, key)
        return list(v) if key == "reason_hist" else v

    def get(self, key, default=None):
        try:
            return self[key]
        except AttributeError:
            return default

    def as_dict(self) -> dict:
        d = self._raw.as_dict()
        d["global_best_index"], d["global_best_cost"] = d["best_index"], d["best_cost"]
        return d


class _ExchangedStepResult(_RawStepResult):
    """local FxResult + the global winner of the in-library exchange"""

    def __init__(self, raw, best_cost, best_index, survivors):
        super().__init__(raw)
        self._g = {"global_best_cost": best_cost, "global_best_index": best_index, "survivors": survivors}

    def __getitem__(self, key):
        if key in self._g:
            return self._g[key]
        return getattr(self._raw, key) if key != "reason_hist" else list(self._raw.reason_hist)

    def as_dict(self) -> dict:
        d = self._raw.as_dict()
        d.update(self._g)
        return d


class ShardedEvaluator:
    """Candidate-sharded plan step.  `engine` is a FrenetEngine (or anything with plan_step / topk /
    topk_to_device / set_stream); `group` a torch.distributed process group (None = default group, or
    no distribution at all when torch.distributed is not initialised)."""

    def __init__(self, engine, k: int = 8, group=None, force_exchange: bool = False):
        import torch
        import torch.distributed as dist
        self.engine, self.k, self.group = engine, int(k), group
        self.force_exchange = bool(force_exchange)  # run the survivor exchange even with a single rank (tests)
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        backend = self.dist.get_backend(group) if self.dist else None
        self.on_device = bool(self.dist) and backend == "nccl" and torch.cuda.is_available()
        self.torch = torch
        if self.on_device:
            dev = torch.device("cuda", torch.cuda.current_device())
            # the engine enqueues on torch's current stream so RCCL sees the survivors in stream order
            engine.set_stream(torch.cuda.current_stream().cuda_stream)
            # one buffer per rank: [cost f64 x k | index i64 x k] so the exchange is a single all-gather of 16*k bytes
            self._surv = torch.empty(2 * self.k, dtype=torch.float64, device=dev)
            self._gath = torch.empty(self.world * 2 * self.k, dtype=torch.float64, device=dev)
            if self.k == 1:  # the selection kernel itself leaves (cost, index) in the exchange buffer: no top-k launch
                engine.set_winner_buffer(self._surv.data_ptr())
        # k = 1: the exchange runs inside the library on a communicator of the engine's own (fx_step_exchange: evaluation,
        # all-gather, publication and result in ONE call, no Python between the launches); FX_EXCHANGE=torch keeps the
        # torch.distributed path
        self.lib_exchange = False
        self.lib_exchange_agents = False
        import os
        if self.on_device and self.k == 1 and hasattr(engine, "comm_init") and (self.world > 1 or self.force_exchange) \
                and os.environ.get("FX_EXCHANGE", "lib") != "torch":
            self.lib_exchange = self._init_library_exchange(dev)

    def _init_library_exchange(self, dev) -> bool:
        """Every rank ends with the same answer.  Nothing collective of the LIBRARY is entered before the ranks have agreed, over
        the torch group, that every one of them can: (1) local preconditions (RCCL present, capacity -- fx_comm_check) MIN-reduced;
        (2) rank 0 draws the id (a flag byte says whether it could) and broadcasts it; (3) only then every rank calls
        ncclCommInitRank -- a rank that would have failed before it would leave its peers inside that call forever;
        (4) the outcome is MIN-reduced once more (a rank whose initialisation failed AFTER the rendezvous)."""
        torch, dist = self.torch, self.dist

        def all_agree(ok: bool) -> bool:
            if self.world == 1:
                return ok
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return bool(flag.item())

        ok = hasattr(self.engine, "comm_check") and bool(self.engine.comm_check(self.world))
        if not all_agree(ok):
            return False
        uid = torch.zeros(129, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            try:
                raw = self.engine.comm_unique_id()
                uid = torch.tensor(list(raw) + [1], dtype=torch.uint8, device=dev)
            except Exception:
                pass
        if self.world > 1:
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(uid, src=src, group=self.group)
        host = uid.cpu().numpy()
        if not bool(host[128]):   # the same byte on every rank
            return False
        try:
            self.engine.comm_init(bytes(host[:128]), self.rank, self.world)
            ok = True
        except Exception:
            ok = False
        all_ok = all_agree(ok)
        if ok and not all_ok:
            self.engine.comm_destroy()
        # (5) the element count of the library's all-gathers is the communicator's: one agent row per rank for the candidate
        # split; setup_agents agrees on the agents per rank
        return all_ok and self._agree_agent_rows(1)

    def _agree_agent_rows(self, n_rows: int) -> bool:
        """Fix the agent rows every rank contributes to the library's exchanges (fx_comm_set_agents) -- only if EVERY rank asks
        for the same number: ranks that entered an all-gather with different counts would hang or corrupt memory.  Returns
        False (library exchange off on every rank, communicator destroyed) otherwise."""
        lo, hi = self._agree_min(int(n_rows)), -self._agree_min(-int(n_rows))
        ok = lo == hi == int(n_rows)
        if ok:
            try:
                self.engine.comm_set_agents(int(n_rows))
            except Exception:
                ok = False
        if not self._agree_min(1 if ok else 0):
            try:
                self.engine.comm_destroy()
            except Exception:
                pass
            return False
        return True

    def shard(self, inputs):
        begin, count = shard_range(inputs.n_candidates_global, self.rank, self.world)
        inputs.shard = (begin, count) if self.world > 1 else None
        return inputs

    def step_enqueued(self) -> dict:
        """Evaluate + exchange for inputs that are already uploaded (bench.py's timed step)."""
        if self.world == 1 and not self.force_exchange:
            if hasattr(self.engine, "step_raw"):   # one call across the boundary; the result stays a C struct
                return _RawStepResult(self.engine.step_raw()[0])
            self.engine.evaluate()
            res = self.engine.finish()[0]
            res["global_best_index"], res["global_best_cost"] = res["best_index"], res["best_cost"]
            return res
        if self.lib_exchange:
            raw, gc, gi = self.engine.step_exchange_raw()
            best_c, best_i, order = merge_survivors(gc[:, 0], gi[:, 0])
            return _ExchangedStepResult(raw[0], best_c, best_i, order)
        self.engine.evaluate()
        self._enqueue_exchange()
        # the local result block is published by the evaluation kernel itself: unpack it while the all-gather runs
        res = self.engine.finish()[0]
        gc, gi = self._collect_exchange()
        best_c, best_i, order = merge_survivors(gc, gi)
        res["global_best_cost"], res["global_best_index"], res["survivors"] = best_c, best_i, order
        return res

    def _enqueue_exchange(self):
        """top-k straight into the torch buffer (torch's stream), ONE RCCL all-gather, publication of W*16*k bytes to
        pinned host memory -- all enqueued behind the evaluation kernel, nothing waited for."""
        k = self.k
        if k > 1:
            self.engine.topk_to_device(k, self._surv.data_ptr(), self._surv.data_ptr() + 8 * k)
        self.dist.all_gather_into_tensor(self._gath, self._surv, group=self.group)
        # the engine copies the gathered block into pinned host memory on the same stream and the host polls for it
        self.engine.publish(self._gath.data_ptr(), self.world * 2 * k)

    def _collect_exchange(self):
        g = self.engine.wait_published().reshape(self.world, 2, self.k)
        return g[:, 0, :].copy(), g[:, 1, :].copy().view(np.int64)

    def _exchange_on_device(self):
        self._enqueue_exchange()
        return self._collect_exchange()

    def plan_step(self, inputs) -> dict:
        """Evaluate this rank's shard, exchange survivors, return the global winner (same on all ranks)."""
        self.shard(inputs)
        if self.world == 1 and not self.force_exchange:
            res = self.engine.plan_step(inputs)
            res["global_best_index"], res["global_best_cost"] = res["best_index"], res["best_cost"]
            return res
        if self.lib_exchange:
            self.engine.upload(inputs)
            return self.step_enqueued().as_dict()
        if self.on_device:
            self.engine.upload(inputs)
            self.engine.evaluate()
            gc, gi = self._exchange_on_device()
            res = self.engine.finish()[0]
        else:
            res = self.engine.plan_step(inputs)
            c, i = self.engine.topk(self.k)
            tc = self.torch.from_numpy(np.ascontiguousarray(c[0]))
            ti = self.torch.from_numpy(np.ascontiguousarray(i[0]))
            gcl = [self.torch.empty_like(tc) for _ in range(self.world)]
            gil = [self.torch.empty_like(ti) for _ in range(self.world)]
            self.dist.all_gather(gcl, tc, group=self.group)
            self.dist.all_gather(gil, ti, group=self.group)
            gc = self.torch.cat(gcl).numpy()
            gi = self.torch.cat(gil).numpy()
        best_c, best_i, order = merge_survivors(gc, gi)
        res["global_best_cost"], res["global_best_index"], res["survivors"] = best_c, best_i, order
        return res


    # ---- the library exchange is trusted only after it has agreed with the torch.distributed exchange once ----
    def uses_library_exchange(self) -> bool:
        return bool(self.lib_exchange or self.lib_exchange_agents)

    def _agree_min(self, value: int) -> int:
        """MIN of an integer over the ranks (torch group; a device tensor under nccl)."""
        if self.dist is None or self.world == 1:
            return int(value)
        dev = self.torch.device("cuda", self.torch.cuda.current_device()) if self.on_device else self.torch.device("cpu")
        t = self.torch.tensor([int(value)], dtype=self.torch.int32, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def _exchange_once(self, library: bool):
        """one step on the resident inputs through ONE of the two exchanges; returns what it reports as the survivors"""
        if self.lib_exchange_agents and getattr(self, "n_local", 0):
            keep = self.lib_exchange_agents
            self.lib_exchange_agents = keep if library else False
            try:
                _, surv = self.step_agents_enqueued()
            finally:
                self.lib_exchange_agents = keep
            return None if surv is None else (np.array(surv[0]), np.array(surv[1]))
        keep = self.lib_exchange
        self.lib_exchange = keep if library else False
        try:
            r = self.step_enqueued()
        finally:
            self.lib_exchange = keep
        return (np.array([r["global_best_cost"]]), np.array([r["global_best_index"]]))

    @staticmethod
    def _same_survivors(a, b) -> bool:
        return a is not None and b is not None and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])

    def crosscheck_exchange(self) -> int:
        """One step exchanged through the library and once more through torch.distributed, on the resident inputs: 1 = both
        agree on every rank, 0 = they differ somewhere (the library exchange is switched off on every rank), -1 = the library
        exchange did not come back within the time bound on some rank (the context is lost).  Collective: every rank calls it.
        The library's direct mode -- the all-gather received straight in the pinned block, a stream-ordered write instead of the
        publication launch (fx_set_exchange_mode 1) -- is tried first and kept only if every rank saw it agree; otherwise the
        device-buffer mode is checked the same way; otherwise torch.distributed carries the exchange.

        Every rank issues the SAME sequence of collectives whatever happens locally: per mode (1) the ranks agree that the mode
        could be set everywhere -- a rank on which it is refused never lets its peers enter the library's all-gather alone --,
        (2) the library half, whose local failures (the library enters its all-gather first and reports afterwards:
        fx_step_exchange*) are caught and MIN-reduced, (3) the torch.distributed half only if the library half succeeded on EVERY
        rank -- a rank that failed in (2) would otherwise go straight to the agreement round while its peers sit in torch's
        all-gather: mismatched collectives on one group --, (4) the agreement on the comparison.  An error inside (3) is not
        swallowed: it may have been raised ahead of torch's collective, the job ends as it did before this check existed."""
        from ._lib import FxError, FxTimeoutError
        if not self.uses_library_exchange():
            return 1
        import os
        can_set = hasattr(self.engine, "set_exchange_mode")
        modes = [1, 0] if can_set and os.environ.get("FX_EXCHANGE_MODE", "1") != "0" else [0]
        state = 0
        for mode in modes:
            ok = 1
            if can_set:
                try:
                    self.engine.set_exchange_mode(mode)
                except (FxError, ValueError):   # this mode is not available here (e.g. the stream-ordered write is refused)
                    ok = 0
            if self._agree_min(ok) == 0:
                state = 0
                continue
            a, ok = None, 1
            try:
                a = self._exchange_once(library=True)
            except FxTimeoutError:
                ok = -1
            except (FxError, ValueError):   # local failure, reported by the library AFTER its all-gather: the peers are not waiting
                ok = 0
            first = self._agree_min(ok)
            if first < 0:
                return -1   # some rank's library exchange ran out of time: its context is lost, nobody goes on
            if first == 0:
                state = 0
                continue
            b = self._exchange_once(library=False)
            state = self._agree_min(1 if self._same_survivors(a, b) else 0)
            if state == 1:
                self.exchange_mode = mode
                break
        if state == 0:   # wrong answers or failures somewhere in every mode: nobody uses the library exchange
            self.lib_exchange = self.lib_exchange_agents = False
            try:
                self.engine.comm_destroy()
            except Exception:
                pass
        return state

    # ---- agent sharding with a top-k survivor gather (BASELINE config 5) ----
    def setup_agents(self, n_local_agents: int):
        """Exchange buffers for `n_local_agents` agents per rank: [cost k | index k] per agent, one all-gather of
        n_local * 16 * k bytes per rank and step."""
        self.n_local = int(n_local_agents)
        if self.on_device:
            dev = self.torch.device("cuda", self.torch.cuda.current_device())
            n = self.n_local * self.k
            self._asurv = self.torch.empty(2 * n, dtype=self.torch.float64, device=dev)
            self._agath = self.torch.empty(self.world * 2 * n, dtype=self.torch.float64, device=dev)
            self.engine.set_winner_buffer(0)
            # the per-agent top-k exchange inside the library (fx_step_exchange_topk) on a communicator of the engine's own;
            # FX_EXCHANGE=torch keeps the torch.distributed path
            import os
            if hasattr(self.engine, "step_exchange_topk_raw") and (self.world > 1 or self.force_exchange) \
                    and os.environ.get("FX_EXCHANGE", "lib") != "torch" and self.world * self.n_local * 2 * self.k <= 16384:
                self.lib_exchange_agents = self.lib_exchange or self._init_library_exchange(dev)
                if self.lib_exchange_agents and not self._agree_agent_rows(self.n_local):
                    self.lib_exchange = self.lib_exchange_agents = False

    def step_agents_enqueued(self, exchange: bool = True):
        """One batched launch over this rank's (already uploaded) agents, per-agent top-k on the device, ONE all-gather
        of every rank's survivors, published to the host.  Returns (results of the local agents, survivors
        (cost [W, n_local, k], index [W, n_local, k]) or None)."""
        if exchange and self.lib_exchange_agents:
            # evaluation, per-agent top-k, ONE all-gather and the publication enqueued by the library back to back
            # (a rank that uploaded more agents than the ranks agreed on still enters the collective -- with "no survivor" rows --
            # and gets FX_ERR_CAPACITY afterwards: fx_step_exchange_topk)
            res, gc, gi = self.engine.step_exchange_topk_raw(self.k)
            return [r.as_dict() for r in res] if hasattr(res[0], "as_dict") else res, (gc, gi)
        self.engine.evaluate()
        surv = None
        if exchange:
            n = self.n_local * self.k
            if self.on_device:
                base = self._asurv.data_ptr()
                self.engine.topk_to_device(self.k, base, base + 8 * n)
                if self.world > 1 or self.force_exchange:
                    self.dist.all_gather_into_tensor(self._agath, self._asurv, group=self.group)
                    src, w = self._agath, self.world
                else:
                    src, w = self._asurv, 1
                self.engine.publish(src.data_ptr(), w * 2 * n)
                g = self.engine.wait_published().reshape(w, 2, self.n_local, self.k)
                surv = (g[:, 0].copy(), g[:, 1].copy().view(np.int64))
            else:
                # host tensors (gloo, or one rank without a device group): the same rows, gathered with torch.distributed on the host
                c, i = self.engine.topk(self.k)
                if self.world > 1:
                    tc = self.torch.from_numpy(np.ascontiguousarray(c, dtype=np.float64))
                    ti = self.torch.from_numpy(np.ascontiguousarray(i, dtype=np.int64))
                    gcl = [self.torch.empty_like(tc) for _ in range(self.world)]
                    gil = [self.torch.empty_like(ti) for _ in range(self.world)]
                    self.dist.all_gather(gcl, tc, group=self.group)
                    self.dist.all_gather(gil, ti, group=self.group)
                    surv = (self.torch.stack(gcl).numpy(), self.torch.stack(gil).numpy())
                else:
                    surv = (c[None], i[None])
        res = self.engine.finish()
        return res, surv

    def plan_agents(self, agent_inputs: Sequence) -> List[Optional[dict]]:
        """Agent sharding with the candidate split for small agent counts (hybrid_assignment): every rank evaluates its items
        -- whole agents, or its contiguous part of an agent's candidates -- in ONE batched launch; ONE all-gather of
        (agent, cost, global index) per item; every rank takes the same lexicographic minimum per agent.  Returns a list over
        ALL agents of dicts {best_cost, best_index} (global winner), plus the full local result under "local" for the
        agents (or agent parts) this rank evaluated."""
        import copy
        n = len(agent_inputs)
        items = hybrid_assignment(n, self.world)
        mine = items[self.rank]
        local_inputs = []
        for a, part, n_parts in mine:
            inp = agent_inputs[a]
            if n_parts > 1:
                inp = copy.copy(inp)  # shallow: the arrays are shared, only the shard differs
                inp.shard = shard_range(agent_inputs[a].n_candidates_global, part, n_parts)
            local_inputs.append(inp)
        local = self.engine.plan_batch(local_inputs) if mine else []
        per = max(len(it) for it in items)
        buf = np.full((per, 3), -1.0)
        for j, (a, _, _) in enumerate(mine):
            buf[j] = (a, local[j]["best_cost"], local[j]["best_index"])
        if self.world == 1:
            gathered = buf
        else:
            t = self.torch.from_numpy(buf)
            if self.on_device:
                t = t.cuda()
                g = self.torch.empty((self.world * per, 3), dtype=t.dtype, device=t.device)
                self.dist.all_gather_into_tensor(g, t, group=self.group)
                gathered = g.cpu().numpy()
            else:
                gl = [self.torch.empty_like(t) for _ in range(self.world)]
                self.dist.all_gather(gl, t, group=self.group)
                gathered = self.torch.cat(gl).numpy()
        winners = merge_agent_parts(n, gathered)
        out: List[Optional[dict]] = [dict(best_cost=c, best_index=i) for c, i in winners]
        for j, (a, part, n_parts) in enumerate(mine):
            out[a]["local"] = local[j]
            out[a]["part"] = (part, n_parts)
            if n_parts == 1:
                out[a] = dict(local[j], **out[a])
        return out


def verified_evaluator(make_engine, k: int, prepare):
    """(engine, evaluator, note) with an exchange that has been SEEN to agree: `make_engine()` -> engine, `prepare(engine,
    evaluator)` uploads the inputs (and calls setup_agents where needed).  With several ranks the library-side exchange
    (fx_step_exchange / fx_step_exchange_topk on the engine's own RCCL communicator) is cross-checked once against the
    torch.distributed exchange of the same step before anything is timed; if the two differ on any rank the library exchange
    is switched off on every rank.  A library exchange that never comes back raises FxTimeoutError after the context's time
    bound (a collective stuck on the device cannot be recovered from inside the process: callers leave through
    exit_on_timeout).  Collective: every rank calls it."""
    eng = make_engine()
    ev = ShardedEvaluator(eng, k=k)
    prepare(eng, ev)
    if not ev.uses_library_exchange():
        return eng, ev, ("torch.distributed" if ev.world > 1 else "none (one rank)")
    state = ev.crosscheck_exchange()
    if state < 0:
        from ._lib import FxTimeoutError
        raise FxTimeoutError("the library-side exchange did not come back in the cross-check")
    how = {1: "received straight in the pinned block", 0: "device receive buffer + publication kernel"}.get(getattr(ev, "exchange_mode", 0))
    return eng, ev, (f"library (RCCL communicator of the engine, {how}), cross-checked against torch.distributed" if state == 1 else
                     "torch.distributed (the library exchange disagreed in the cross-check)")
