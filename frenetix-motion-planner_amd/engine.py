"""FrenetEngine: thin object wrapper over the C-ABI (one context = one planner's device state).

Plays the role of `frenetix.TrajectoryHandler` behind reactive_planner_cpp.py:49,255-256,345-353: it takes
the shared inputs of a plan step, runs the fused HIP pipeline and hands back the winner, counters and
lazily-materialised per-candidate data.  Batched form (`plan_batch`) evaluates several agents in one launch
(the GPU analogue of AgentBatch._step_agents, agent_batch.py:186-189).
"""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _abi
from ._lib import check, lib
from .problem import PlanInputs


def build_obstacle_hulls(n_pred, pos, yaw, length, width) -> np.ndarray:
    """OBB-sum hulls of consecutive predicted boxes (host helper of the library)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    yaw = np.ascontiguousarray(yaw, dtype=np.float64)
    out = np.zeros((max(int(n_pred) - 1, 1), 6))
    n = C.c_int32(0)
    check(lib().fx_build_obstacle_hulls(int(n_pred), pos.ctypes.data_as(C.POINTER(C.c_double)),
                                        yaw.ctypes.data_as(C.POINTER(C.c_double)), float(length), float(width),
                                        out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)))
    return out[:n.value]


def _build_obstacle_hulls_batch(n_use, pos, yaw, length, width, hull, nhull):
    """fx_build_obstacle_hulls_batch: pos [K][P][2], yaw [K][P] -> hull [K][P-1][6], nhull [K] (in place)"""
    K, P = pos.shape[0], pos.shape[1]
    length, width = np.ascontiguousarray(length, dtype=np.float64), np.ascontiguousarray(width, dtype=np.float64)
    check(lib().fx_build_obstacle_hulls_batch(K, P, n_use.ctypes.data, pos.ctypes.data, yaw.ctypes.data, length.ctypes.data,
                                              width.ctypes.data, hull.ctypes.data, nhull.ctypes.data))


def invert_cov2(m: np.ndarray) -> np.ndarray:
    """[n, 2, 2] -> [n, 4]: np.linalg.inv of every matrix, bit for bit (fx_invert_cov2); LinAlgError for a singular one"""
    m = np.ascontiguousarray(m, dtype=np.float64).reshape(-1, 4)
    out = np.empty_like(m)
    if lib().fx_invert_cov2(len(m), m.ctypes.data, out.ctypes.data) != 0:
        raise np.linalg.LinAlgError("Singular matrix")
    return out


def _pack_predictions_c(entries, P: int, n_samples: int):
    """fx_pack_predictions: entries = [(pos [n,2], cov [n,2,2], yaw [n] or None, length, width)] -> packed arrays"""
    K = len(entries)
    lens = [len(e[0]) for e in entries]
    i64 = C.c_int64 * K
    if all(e[2] is not None and len(e[2]) == n and len(e[1]) == n for e, n in zip(entries, lens)):
        # the usual case (every obstacle with orientations and a shape): three concatenations, the pointer columns are
        # base + offset -- one .ctypes access per kind instead of one per array
        pos_all = np.concatenate([e[0] for e in entries])
        cov_all = np.concatenate([e[1] for e in entries])
        yaw_all = np.concatenate([e[2] for e in entries])
        bp, bc, by = pos_all.ctypes.data, cov_all.ctypes.data, yaw_all.ctypes.data
        pp, pc, py, off = [], [], [], 0
        for n in lens:
            pp.append(bp + 16 * off); pc.append(bc + 32 * off); py.append(by + 8 * off)
            off += n
    else:
        pp = [e[0].ctypes.data for e in entries]
        pc = [e[1].ctypes.data for e in entries]
        py = [0 if e[2] is None else e[2].ctypes.data for e in entries]
    f64 = C.c_double * K
    sizes = (2 * K * P, 4 * K * P, 6 * K * (P - 1))
    out = np.empty(sum(sizes))   # outputs in one block, handed out as views
    cnt = np.empty((2, K), np.int32)
    o0, c0 = out.ctypes.data, cnt.ctypes.data
    rc = lib().fx_pack_predictions(K, P, int(n_samples), (C.c_int32 * K)(*lens), i64(*pp), i64(*pc), i64(*py),
                                   f64(*[e[3] for e in entries]), f64(*[e[4] for e in entries]),
                                   o0, o0 + 8 * sizes[0], c0, o0 + 8 * (sizes[0] + sizes[1]), c0 + 4 * K)
    if rc != 0:
        if b"singular" in lib().fx_last_error():
            raise np.linalg.LinAlgError("Singular matrix")
        check(rc)
    return dict(K=K, P=P, pos=out[:sizes[0]].reshape(K, P, 2), cov_inv=out[sizes[0]:sizes[0] + sizes[1]].reshape(K, P, 4),
                npred=cnt[0], hull=out[sizes[0] + sizes[1]:].reshape(K, P - 1, 6), nhull=cnt[1])


_FXHOST = None


def _fxhost():
    """the CPython extension `_fxhost` (csrc/fx_host_ext.c, built next to the package by `make`); False when it is not there"""
    global _FXHOST
    if _FXHOST is None:
        try:
            from . import _fxhost as m
            _FXHOST = (m, C.cast(lib().fx_pack_predictions, C.c_void_p).value, C.cast(lib().fx_plan_batch_packaged, C.c_void_p).value,
                       C.cast(lib().fx_plan_batch_begin, C.c_void_p).value, C.cast(lib().fx_plan_batch_end, C.c_void_p).value)
        except ImportError:
            _FXHOST = False
    return _FXHOST


def _pack_predictions_dict(predictions: dict, n_samples: int, max_obstacles: int):
    """The whole predictions dict in one call: walked in C (buffer protocol), packed by fx_pack_predictions.  None when some
    entry is not a contiguous float64 array (the caller converts and takes the general path)."""
    h = _fxhost()
    if not h:
        return None
    try:
        r = h[0].pack_predictions(h[1], predictions, int(n_samples), int(max_obstacles))
    except ValueError as e:
        if b"singular" in lib().fx_last_error():
            raise np.linalg.LinAlgError("Singular matrix") from e
        raise
    if r is None:
        return None
    K, P, out, cnt = r
    o = np.frombuffer(out, dtype=np.float64)
    c = np.frombuffer(cnt, dtype=np.int32)
    n_pos, n_cov = 2 * K * P, 4 * K * P
    return dict(K=K, P=P, pos=o[:n_pos].reshape(K, P, 2), cov_inv=o[n_pos:n_pos + n_cov].reshape(K, P, 4), npred=c[:K],
                hull=o[n_pos + n_cov:].reshape(K, P - 1, 6), nhull=c[K:])


build_obstacle_hulls.pack = _pack_predictions_c
build_obstacle_hulls.pack_dict = _pack_predictions_dict
build_obstacle_hulls.batch = _build_obstacle_hulls_batch
build_obstacle_hulls.invert_cov2 = invert_cov2


def build_boundary_bins(ref_xy, segments, max_len: float, reach: float):
    """fx_build_boundary_bins (host geometry of the library): pieces [n][4], bins [M+1], items."""
    ref = np.ascontiguousarray(ref_xy, dtype=np.float64)
    rx, ry = np.ascontiguousarray(ref[:, 0]), np.ascontiguousarray(ref[:, 1])
    seg = np.ascontiguousarray(segments, dtype=np.float64).reshape(-1, 4)
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    n_piece, n_item = C.c_int32(0), C.c_int32(0)
    piece = np.zeros((max(len(seg), 1) * 2, 4))
    item = np.zeros(max(len(ref) * 16, 1), dtype=np.int32)
    bins = np.zeros(len(ref) + 1, dtype=np.int32)
    for _ in range(3):
        rc = lib().fx_build_boundary_bins(len(ref), rx.ctypes.data_as(pd), ry.ctypes.data_as(pd), len(seg), seg.ctypes.data_as(pd),
                                          float(max_len), float(reach), len(piece), piece.ctypes.data_as(pd), C.byref(n_piece),
                                          bins.ctypes.data_as(pi), len(item), item.ctypes.data_as(pi), C.byref(n_item))
        if rc == 0:
            return piece[:n_piece.value].copy(), bins, item[:n_item.value].copy()
        if n_piece.value > len(piece):
            piece = np.zeros((n_piece.value, 4))
        elif n_item.value > len(item):
            item = np.zeros(n_item.value, dtype=np.int32)
        else:
            check(rc)
    check(rc)


class WinnerPackage:
    """The chosen trajectory as the library packaged it (fx_read_package): `block` [FX_PKG_ROWS][S] = the 14 planes, yaw rate,
    steering angle, shifted heading; coefficients, raw partial costs, cost, flag word, horizon, global index."""

    def __init__(self, pkg: _abi.FxPackage, block: np.ndarray, inputs: PlanInputs):
        self.block = block
        self.index = pkg.index
        self.cost = pkg.cost
        self.flags = pkg.flags
        self.traj_len = pkg.traj_len
        self.tau_lat = pkg.tau_lat   # delta_tau of the lateral polynomial (reactive_planner.py:161-171)
        self._pkg = pkg
        self._costmap = bool(inputs.write_costmap)

    @property
    def lon(self) -> np.ndarray:
        return np.array(self._pkg.coeff_lon)

    @property
    def lat(self) -> np.ndarray:
        return np.array(self._pkg.coeff_lat)

    @property
    def raw_costs(self):
        return np.array(self._pkg.raw_costs[:self._pkg.n_cost]) if self._costmap else None

    def raw_cost_list(self):
        return list(self._pkg.raw_costs[:self._pkg.n_cost]) if self._costmap else None

    @property
    def planes(self) -> np.ndarray:
        return self.block[:_abi.FX_NUM_PLANES]


def math_selftest(x: np.ndarray):
    """(atan, sin, cos) of the device math kernels for the values in x."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    outs = [np.zeros_like(x) for _ in range(3)]
    check(lib().fx_math_selftest(x.size, x.ctypes.data_as(C.POINTER(C.c_double)),
                                 *[o.ctypes.data_as(C.POINTER(C.c_double)) for o in outs]))
    return outs


def device_count() -> int:
    n = C.c_int32(0)
    lib().fx_device_count(C.byref(n))
    return n.value


class FrenetEngine:
    def __init__(self, max_candidates: int, max_steps: int = 30, max_ref_knots: int = 1024, max_obstacles: int = 32,
                 max_pred_steps: int = 64, device: int = 0, max_agents: int = 1):
        self._ctx = C.c_void_p()
        self._inputs: List[PlanInputs] = []
        self._resident_key = None
        self._resident_keys = None
        self.packaging = False
        check(lib().fx_create_batch(C.byref(self._ctx), device, max_agents, int(max_candidates), int(max_steps),
                                    int(max_ref_knots), int(max_obstacles), int(max_pred_steps)))
        self.device = device
        self.max_agents = max_agents

    # -- lifetime --
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            lib().fx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, hip_stream_ptr: int):
        """Run on an external HIP stream (e.g. torch.cuda.current_stream().cuda_stream)."""
        check(lib().fx_set_stream(self._ctx, C.c_void_p(hip_stream_ptr)))

    def set_tuning(self, lanes_per_candidate: int = 0, waves_per_simd: int = 0, kernel_variant: int = 0,
                   block_size: int = 0, mapping: int = 0):
        """Override the automatic work decomposition (0 = automatic; kernel_variant 1 = generic, 2 = grid;
        block_size 64/128/256 lanes for the grid kernel); results are unaffected."""
        check(lib().fx_set_tuning(self._ctx, int(lanes_per_candidate), int(waves_per_simd), int(kernel_variant)))
        check(lib().fx_set_block_size(self._ctx, int(block_size)))
        check(lib().fx_set_part_mapping(self._ctx, int(mapping)))

    def set_store_mode(self, mode: int = 0):
        """Plane stores of the bundle: 0 auto, 1 write-back, 2 write-through (agent scope); results are unaffected."""
        check(lib().fx_set_store_mode(self._ctx, int(mode)))

    def set_winner_buffer(self, d_ptr: int):
        """Device buffer [n_agents][2] (cost f64, global index i64) every step's winner is also written to."""
        check(lib().fx_set_winner_buffer(self._ctx, C.c_void_p(d_ptr)))

    def publish(self, d_ptr: int, n: int):
        """Enqueue the copy of n doubles at device address d_ptr into the pinned publication block."""
        check(lib().fx_publish(self._ctx, C.c_void_p(d_ptr), int(n)))
        self._pub_n = int(n)

    def wait_published(self) -> np.ndarray:
        out = np.empty(self._pub_n)
        check(lib().fx_wait_published(self._ctx, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def set_fused_selection(self, enabled):
        """Selection fused into the evaluation kernel (fx_set_fused_selection; default on): the agent's last workgroup reduces
        the arg-min partials, counts the collisions in front of the winner (agents of at most 8 192 candidates whose obstacle
        stage runs in the evaluation kernel), gathers the winner package and publishes -- one launch per step.  False / 0: always
        the separate selection kernel; 2: in-kernel whatever the candidate count.  Takes effect at the next upload."""
        check(lib().fx_set_fused_selection(self._ctx, 2 if enabled == 2 else int(bool(enabled))))

    def set_obstacle_stage(self, stage: int = 0, steps_per_item: int = 0):
        """Where the obstacle stage runs: 0 auto, 1 fused into the walk, 2 its own (candidate x step)-parallel kernel
        (fx_set_obstacle_stage; steps_per_item 0 auto / 2 / 3 / 5); takes effect at the next upload.  Decisions are unaffected."""
        check(lib().fx_set_obstacle_stage(self._ctx, int(stage), int(steps_per_item)))
        self._resident_key = None   # the decomposition is chosen at upload time
        self._resident_keys = None

    def set_step_kernel(self, mode: int = 0, steps_per_item: int = 0):
        """The whole plan step in ONE launch (fx_set_step_kernel, csrc/fx_step_kernel.h), opt-in: 0 / 1 off (walk, obstacle kernel,
        selection as three launches -- faster on the MI355X), 2 on where applicable; steps_per_item 0 auto / 3 / 5 / 8.  Takes effect
        at the next upload.  Same results (bit for bit at three steps per item)."""
        check(lib().fx_set_step_kernel(self._ctx, int(mode), int(steps_per_item)))
        self._resident_key = None
        self._resident_keys = None

    def step_info(self) -> dict:
        """how the last evaluation was launched (fx_step_info_ex)"""
        v = np.zeros(16, np.int64)
        check(lib().fx_step_info_ex(self._ctx, v.ctypes.data))
        keys = ("grid_kernel", "lanes_per_candidate", "waves_per_simd", "block", "wave_split", "fused_selection", "blocks", "agents",
                "package", "lds_bytes", "obstacle_kernel", "obstacle_steps_per_item", "obstacle_items", "obstacle_lds_bytes",
                "obstacle_workgroup_waves", "tail")
        info = dict(zip(keys, (int(x) for x in v)))
        # [15]: bits 0-1 what the agent's last workgroup did (fx_tail.h), bits 8-9 how the latest inputs reached the device:
        # 1 DMA copy, 2 staging kernel, 3 written by the host into device memory (large BAR)
        info["staging"] = ("none", "dma", "kernel", "host_writes")[(info["tail"] >> 8) & 3]
        info["step_kernel"] = (info["tail"] >> 16) & 1   # the whole step ran as ONE launch (fx_step_kernel.h)
        info["tail"] &= 0xff
        return info

    def obstacle_kernel_times(self, max_n: int = 256):
        """obstacle-kernel ms of the most recent timed steps, oldest first (0 where the stage ran fused)"""
        ob = np.zeros(max_n, dtype=np.float64)
        n = C.c_int32(0)
        check(lib().fx_read_obstacle_kernel_times(self._ctx, max_n, ob.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)))
        return ob[:n.value]

    @property
    def last_obstacle_kernel_ms(self) -> float:
        return float(lib().fx_last_obstacle_kernel_ms(self._ctx))

    TIMING = {"off": 0, "stream": 1, "kernel": 2}

    def set_timing(self, mode, every: int = 1):
        """HIP-event timing of (every n-th) step: "off" / False (default), "stream" / True (stream events around the
        kernels) or "kernel" (events attached to the evaluation kernel itself -- what a kernel trace reports).
        Times are read lazily: `last_kernel_ms`, `last_eval_kernel_ms`, `kernel_times()`."""
        if isinstance(mode, str):
            mode = self.TIMING[mode]
        check(lib().fx_set_timing(self._ctx, int(mode)))
        check(lib().fx_set_timing_interval(self._ctx, int(every)))

    def kernel_times(self, max_n: int = 256):
        """(evaluation-kernel ms, whole-step device ms) of the most recent timed steps, oldest first."""
        ev = np.zeros(max_n, dtype=np.float64)
        st = np.zeros(max_n, dtype=np.float64)
        n = C.c_int32(0)
        check(lib().fx_read_kernel_times(self._ctx, max_n, ev.ctypes.data_as(C.POINTER(C.c_double)),
                                         st.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)))
        return ev[:n.value], st[:n.value]

    # -- plan step, split so callers can overlap host work (upload/evaluate enqueue only) --
    def upload(self, inputs):
        batch = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        self._inputs = batch
        self._resident_key = None
        self._resident_keys = None
        arr = (_abi.FxProblem * len(batch))(*[b.as_struct() for b in batch])
        self._structs = arr  # keep pointers alive until the copy has been enqueued (h2d staging is synchronous memcpy)
        check(lib().fx_upload_batch(self._ctx, len(batch), arr))

    @staticmethod
    def make_state_update(x0_lon=None, x0_lat=None, x0_orientation=None, v_des=None, low_vel_mode=None, t_samp=None,
                          v_samp=None, d_samp=None, obstacles=None):
        """FxStateUpdate (+ the arrays it points into) for `update_state`: what changes between two plan steps of a
        planner that keeps its reference path, grid shape and cost function.  `obstacles` = a packed predictions dict
        (problem.pack_predictions) with the K and P of the upload.  Build it once, reuse it as often as wanted."""
        u = _abi.FxStateUpdate()
        keep = []

        def arr(a, typ=np.float64):
            a = np.ascontiguousarray(a, dtype=typ)
            keep.append(a)
            return a.ctypes.data

        if x0_lon is not None: u.x0_lon = arr(x0_lon)
        if x0_lat is not None: u.x0_lat = arr(x0_lat)
        u.x0_orientation = float("nan") if x0_orientation is None else float(x0_orientation)
        u.v_des = float("nan") if v_des is None else float(v_des)
        u.low_vel_mode = -1 if low_vel_mode is None else int(bool(low_vel_mode))
        # (the shapes travel with the pointers: fx_update_state refuses arrays that are not the upload's size)
        if t_samp is not None: u.t_samp, u.nT = arr(t_samp), len(keep[-1])
        if v_samp is not None: u.v_samp, u.nV = arr(v_samp), len(keep[-1])
        if d_samp is not None: u.d_samp, u.nD = arr(d_samp), len(keep[-1])
        if obstacles is not None and obstacles["K"] > 0:
            u.K, u.P = int(obstacles["K"]), int(obstacles["P"])
            if np.shape(obstacles["pos"]) != (u.K, u.P, 2) or np.size(obstacles["cov_inv"]) != 4 * u.K * u.P \
                    or np.size(obstacles["npred"]) != u.K:
                raise ValueError("packed predictions: array shapes do not match their K / P")
            u.obs_pos, u.obs_cov_inv = arr(obstacles["pos"]), arr(obstacles["cov_inv"])
            u.obs_npred = arr(obstacles["npred"], np.int32)
            if obstacles.get("hull") is not None and obstacles["hull"].size:  # also when no hull is left: the old ones must go
                u.obs_hull, u.obs_nhull = arr(obstacles["hull"]), arr(obstacles["nhull"], np.int32)
        u._keep = keep
        return u

    @staticmethod
    def _state_update_of(inp: PlanInputs, u=None):
        """FxStateUpdate straight from a PlanInputs (its arrays are contiguous float64 / int32 already: no conversions); the
        addresses are taken in C (`_fxhost.state_update`: ten `array.ctypes.data` look-ups cost 8 us of a plan step)"""
        if u is None:
            u = _abi.FxStateUpdate()
        o = inp.obstacles
        h = _fxhost()
        if h:
            try:
                k = o["K"] > 0
                hull = k and o["hull"].size > 0
                h[0].state_update(C.addressof(u), inp.x0_lon, inp.x0_lat, float(inp.x0_orientation), float(inp.v_des),
                                  int(bool(inp.low_vel_mode)), inp.t_samp, inp.v_samp, inp.d_samp,
                                  o["pos"] if k else None, o["cov_inv"] if k else None, o["npred"] if k else None,
                                  o["hull"] if hull else None, o["nhull"] if hull else None)
                u._keep = inp   # the arrays live as long as the inputs do
                return u
            except (TypeError, ValueError, BufferError):
                C.memset(C.addressof(u), 0, C.sizeof(u))   # something is not a plain contiguous array: the long way
        u.x0_lon, u.x0_lat = inp.x0_lon.ctypes.data, inp.x0_lat.ctypes.data
        u.x0_orientation, u.v_des, u.low_vel_mode = float(inp.x0_orientation), float(inp.v_des), int(bool(inp.low_vel_mode))
        u.t_samp, u.v_samp, u.d_samp = inp.t_samp.ctypes.data, inp.v_samp.ctypes.data, inp.d_samp.ctypes.data
        u.nT, u.nV, u.nD = len(inp.t_samp), len(inp.v_samp), len(inp.d_samp)
        if o["K"] > 0:
            u.K, u.P = int(o["K"]), int(o["P"])
            if o["pos"].shape != (u.K, u.P, 2) or o["cov_inv"].size != 4 * u.K * u.P or o["npred"].size != u.K:
                raise ValueError("packed predictions: array shapes do not match their K / P")
            u.obs_pos, u.obs_cov_inv, u.obs_npred = o["pos"].ctypes.data, o["cov_inv"].ctypes.data, o["npred"].ctypes.data
            if o["hull"].size:
                u.obs_hull, u.obs_nhull = o["hull"].ctypes.data, o["nhull"].ctypes.data
        u._keep = inp   # the arrays live as long as the inputs do
        return u

    def update_state(self, update, agent: int = 0):
        """Rewrite the step-dependent inputs of one uploaded agent in place (fx_update_state): one small host-to-device
        copy in front of the next evaluation instead of a full upload."""
        check(lib().fx_update_state(self._ctx, int(agent), C.byref(update)))

    def update_step_raw(self, update):
        """update_state(agent 0) + step_raw() in ONE call across the boundary"""
        res = getattr(self, "_res_buf", None)
        if res is None or len(res) != len(self._inputs):
            res = self._res_buf = (_abi.FxResult * len(self._inputs))()
        check(lib().fx_update_step(self._ctx, C.byref(update), res))
        return res

    def evaluate(self):
        check(lib().fx_evaluate(self._ctx))

    def finish(self):
        n = len(self._inputs)
        res = (_abi.FxResult * n)()
        check(lib().fx_finish_batch(self._ctx, res))
        out = [r.as_dict() for r in res]
        return out

    def step_raw(self):
        """evaluate() + finish() for the resident inputs in ONE call across the boundary; returns the FxResult array
        (fields as attributes: best_index, best_cost, n_feasible, ...) without building Python dicts."""
        n = len(self._inputs)
        res = getattr(self, "_res_buf", None)
        if res is None or len(res) != n:
            res = self._res_buf = (_abi.FxResult * n)()
        check(lib().fx_step(self._ctx, res))
        return res

    def plan_step(self, inputs: PlanInputs) -> dict:
        self.upload(inputs)
        self.evaluate()
        return self.finish()[0]

    def plan_step_packaged(self, inputs: PlanInputs, yaw_rate0: float = 0.0):
        """A planner's closed-loop plan step in ONE call across the boundary (fx_plan_and_package): when the resident upload
        has the structure of `inputs` (PlanInputs.structure_key) only the ego state, sampling values and predictions are
        rewritten in place, else everything is uploaded; evaluation; result; the winner's arrays, which the device gathers
        into pinned host memory behind the selection.  Returns (result dict, WinnerPackage or None)."""
        key = inputs.structure_key()
        upd = None
        h = _fxhost()
        if h and inputs.sampling_matrix is None and self._resident_key == key and len(self._inputs) == 1 and inputs.write_bundle:
            # the closed loop's usual step through the extension (fx_plan_batch_packaged with one agent): the inputs' arrays are
            # read in C, the result comes back as a dict -- no struct filling, no per-field conversion on this side
            self._inputs = [inputs]
            pkgs = (_abi.FxPackage * 1)()
            block = np.empty((1, _abi.FX_PKG_ROWS, inputs.n_samples))
            out = h[0].plan_batch(h[2], self._ctx.value, self._inputs, (float(yaw_rate0),), block, C.addressof(pkgs), True)
            if out.__class__ is int:
                check(out)
            return out[0], (WinnerPackage(pkgs[0], block[0], inputs) if pkgs[0].found else None)
        if inputs.sampling_matrix is None and self._resident_key == key and len(self._inputs) == 1:
            upd = getattr(self, "_upd", None)
            if upd is None:
                upd = self._upd = _abi.FxStateUpdate()   # consumed inside the call below: one struct per engine
            upd = self._state_update_of(inputs, upd)
            self._inputs = [inputs]
        else:
            self.upload(inputs)
            self._resident_key = key
        res = (_abi.FxResult * 1)()
        pkg = _abi.FxPackage()
        block = np.empty((_abi.FX_PKG_ROWS, inputs.n_samples))
        check(lib().fx_plan_and_package(self._ctx, C.byref(upd) if upd is not None else None, float(yaw_rate0), res,
                                        C.byref(pkg), block.ctypes.data))
        return res[0].as_dict(), (WinnerPackage(pkg, block, inputs) if pkg.found else None)

    # -- survivor exchange inside the library (fx_comm_*) --
    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: the 128-byte id every rank hands to comm_init (broadcast it with whatever the program has)"""
        buf = (C.c_uint8 * 128)()
        check(lib().fx_comm_unique_id(C.addressof(buf)))
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib().fx_comm_init(self._ctx, C.addressof(buf), int(rank), int(world)))
        self._comm_world = int(world)
        self._comm_agents = int(self.max_agents)   # (fx_comm_init's default; comm_set_agents changes it)

    def comm_check(self, world: int) -> bool:
        """Local preconditions of comm_init (RCCL present, capacity), nothing collective: the ranks agree on the answer
        BEFORE any of them calls comm_init (fx_comm_check)."""
        return lib().fx_comm_check(self._ctx, int(world)) == 0

    def set_timeout_ms(self, timeout_ms: int):
        """Bound of every host wait on device work (default 20 000 ms, the reference's TIMEOUT); FxTimeoutError beyond it."""
        check(lib().fx_set_timeout_ms(self._ctx, int(timeout_ms)))

    def comm_set_agents(self, n_agents: int):
        """The agent rows every rank contributes to an exchange (fx_comm_set_agents): the ranks agree on it beforehand."""
        check(lib().fx_comm_set_agents(self._ctx, int(n_agents)))
        self._comm_agents = int(n_agents)

    def set_exchange_mode(self, mode: int):
        """0: the library's all-gather lands in device memory and a publication kernel copies it out; 1: it lands straight in the
        pinned block and a stream-ordered write signals it (fx_set_exchange_mode)."""
        check(lib().fx_set_exchange_mode(self._ctx, int(mode)))
        self._exchange_mode = int(mode)

    def comm_info(self) -> dict:
        """rank, world, the rank count RCCL itself reports (ncclCommCount; -1 if unavailable), agent rows per rank"""
        v = (C.c_int32 * 4)()
        check(lib().fx_comm_info(self._ctx, v))
        return dict(rank=int(v[0]), world=int(v[1]), rccl_ranks=int(v[2]), agent_rows=int(v[3]))

    def step_exchange_topk_raw(self, k: int):
        """evaluate + per-agent top-k + ONE all-gather + finish in one call (fx_step_exchange_topk):
        (FxResult array, cost [W, A, k], index [W, A, k]) with A = the communicator's agent rows (comm_set_agents)"""
        n, A = len(self._inputs), self._comm_agents
        res = getattr(self, "_res_buf", None)
        if res is None or len(res) != n:
            res = self._res_buf = (_abi.FxResult * n)()
        x = getattr(self, "_xchg_topk", None)
        if x is None or x[0].shape != (self._comm_world, A, k):
            x = self._xchg_topk = (np.empty((self._comm_world, A, k)), np.empty((self._comm_world, A, k), np.int64))
        check(lib().fx_step_exchange_topk(self._ctx, int(k), res, x[0].ctypes.data, x[1].ctypes.data))
        return res, x[0], x[1]

    def comm_destroy(self):
        check(lib().fx_comm_destroy(self._ctx))
        self._comm_world = 0

    def step_exchange_raw(self):
        """evaluate + all-gather of every rank's winner(s) + finish in ONE call: (FxResult array, cost [W, A], index [W, A]) with
        A = the communicator's agent rows (comm_set_agents; rows a rank does not fill say cost inf / index -1)"""
        n, A = len(self._inputs), self._comm_agents
        res = getattr(self, "_res_buf", None)
        if res is None or len(res) != n:
            res = self._res_buf = (_abi.FxResult * n)()
        x = getattr(self, "_xchg_buf", None)
        if x is None or x[0].shape != (self._comm_world, A):
            x = self._xchg_buf = (np.empty((self._comm_world, A)), np.empty((self._comm_world, A), np.int64))
        check(lib().fx_step_exchange(self._ctx, res, x[0].ctypes.data, x[1].ctypes.data))
        return res, x[0], x[1]

    def set_package(self, enabled: bool):
        """Gather the winner's arrays into pinned host memory behind every evaluation that writes the bundle (fx_set_package);
        read them with `package(agent)` after finish()."""
        check(lib().fx_set_package(self._ctx, int(bool(enabled))))
        self.packaging = bool(enabled)

    def package(self, agent: int = 0, yaw_rate0: float = 0.0):
        pkg = _abi.FxPackage()
        inp = self._inputs[agent]
        block = np.empty((_abi.FX_PKG_ROWS, inp.n_samples))
        check(lib().fx_read_package(self._ctx, int(agent), float(yaw_rate0), C.byref(pkg), block.ctypes.data_as(C.POINTER(C.c_double))))
        return WinnerPackage(pkg, block, inp) if pkg.found else None

    def plan_batch(self, inputs: Sequence[PlanInputs]) -> List[dict]:
        """One batched launch over the agents.  When the resident upload has the same agents' structures
        (PlanInputs.structure_key) only their states, sampling values and predictions are rewritten in place
        (fx_update_state per agent, one staging copy in front of the evaluation); otherwise everything is uploaded."""
        inputs = list(inputs)
        keys = [inp.structure_key() if inp.sampling_matrix is None else None for inp in inputs]
        if self._resident_keys is not None and keys == self._resident_keys and None not in keys:
            for a, inp in enumerate(inputs):
                self.update_state(self._state_update_of(inp), a)
            self._inputs = inputs
        else:
            self.upload(inputs)
            self._resident_keys = keys
        self.evaluate()
        return self.finish()

    def plan_batch_packaged(self, inputs: Sequence[PlanInputs], yaw_rates: Sequence[float]):
        """plan_batch + every agent's winner package in ONE call across the boundary (fx_plan_batch_packaged through
        `_fxhost.plan_batch`: the inputs' arrays are read in C, the results come back as dicts): ([result dict],
        [WinnerPackage or None]).  The general path -- different structures, a sampling matrix, horizons of different
        lengths, no extension -- is plan_batch() followed by package() per agent."""
        inputs = list(inputs)
        n = len(inputs)
        keys = [inp.structure_key() if inp.sampling_matrix is None else None for inp in inputs]
        h = _fxhost()
        S = inputs[0].n_samples if n else 0
        fast = bool(h) and n > 0 and None not in keys and all(inp.n_samples == S and inp.write_bundle for inp in inputs)
        if not fast:
            res = self.plan_batch(inputs)
            return res, [self.package(a, yaw_rates[a]) if inputs[a].write_bundle and getattr(self, "packaging", False) else None
                         for a in range(n)]
        update = self._resident_keys is not None and keys == self._resident_keys
        if not update:
            self.upload(inputs)
            self._resident_keys = keys
        else:
            self._inputs = inputs
        pkgs = (_abi.FxPackage * n)()
        blocks = np.empty((n, _abi.FX_PKG_ROWS, S))
        out = h[0].plan_batch(h[2], self._ctx.value, inputs, yaw_rates, blocks, C.addressof(pkgs), update)
        if out.__class__ is int:
            check(out)
        return out, [WinnerPackage(pkgs[a], blocks[a], inputs[a]) if pkgs[a].found else None for a in range(n)]

    def plan_batch_begin(self, inputs: Sequence[PlanInputs]):
        """First half of plan_batch_packaged: the agents' states rewritten in place (or everything uploaded) and the evaluation
        launched -- returns without waiting (fx_plan_batch_begin).  The token goes to plan_batch_end.  None when the batch
        cannot take the packaged path (see plan_batch_packaged): the caller then uses that one."""
        inputs = list(inputs)
        n = len(inputs)
        h = _fxhost()
        if not h or n == 0 or any(inp.sampling_matrix is not None or not inp.write_bundle for inp in inputs):
            return None
        S = inputs[0].n_samples
        if any(inp.n_samples != S for inp in inputs):
            return None
        keys = [inp.structure_key() for inp in inputs]
        update = self._resident_keys is not None and keys == self._resident_keys
        if not update:
            self.upload(inputs)
            self._resident_keys = keys
        else:
            self._inputs = inputs
        rc = h[0].plan_batch_begin(h[3], self._ctx.value, inputs, update)
        if rc is not None:
            check(rc)
        return (inputs, S)

    def plan_batch_end(self, token, yaw_rates: Sequence[float]):
        """Second half: wait for the evaluation plan_batch_begin launched, ([result dict], [WinnerPackage or None])"""
        inputs, S = token
        n = len(inputs)
        h = _fxhost()
        pkgs = (_abi.FxPackage * n)()
        blocks = np.empty((n, _abi.FX_PKG_ROWS, S))
        out = h[0].plan_batch_end(h[4], self._ctx.value, n, yaw_rates, blocks, C.addressof(pkgs))
        if out.__class__ is int:
            check(out)
        return out, [WinnerPackage(pkgs[a], blocks[a], inputs[a]) if pkgs[a].found else None for a in range(n)]

    # -- read-back --
    def costs(self, agent: int = 0):
        n = self._inputs[agent].n_candidates
        cost = np.zeros(n)
        flags = np.zeros(n, np.uint32)
        check(lib().fx_read_costs_agent(self._ctx, agent, cost.ctypes.data_as(C.POINTER(C.c_double)),
                                        flags.ctypes.data_as(C.POINTER(C.c_uint32))))
        return cost, flags

    def boundary_steps(self, agent: int = 0) -> np.ndarray:
        """first step at which each candidate's footprint meets the road boundary, -1 if never"""
        out = np.zeros(self._inputs[agent].n_candidates, dtype=np.int32)
        check(lib().fx_read_boundary_steps_agent(self._ctx, agent, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def costmap(self, agent: int = 0) -> np.ndarray:
        """[C, n_cost] raw (unweighted) partial costs, columns in inputs.cost_names order."""
        inp = self._inputs[agent]
        raw = np.zeros((max(len(inp.cost_names), 1), inp.n_candidates))
        check(lib().fx_read_costmap_agent(self._ctx, agent, raw.ctypes.data_as(C.POINTER(C.c_double))))
        return np.ascontiguousarray(raw[:len(inp.cost_names)].T)

    def coeffs(self, index: int, agent: int = 0):
        lon, lat = np.zeros(6), np.zeros(6)
        tl = C.c_int32(0)
        check(lib().fx_read_coeffs_agent(self._ctx, agent, int(index), lon.ctypes.data_as(C.POINTER(C.c_double)),
                                         lat.ctypes.data_as(C.POINTER(C.c_double)), C.byref(tl)))
        tau = C.c_double(0.0)
        check(lib().fx_read_lat_tau_agent(self._ctx, agent, int(index), C.byref(tau)))
        return lon, lat, tl.value, tau.value

    def sample(self, index: int, agent: int = 0) -> np.ndarray:
        """[14, S] planes of one candidate (gathered from the SoA bundle)."""
        inp = self._inputs[agent]
        out = np.zeros((_abi.FX_NUM_PLANES, inp.n_samples))
        check(lib().fx_read_sample_agent(self._ctx, agent, int(index), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def candidate(self, index: int, agent: int = 0) -> dict:
        """Everything one trajectory object exposes, with one synchronisation: planes [14, S], lon / lat coefficients,
        the lateral polynomial's delta_tau, traj_len, raw partial costs (inputs.cost_names order), total cost, flag word."""
        inp = self._inputs[agent]
        planes = np.zeros((_abi.FX_NUM_PLANES, inp.n_samples))
        co = np.zeros(13)
        raw = np.zeros(max(len(inp.cost_names), 1))
        tl, cost, flags = C.c_int32(0), C.c_double(0.0), C.c_uint32(0)
        pd = C.POINTER(C.c_double)
        have_b, have_c = bool(inp.write_bundle), bool(inp.write_costmap) and len(inp.cost_names) > 0
        check(lib().fx_read_candidate_agent(self._ctx, agent, int(index), planes.ctypes.data_as(pd) if have_b else None,
                                            co.ctypes.data_as(pd) if have_b else None, C.byref(tl) if have_b else None,
                                            raw.ctypes.data_as(pd) if have_c else None, C.byref(cost), C.byref(flags)))
        return dict(planes=planes if have_b else None, lon=co[:6].copy(), lat=co[6:12].copy(), tau_lat=float(co[12]), traj_len=tl.value,
                    raw_costs=raw[:len(inp.cost_names)] if have_c else None, cost=cost.value, flags=flags.value)

    def plane(self, name_or_index, agent: int = 0) -> np.ndarray:
        """[S, C] one plane of every candidate."""
        inp = self._inputs[agent]
        p = _abi.PLANE_INDEX[name_or_index] if isinstance(name_or_index, str) else int(name_or_index)
        out = np.zeros((inp.n_samples, inp.n_candidates))
        check(lib().fx_read_plane_agent(self._ctx, agent, p, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def bundle(self, agent: int = 0) -> np.ndarray:
        """[C, 14, S] whole bundle (D2H of every plane; debugging / logging only)."""
        return np.ascontiguousarray(np.stack([self.plane(p, agent) for p in range(_abi.FX_NUM_PLANES)]).transpose(2, 0, 1))

    def topk(self, k: int):
        n = len(self._inputs)
        cost = np.zeros((n, k))
        idx = np.zeros((n, k), np.int64)
        check(lib().fx_read_topk_batch(self._ctx, int(k), cost.ctypes.data_as(C.POINTER(C.c_double)),
                                       idx.ctypes.data_as(C.POINTER(C.c_int64))))
        return cost, idx

    def topk_to_device(self, k: int, d_cost_ptr: int, d_index_ptr: int):
        check(lib().fx_topk_to_device(self._ctx, int(k), C.c_void_p(d_cost_ptr), C.c_void_p(d_index_ptr)))

    @property
    def device_bytes(self) -> int:
        return int(lib().fx_device_bytes(self._ctx))

    @property
    def last_kernel_ms(self) -> float:
        return float(lib().fx_last_kernel_ms(self._ctx))

    @property
    def last_eval_kernel_ms(self) -> float:
        return float(lib().fx_last_eval_kernel_ms(self._ctx))
