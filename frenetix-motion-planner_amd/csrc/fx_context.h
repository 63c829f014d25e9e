// fx_context.h -- what the translation units of the C-ABI share: the context, its helpers, the launchers' declarations.
//   fx_api.hip           context life cycle, settings, measurement hooks
//   fx_api_step.hip      upload, launch policy (fx_evaluate), results, state updates, winner package, batched plan calls
//   fx_api_exchange.hip  survivor exchange inside the library (RCCL)
//   fx_api_host.hip      host geometry of the callers either side of the path, read-back
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>

#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <tuple>
#include <vector>
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include "fx_device.h"

extern "C" hipError_t fx_launch_eval(const DevProblem *d_probs, int n_agents, int max_blocks, size_t lds_bytes, int G,
                                     bool bundle, bool obst, bool extra, int wpe, hipEvent_t ev_start, hipEvent_t ev_stop,
                                     FuseArgs fuse, hipStream_t stream);
extern "C" hipError_t fx_launch_eval_grid(const DevProblem *d_probs, int n_agents, int max_blocks, int block_size,
                                          size_t lds_bytes, int G, bool bundle, bool obst, int wpe, bool wsplit,
                                          hipEvent_t ev_start, hipEvent_t ev_stop, FuseArgs fuse, hipStream_t stream);
extern "C" hipError_t fx_launch_obstacle(const DevProblem *d_probs, int n_agents, int max_items, size_t lds_bytes, int CH,
                                         hipEvent_t ev_start, hipEvent_t ev_stop, hipStream_t stream, int wg_waves, int max_tiles);
extern "C" hipError_t fx_step_kernel_capacity(int CH, size_t lds_bytes, int *blocks_out);
extern "C" hipError_t fx_launch_step(const DevProblem *d_probs, int n_agents, int blocks, size_t lds_bytes, int CH, hipEvent_t ev_start,
                                     hipEvent_t ev_stop, FuseArgs fuse, StepArgs sa, hipStream_t stream);
extern "C" hipError_t fx_launch_select(const DevProblem *d_probs, int n_agents, int64_t max_candidates, unsigned long long *host_result,
                                       unsigned long long seq, double *dev_winner, double *host_pkg, int pkg_stride, int pkg_plane_rows,
                                       hipStream_t stream);
extern "C" hipError_t fx_launch_math_test(int n, const double *x, double *at, double *sn, double *cs, hipStream_t stream);
extern "C" hipError_t fx_launch_publish(const double *src, int n, double *host_dst, unsigned long long *host_seq,
                                        unsigned long long seq, hipStream_t stream);
extern "C" hipError_t fx_launch_stage(const void *src_mapped, void *dst, size_t bytes, hipStream_t stream);
extern "C" hipError_t fx_launch_package(const DevProblem *d_probs, int n_agents, const double *winner, double *host_pkg, int stride,
                                        int plane_rows, unsigned long long seq, hipStream_t stream);
extern "C" hipError_t fx_launch_topk(const DevProblem *d_probs, int n_agents, int64_t max_candidates, int k, double *scr_cost, long long *scr_idx,
                                     double *out_cost, long long *out_idx, hipStream_t stream);


extern thread_local char g_err[512];   // (defined in fx_api.hip)

inline int set_err(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return set_err(FX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                             __FILE__, __LINE__);                                           \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }


// bundles up to this size use write-through plane stores.  tools/store_sweep.py on MI355X: write-through is faster up to
// ~0.5 GB (41 vs 45 us at 175 MB) and equal beyond (740 vs 746 us at 3.5 GB), so there is no upper limit by default.
#define FX_STORE_WT_MAX_BYTES (~(size_t)0)
// state updates up to this many bytes are staged by a copy kernel reading the mapped pinned block, larger ones by the DMA engine
#define FX_STAGE_KERNEL_MAX ((size_t)1 << 20)
#define FX_STAGE_HOST_MAX ((size_t)4 << 20)     // host writes into the device arena (large BAR): ~50 GB/s of posted writes
#define FX_PUB_MAX 16384  // doubles: 8 ranks x 64 survivors x 2 x 16 agents

struct FxAgentSlot {
    int64_t C = 0, ld = 0, cand_off = 0;  // cand_off: offset of this agent in the per-candidate arrays
    int32_t S = 0, n_cost = 0, n_blocks = 0;
    uint32_t mode = 0;
    // where the step-dependent inputs of this agent sit in the pinned staging block (fx_update_state rewrites them in
    // place); (size_t)-1 = not present
    size_t off_t = (size_t)-1, off_v = (size_t)-1, off_d = (size_t)-1, off_ref = (size_t)-1;
    size_t off_pos = (size_t)-1, off_cov = (size_t)-1, off_npred = (size_t)-1, off_hull = (size_t)-1, off_nhull = (size_t)-1;
    size_t off_rec = (size_t)-1, off_pm = (size_t)-1, off_hm = (size_t)-1, off_hot = (size_t)-1;
    size_t dyn_end = 0;   // end of this agent's step-dependent inputs that the walk reads every step
    size_t raw_end = 0;   // end of the raw predictions behind them
    int32_t nT = 0, nV = 0, nD = 0, K = 0, P = 0, M = 0;
    bool have_hull = false, want_collision = false;
};

struct FxContext {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // timing ring: per timed step (start, evaluation end, step end) events; elapsed times are read lazily, so a
    // timed step never waits for its own events
    struct TimeSlot {
        hipEvent_t e0 = nullptr, e_eval = nullptr, e_end = nullptr, e_obs0 = nullptr, e_obs1 = nullptr;
        bool eval_launched = false, fused = false, fetched = false, obst_timed = false;
        float step_ms = 0.f, eval_ms = 0.f, obst_ms = 0.f;
    };
    static constexpr int kTimeRing = 256;
    TimeSlot ring[kTimeRing];
    long long n_timed = 0;       // timed steps so far (slot = (n_timed - 1) % kTimeRing is the latest)
    long long n_steps = 0;       // evaluations so far
    int timing_every = 1;        // time every n-th step
    // capacities
    int64_t max_cand = 0;
    int32_t max_steps = 0, max_knots = 0, max_obs = 0, max_pred = 0, max_agents = 1;
    // input arena
    size_t in_bytes = 0;
    char *h_in = nullptr;   // pinned + mapped
    char *h_in_dev = nullptr;  // device address of the same block (the staging kernel reads it)
    int stage_mode = 0;        // 0 auto: kernel copy up to FX_STAGE_KERNEL_MAX bytes, DMA above; 1 DMA; 2 kernel; 3 host writes into
                               // device memory -- OPT-IN, or fail (FX_STAGE=dma|kernel|bar)
    bool user_stream = false;  // fx_set_stream handed in a caller's stream: what else is queued on it is unknown
    bool bar_ok = false;       // FX_STAGE=bar and the input arena d_in is mapped into this process (large BAR), host stores reach it
                               // and a kernel that had the lines cached sees a second write (probed at fx_create)
    volatile uint32_t *hdp_flush = nullptr;   // the device's HDP flush register (hipDeviceAttributeHdpMemFlushCntl), written behind host stores
    int stage_path = 0;        // how the latest inputs reached the device: 1 DMA copy, 2 staging kernel, 3 host writes
    char *d_in = nullptr;
    // problems
    DevProblem *h_probs = nullptr;  // [max_agents], the front of the pinned staging block h_in ...
    DevProblem *d_probs = nullptr;  // ... and of its device twin d_in: problems and inputs travel in ONE copy
    size_t probs_bytes = 0;
    // outputs
    double *d_cost = nullptr;
    double *d_cost_tail = nullptr;     // [total_ld] cost terms behind the prediction term (obstacle stage as its own kernel)
    // obstacle kernel scratch (allocated on first use): partial sums [chunks][ld], collision ballots [chunks][tiles], tile tickets
    double *d_obs_part = nullptr;
    unsigned long long *d_obs_colm = nullptr;
    unsigned int *d_obs_ticket = nullptr;
    int32_t *d_obs_list = nullptr;     // [total_ld] the walk's list of costed candidates per agent (obstacle stage as its own kernel)
    size_t obs_part_cap = 0, obs_colm_cap = 0;
    uint32_t *d_flags = nullptr;
    double *d_costmap = nullptr;
    double *d_coeffs = nullptr;
    int32_t *d_trajlen = nullptr;
    int32_t *d_bstep = nullptr;       // [total_ld] first road-boundary step per candidate
    char *h_bound = nullptr, *d_bound = nullptr;  // road-boundary pieces / bins / items (grown on demand)
    size_t bound_cap = 0;
    double *d_planes = nullptr;
    size_t planes_bytes = 0;
    double *d_part_cost = nullptr;
    int64_t *d_part_idx = nullptr;
    unsigned long long *d_counters = nullptr;  // [max_agents][FX_CNT_COUNT]
    unsigned long long *h_counters = nullptr;  // pinned + mapped: [max_agents][FX_CNT_COUNT + 1], last word = sequence
    unsigned long long *h_counters_dev = nullptr;  // device address of the same block
    unsigned long long seq = 0;
    // survivor exchange inside the library (fx_comm_init): an RCCL communicator of this context's own, the gathered winners
    void *comm = nullptr;                  // ncclComm_t
    int comm_rank = 0, comm_world = 0;
    int comm_agents = 0;                   // agent rows EVERY rank contributes to an exchange (fx_comm_set_agents; default max_agents)
    int comm_rows_clean = 0;               // send-buffer rows [comm_rows_clean, comm_agents) hold "no survivor"
    int comm_k_clean = 0;                  // ... for this k (0: the winner buffer)
    bool comm_init_failed = false;         // an fx_comm_init on this context timed out: never retried
    int exchange_mode = 0;                 // fx_set_exchange_mode: 0 receive in device memory + publication kernel, 1 receive straight in the pinned block
    double *d_gather = nullptr;            // [world][max_agents][2] (grown to [world][max_agents][2 k] by the top-k exchange)
    size_t gather_cap = 0;                 // doubles
    double *d_xsend = nullptr;             // [max_agents][2][64]: a rank's survivors, [cost n k | index n k], the all-gather's send buffer
    int timeout_ms = 20000;                // bound of every host wait on device work (fx_set_timeout_ms)
    bool timed_out = false;                // a wait ran out: the stream may never drain, the context refuses further steps
    double *h_pub = nullptr, *h_pub_dev = nullptr;   // pinned + mapped [FX_PUB_MAX + 1]: published buffer, last word = sequence
    unsigned long long pub_seq = 0;
    int pub_n = 0;
    double *dev_winner = nullptr;          // caller-owned device buffer [n_agents][2] the selection kernel also fills
    bool in_flight = false;                // work enqueued whose completion the host has not observed yet
    bool tail_work = false;                // work queued behind the evaluation whose completion NO sequence word reports (top-k, publication,
                                           // exchange) or a caller's own stream (fx_set_stream): cleared only by a stream synchronise --
                                           // fx_finish_batch clears in_flight when the evaluation's words arrive, which says nothing about these
    double *d_topk_cost = nullptr;
    long long *d_topk_idx = nullptr;
    double *d_topk_scr_cost = nullptr;     // [max_agents][64 slices][64]
    long long *d_topk_scr_idx = nullptr;
    double *h_topk_cost = nullptr;
    // winner package (fx_set_package): pinned + mapped [max_agents][pkg_stride] doubles the package kernel fills behind the selection
    double *h_pkg = nullptr, *h_pkg_dev = nullptr;
    int pkg_stride = 0, pkg_plane_rows = 0;
    double *d_winner_own = nullptr;        // [max_agents][2]: the winner stays device-resident for the package kernel
    bool package_enabled = false, pkg_step = false;
    double *h_cand = nullptr;  // pinned staging of fx_read_candidate_agent: planes | coeffs | raw costs | cost | traj_len | flags
    size_t h_cand_doubles = 0;
    long long *h_topk_idx = nullptr;
    int64_t total_ld = 0;  // capacity of per-candidate arrays (elements)
    int64_t max_blocks_total = 0;
    // current step
    int n_agents = 0;
    std::vector<FxAgentSlot> slots;
    bool uploaded = false, evaluated = false;
    size_t in_used = 0;                    // bytes of the staging block the last upload filled
    size_t dirty_lo = (size_t)-1, dirty_hi = 0;  // staging range rewritten by fx_update_state, copied by the next evaluation
    bool probs_dirty = false;
    int max_blocks_step = 0, M_max_step = 0, S_max_step = 0, K_max_step = 0;
    int G_step = 1, wpe_step = 2;          // lanes per candidate / occupancy target of the current step
    int G_force = 0, wpe_force = 0;        // fx_set_tuning overrides (0 = automatic)
    int variant_force = 0;                 // 0 auto, 1 generic kernel, 2 grid kernel
    int block_force = 0;                   // grid-kernel workgroup size override (0 auto)
    int wsplit_force = 0;                  // 0 auto, 1 lane split, 2 wave split
    int obst_force = 0;                    // obstacle stage: 0 auto, 1 fused into the walk, 2 its own kernel (fx_set_obstacle_stage)
    int obst_CH = 0;                       // steps per work item of the obstacle kernel (0 auto)
    bool split_step = false;               // current step runs fx_obstacle_kernel behind the walk
    int split_CH = 3, obs_blocks_step = 0;
    int obs_wg_waves = 0, obs_tiles_step = 0;   // obstacle kernel with one workgroup per tile: waves per workgroup (0: one wave per (tile, chunk) item), tiles
    int obs_wg_step = 0;                        // waves per workgroup of the last obstacle-kernel launch (0: single-wave items)
    size_t obs_lds_step = 0;
    // the whole step in ONE launch (fx_step_kernel.h): walk | grid barrier | obstacle items | grid barrier | selection
    int step_kernel_force = 0;             // 0 auto, 1 off, 2 on where applicable (fx_set_step_kernel; FX_STEP_KERNEL=0/1 in the environment)
    int step_kernel_CH = 0;                // steps per obstacle item in that kernel (0 auto; 3, 5 or 8)
    bool step_kernel_ok = false;           // the upload's step qualifies
    bool step_kernel_step = false;         // the last evaluation ran it
    int step_blocks = 0, step_CH = 0;      // workgroups per agent / steps per item of that launch
    size_t step_lds = 0;
    int64_t last_live = -1;                // costed candidates of the previous step's agents (max): sizes the obstacle items
    unsigned long long *d_bar = nullptr;   // the two grid barriers' counter + release-flag blocks, monotonic
    unsigned long long bar_base = 0;       // their value before the next launch
    int store_force = 0;                   // 0 auto, 1 write-back, 2 write-through plane stores
    bool wsplit_step = false;
    int block_step = FX_BLOCK;
    bool use_grid = false;                 // current step runs fx_eval_grid_kernel
    size_t lds_step = 0;
    bool any_bundle = false, any_obst = false, any_extra = false;
    int timing = FX_TIMING_OFF;
    bool timed_step = false, eval_launched = false;
    bool fuse_enabled = true, fusable_step = false, fused_step = false;
    // fused tail (fx_tail.h): the step's last workgroup also counts the collisions in front of the winner / gathers the package
    bool fuse_any_size = false;       // fx_set_fused_selection(ctx, 2): no candidate bound on the in-kernel collision count
    bool count_step = false;          // some agent of the upload runs the collision stage inside the evaluation kernel
    bool wt_step = false;             // the upload's plane stores are write-through
    uint32_t tail_step = 0;           // FX_TAIL_* of the last evaluation
    size_t gen_rec_lds = 0;           // generic kernel, >= 4 lanes per candidate: bytes of the staged obstacle records + step masks
    int64_t dev_bytes = 0;
};

extern "C" int32_t fx_wait_word(const volatile unsigned long long *word, unsigned long long expected, int32_t timeout_ms);   // (fx_api.hip)



// wait for a word of this context's pinned blocks; a timeout poisons the context (its stream may never drain)
inline int wait_seq(FxContext *c, const volatile unsigned long long *word, unsigned long long expected) {
    const int rc = fx_wait_word(word, expected, c->timeout_ms);
    if (rc == FX_ERR_TIMEOUT) c->timed_out = true;
    return rc;
}

template <typename T>
inline int dev_alloc(FxContext *c, T **p, size_t n) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T)));
    c->dev_bytes += (int64_t)(std::max<size_t>(n, 1) * sizeof(T));
    return FX_OK;
}

// per-step obstacle masks: one 64-bit word per 64 obstacles, word-major ([word][step]) so that the first word is the whole
// table for K <= 64 (the only case the grid kernel's staged obstacle path handles)
inline int mask_words(int K) { return K > 0 ? (K + 63) / 64 : 1; }

inline size_t input_bytes_for(int64_t cand, int S, int M, int K, int Pn, bool matrix) {
    size_t b = 0;
    b += align_up(sizeof(double) * 5 * S, 256);
    b += align_up(sizeof(double) * 3 * 4096, 256);              // t/v/d ranges
    if (matrix) b += align_up(sizeof(double) * 13 * (size_t)cand, 256);
    b += align_up(sizeof(double) * FX_REF_FIELDS * (size_t)M, 256);
    b += align_up(sizeof(double) * 2 * (size_t)K * Pn, 256);
    b += align_up(sizeof(double) * 4 * (size_t)K * Pn, 256);
    b += align_up(sizeof(double) * 6 * (size_t)K * (Pn > 0 ? Pn : 1), 256);
    b += align_up(sizeof(double) * 12 * (size_t)K * S, 256) + 2 * align_up(sizeof(unsigned long long) * S * (size_t)mask_words(K), 256);
    b += align_up(sizeof(double) * FX_HOT_STRIDE * (size_t)K * S, 256);  // hot obstacle table
    b += 2 * align_up(sizeof(int32_t) * (size_t)K, 256);
    b += align_up(sizeof(double) * 2 * (size_t)K, 256);           // dto positions (<= K)
    return b + 4096;
}

struct Arena {
    char *h, *d;
    size_t off, cap;
    template <typename T>
    const T *put(const T *src, size_t n, bool *ok) {
        size_t bytes = align_up(n * sizeof(T), 256);
        if (off + bytes > cap) { *ok = false; return nullptr; }
        if (n && src) memcpy(h + off, src, n * sizeof(T));
        const T *dp = reinterpret_cast<const T *>(d + off);
        off += bytes;
        return dp;
    }
    template <typename T>
    T *host_slot(size_t n, const T **dev, bool *ok) {
        size_t bytes = align_up(n * sizeof(T), 256);
        if (off + bytes > cap) { *ok = false; return nullptr; }
        T *hp = reinterpret_cast<T *>(h + off);
        *dev = reinterpret_cast<const T *>(d + off);
        off += bytes;
        return hp;
    }
};

// elapsed times of one ring slot (waits for the slot's last event if it is still pending)
inline int fetch_slot(FxContext *c, FxContext::TimeSlot &t) {
    if (t.fetched) return FX_OK;
    hipEvent_t end = t.fused ? t.e_eval : t.e_end;
    HIP_TRY(hipEventSynchronize(end));
    HIP_TRY(hipEventElapsedTime(&t.step_ms, t.e0, end));
    if (t.eval_launched) HIP_TRY(hipEventElapsedTime(&t.eval_ms, t.e0, t.e_eval));
    else t.eval_ms = 0.f;
    t.obst_ms = 0.f;
    if (t.obst_timed) HIP_TRY(hipEventElapsedTime(&t.obst_ms, t.e_obs0, t.e_obs1));
    t.fetched = true;
    return FX_OK;
}

inline int ensure_planes(FxContext *c, size_t bytes) {
    if (bytes <= c->planes_bytes) return FX_OK;
    if (c->d_planes) {
        HIP_TRY(hipFree(c->d_planes));
        c->dev_bytes -= (int64_t)c->planes_bytes;
        c->d_planes = nullptr;
        c->planes_bytes = 0;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_planes), bytes));
    c->planes_bytes = bytes;
    c->dev_bytes += (int64_t)bytes;
    return FX_OK;
}

inline int validate(const FxProblem *p) {
    if (!p) return set_err(FX_ERR_INVALID_ARGUMENT, "problem is NULL");
    if (p->N < 1 || p->N + 1 > FX_MAX_SAMPLES) return set_err(FX_ERR_INVALID_ARGUMENT, "N=%d outside [1,%d]", p->N, FX_MAX_SAMPLES - 1);
    if (!(p->dt > 0)) return set_err(FX_ERR_INVALID_ARGUMENT, "dt must be > 0");
    if (p->M < 2 || !p->ref_pos || !p->ref_x || !p->ref_y || !p->ref_nx || !p->ref_ny || !p->ref_theta || !p->ref_curv ||
        !p->ref_curv_d)
        return set_err(FX_ERR_NOT_READY, "reference path not set (M=%d)", p->M);
    if (!p->tpow) return set_err(FX_ERR_INVALID_ARGUMENT, "tpow table missing");
    if (p->sampling_matrix) {
        if (p->n_rows < 0) return set_err(FX_ERR_INVALID_ARGUMENT, "n_rows < 0");
    } else {
        if (p->nT < 0 || p->nV < 0 || p->nD < 0 || (p->nT && !p->t_samp) || (p->nV && !p->v_samp) || (p->nD && !p->d_samp))
            return set_err(FX_ERR_INVALID_ARGUMENT, "sampling ranges missing");
        if (p->nT > 4096 || p->nV > 4096 || p->nD > 4096) return set_err(FX_ERR_CAPACITY, "sampling range longer than 4096");
    }
    if (p->n_cost < 0 || p->n_cost > FX_NUM_COSTS) return set_err(FX_ERR_INVALID_ARGUMENT, "n_cost=%d", p->n_cost);
    if (p->lon_mode != FX_LON_VELOCITY_KEEPING && p->lon_mode != FX_LON_STOP_POINT)
        return set_err(FX_ERR_INVALID_ARGUMENT, "lon_mode=%d", p->lon_mode);
    if (p->lon_mode == FX_LON_STOP_POINT && p->sampling_matrix)
        return set_err(FX_ERR_INVALID_ARGUMENT, "stop-point sampling takes ranges, not a C x 13 matrix");
    for (int n = 0; n < p->n_cost; n++) {
        if (p->cost_id[n] < 0 || p->cost_id[n] >= FX_NUM_COSTS) return set_err(FX_ERR_INVALID_ARGUMENT, "unknown cost id %d", p->cost_id[n]);
        if (n && p->cost_id[n] <= p->cost_id[n - 1]) return set_err(FX_ERR_INVALID_ARGUMENT, "cost ids must be strictly ascending");
    }
    if (p->K > FX_MAX_OBSTACLES) return set_err(FX_ERR_CAPACITY, "at most %d obstacles per agent (K=%d)", FX_MAX_OBSTACLES, p->K);
    if (p->K < 0 || p->P < 0 || (p->K > 0 && (p->P < 2 || !p->obs_pos || !p->obs_cov_inv || !p->obs_npred)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "obstacle arrays inconsistent (K=%d, P=%d)", p->K, p->P);
    if ((p->mode & FX_MODE_COLLISION) && p->K > 0 && (!p->obs_hull || !p->obs_nhull))
        return set_err(FX_ERR_INVALID_ARGUMENT, "collision stage requested without obstacle hulls");
    if (p->n_dto < 0 || (p->n_dto > 0 && !p->dto_pos)) return set_err(FX_ERR_INVALID_ARGUMENT, "dto_pos missing");
    if (p->n_lane < 0 || (p->n_lane > 0 && (!p->lane_bbox || !p->lane_poly_off || !p->lane_poly || !p->lane_ctr_off || !p->lane_ctr)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "lanelet arrays missing (n_lane=%d)", p->n_lane);
    for (int l = 0; l < p->n_lane; l++)
        if (p->lane_poly_off[l + 1] < p->lane_poly_off[l] || p->lane_ctr_off[l + 1] < p->lane_ctr_off[l] || p->lane_poly_off[0] != 0 ||
            p->lane_ctr_off[0] != 0)
            return set_err(FX_ERR_INVALID_ARGUMENT, "lanelet offsets not ascending from 0 at lanelet %d", l);
    if (p->n_bound < 0 || (p->n_bound > 0 && (!p->bound_piece || !p->bound_bin || !p->bound_item)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "road boundary arrays missing (n_bound=%d)", p->n_bound);
    if ((p->mode & FX_MODE_ROAD_BOUNDARY) && p->n_bound > 0) {
        if (p->bound_bin[0] != 0) return set_err(FX_ERR_INVALID_ARGUMENT, "bound_bin[0] must be 0");
        for (int k = 0; k < p->M; k++)
            if (p->bound_bin[k + 1] < p->bound_bin[k]) return set_err(FX_ERR_INVALID_ARGUMENT, "bound_bin not ascending at %d", k);
        const int32_t n_item = p->bound_bin[p->M];
        for (int32_t j = 0; j < n_item; j++)
            if (p->bound_item[j] < 0 || p->bound_item[j] >= p->n_bound)
                return set_err(FX_ERR_INVALID_ARGUMENT, "bound_item[%d]=%d out of range", j, p->bound_item[j]);
        if (!(p->bound_d_reach > 0.0)) return set_err(FX_ERR_INVALID_ARGUMENT, "bound_d_reach must be positive");
    }
    return FX_OK;
}


// Step-major obstacle tables of one agent: rec[S][K][12] (mu, inverse covariance of prediction i-1; hull i-2), the per-step
// masks, and the hot table hot[S][K][FX_HOT_STRIDE] in the form the walk consumes (fx_walk.h, ObsHot).  Returns the margin of
// the broad phase's expanded circle test.
inline double pack_obstacle_tables(int S, int K, int P, const double *obs_pos, const double *obs_cov_inv, const int32_t *obs_npred,
                            const double *obs_hull, const int32_t *obs_nhull, bool have_hull, double ox, double oy,
                            double *rec, unsigned long long *pm, unsigned long long *hm, double *hot) {
    double r2_max = 0.0;
    for (int i = 0; i < S * mask_words(K); i++) pm[i] = hm[i] = 0ULL;
    for (int i = 0; i < S; i++) {
        for (int k = 0; k < K; k++) {
            const size_t mw = (size_t)(k >> 6) * S + i;   // word-major masks
            const unsigned long long mbit = 1ULL << (k & 63);
            double *q = rec + ((size_t)i * K + k) * 12;
            for (int e = 0; e < 12; e++) q[e] = 0.0;
            double *h = hot + ((size_t)i * K + k) * FX_HOT_STRIDE;
            for (int e = 0; e < FX_HOT_STRIDE; e++) h[e] = 0.0;
            if (i >= 1 && i < obs_npred[k] && i - 1 < P) {
                const double *mu = obs_pos + ((size_t)k * P + (i - 1)) * 2;
                const double *iv = obs_cov_inv + ((size_t)k * P + (i - 1)) * 4;
                q[0] = mu[0]; q[1] = mu[1]; q[2] = iv[0]; q[3] = iv[1]; q[4] = iv[2]; q[5] = iv[3];
                pm[mw] |= mbit;
                // Cholesky factor of the symmetric part of the inverse covariance: A = L^T L, L = [[l11, l12], [0, l22]];
                // the quadratic form r0 e0 + r1 e1 of the reference only sees that symmetric part.  No factor (not
                // positive definite, not finite): the entry stays zero, the form evaluates to 0 and the kernel redoes the
                // step from `rec`.
                const double a = iv[0], b = 0.5 * (iv[1] + iv[2]), dd = iv[3];
                const double l11 = std::sqrt(a), l12 = b / l11, l22sq = dd - l12 * l12;
                if (a > 0.0 && l22sq > 0.0 && std::isfinite(l11) && std::isfinite(l12) && std::isfinite(l22sq)) {
                    const double l22 = std::sqrt(l22sq), mx = mu[0] - ox, my = mu[1] - oy;
                    h[0] = l11; h[1] = l12; h[2] = l11 * mx + l12 * my; h[3] = l22; h[4] = l22 * my;
                }
            }
            if (have_hull && i >= 2 && i - 2 < obs_nhull[k]) {
                const double *oh = obs_hull + ((size_t)k * (P - 1) + (i - 2)) * 6;
                for (int e = 0; e < 6; e++) q[6 + e] = oh[e];
                hm[mw] |= mbit;
                // broad phase: circle that holds the hull (radius h1 + h2, with slack) in expanded form
                const double hx = oh[0] - ox, hy = oh[1] - oy, hr = (oh[4] + oh[5]) * 1.000001;
                h[5] = -2.0 * hx; h[6] = -2.0 * hy; h[7] = -2.0 * hr; h[8] = hx * hx + hy * hy - hr * hr;
                r2_max = std::max(r2_max, hx * hx + hy * hy);
            }
        }
    }
    // centre-gap values up to this margin go to the exact axis test: covers the rounding of the expanded form for ego hulls
    // within ~1 km of the origin (the pairs it adds are decided exactly, so decisions do not move)
    return 1e-6 + 4e-15 * (r2_max + 1e6);
}

// origin of the hot table's coordinates: the reference knot at the ego's arc length (any point near the ego does; it only
// keeps the products of the transformed forms small against their differences).  knots = [M][FX_REF_FIELDS] AoS.
inline void hot_origin_of(const double *knots, int M, double s0, double *origin) {
    int lo = 0, hi = M;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (knots[(size_t)mid * FX_REF_FIELDS] > s0) hi = mid; else lo = mid + 1;
    }
    const int ko = std::min(std::max(lo - 1, 0), M - 1);
    origin[0] = knots[(size_t)ko * FX_REF_FIELDS + 4];
    origin[1] = knots[(size_t)ko * FX_REF_FIELDS + 5];
}


extern "C" hipError_t fx_launch_probe_read(const void *src, void *dst, int blocks, hipStream_t stream);
// Host writes into device memory (large BAR) -- OPT-IN (FX_STAGE=bar).  On boxes where the whole VRAM is mapped into the process a
// state update needs no staging launch: the host copies the rewritten range of its pinned block into the device arena itself --
// posted PCIe writes, ~50 GB/s and no round trip (tools/micro/bar_write.hip, bar_bw.hip; ~3 us of a host-fed step).  Round 5 ran it
// by default; it is opt-in now because two links of the chain are not a documented contract of HIP: (a) the writes pass through the
// host data path (HDP) of the GPU, which the driver flushes behind ITS OWN writes to VRAM -- the library now does the same: it reads
// the device's HDP flush register address (hipDeviceAttributeHdpMemFlushCntl), writes 1 behind the stores and reads it back; without
// that attribute the path is refused -- and (b) a kernel must not find a stale copy of a rewritten line in an XCD's L2: kernel start
// invalidates the L2s on gfx942 / gfx950, and the probe below checks exactly that (write, kernel reads every line, write again,
// kernel reads again), failing closed to the staging kernel.  The default path -- staging kernel reading the pinned block, DMA above
// 1 MiB -- is stream-ordered and needs neither.
// The probe cannot fault: the mapping is tested through a system call (read(2) into the address returns EFAULT where nothing is
// mapped), then patterns written by the host are read by a device-to-host copy AND by a kernel launch, twice with different
// contents (the second read finds the first pattern's lines in whatever cache kept them).
inline bool probe_host_writes(FxContext *c, int device, char *d_in, size_t bytes) {
    int large = 0;
    if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, device) != hipSuccess || !large || bytes < 64) return false;
    uint32_t *flush = nullptr;
    if (hipDeviceGetAttribute(reinterpret_cast<int *>(&flush), hipDeviceAttributeHdpMemFlushCntl, device) != hipSuccess || !flush) {
        (void)hipGetLastError();
        return false;
    }
    const int fd = open("/dev/zero", O_RDONLY);
    if (fd < 0) return false;
    char *first = d_in, *last = d_in + ((bytes - 64) & ~(size_t)63);
    const bool mapped = read(fd, first, 64) == 64 && read(fd, last, 64) == 64;
    close(fd);
    if (!mapped) return false;
    c->hdp_flush = flush;
    // a kernel reads what the host wrote: 64 workgroups (all XCDs) copy the probed line into a scratch buffer, which a plain copy
    // brings back; three rounds with different patterns over the SAME lines
    constexpr int kProbeBlocks = 64;
    char *scratch = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&scratch), 64 * kProbeBlocks) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool ok = true;
    unsigned long long pat[8], back[8];
    std::vector<unsigned long long> seen(8 * kProbeBlocks);
    for (int round = 0; round < 3 && ok; round++)
        for (char *at : {first, last}) {
            for (int i = 0; i < 8; i++)
                pat[i] = 0x9e3779b97f4a7c15ULL * (unsigned long long)(i + 1 + 8 * round) ^ (unsigned long long)(uintptr_t)at;
            memcpy(at, pat, sizeof(pat));
            __builtin_ia32_sfence();
            *c->hdp_flush = 1u; (void)*c->hdp_flush;
            ok = ok && fx_launch_probe_read(at, scratch, kProbeBlocks, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
                 hipMemcpy(seen.data(), scratch, 64 * kProbeBlocks, hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(back, at, sizeof(back), hipMemcpyDeviceToHost) == hipSuccess && !memcmp(pat, back, sizeof(pat));
            for (int b = 0; b < kProbeBlocks && ok; b++) ok = !memcmp(pat, seen.data() + 8 * b, sizeof(pat));
        }
    (void)hipFree(scratch);
    if (!ok) { (void)hipGetLastError(); c->hdp_flush = nullptr; }
    return ok;
}
// the rewritten range of the pinned block, copied by the host (bar_ok; the caller has made sure the context's stream is idle)
inline void host_stage(FxContext *c, size_t lo, size_t hi) {
    memcpy(c->d_in + lo, c->h_in + lo, hi - lo);
    __builtin_ia32_sfence();
    *c->hdp_flush = 1u; (void)*c->hdp_flush;   // flush the GPU's host data path behind the stores (what the driver does behind its own)
    c->stage_path = 3;
}
// host writes only while NOTHING of this context's stream is pending: not the evaluation, not a top-k or publication kernel queued
// behind it, not work a caller put on a stream handed in with fx_set_stream -- any of them may still read the arena
inline bool host_stage_allowed(FxContext *c, size_t bytes) {
    // (hipStreamQuery is no help here: it reports hipErrorNotReady for a stream whose last kernel ended milliseconds ago until somebody
    // synchronises -- measured on ROCm 7.2 -- so the library keeps its own account of what may still read the arena)
    return c->bar_ok && !c->in_flight && !c->tail_work && bytes <= FX_STAGE_HOST_MAX;
}

inline int64_t max_candidates_of(const FxContext *c) {
    int64_t m = 0;
    for (int a = 0; a < c->n_agents; a++) m = std::max(m, c->slots[a].C);
    return m;
}
inline int check_agent(FxContext *c, int a) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (!c->evaluated) return set_err(FX_ERR_NOT_READY, "no evaluated plan step");
    if (a < 0 || a >= c->n_agents) return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d out of range", a);
    return FX_OK;
}
