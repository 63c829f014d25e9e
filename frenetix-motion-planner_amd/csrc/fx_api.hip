// fx_api.hip -- host side of libfxplan.so: the C-ABI declared in include/fxplan.h.
//
// A context owns (a) one pinned host staging buffer + one device arena for all per-step inputs, so a plan
// step is ONE H2D copy, (b) the per-candidate outputs (cost, flags, cost map, SoA bundle), (c) a pinned
// read-back block for the counters/winner, so a plan step is ONE small D2H copy.  All work is enqueued on the
// context's HIP stream; nothing synchronises until fx_finish().
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

namespace {

thread_local char g_err[512] = "";

int set_err(int code, const char *fmt, ...) {
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

}  // namespace

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

// Time-bounded wait for a sequence word the device publishes into pinned host memory.  The reference bounds every hand-off
// between its processes with TIMEOUT = 20 s (cr_scenario_handler/simulation/simulation.py:637,655, agent_batch.py:98): a
// peer that never joins a collective, or a faulted kernel, must surface as an error, not as a hang.  Pure host code.
extern "C" int32_t fx_wait_word(const volatile unsigned long long *word, unsigned long long expected, int32_t timeout_ms) {
    if (!word) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_wait_word: NULL argument");
    for (int spin = 0; spin < 4096; spin++)
        if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == expected) return FX_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const auto limit = std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
    for (;;) {
        for (int spin = 0; spin < 1024; spin++)
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == expected) return FX_OK;
        if (std::chrono::steady_clock::now() - t0 >= limit)
            return set_err(FX_ERR_TIMEOUT, "no answer from the device within %d ms (a peer that never joined the collective, or a "
                           "faulted kernel)", timeout_ms);
    }
}

namespace {

// wait for a word of this context's pinned blocks; a timeout poisons the context (its stream may never drain)
int wait_seq(FxContext *c, const volatile unsigned long long *word, unsigned long long expected) {
    const int rc = fx_wait_word(word, expected, c->timeout_ms);
    if (rc == FX_ERR_TIMEOUT) c->timed_out = true;
    return rc;
}

template <typename T>
int dev_alloc(FxContext *c, T **p, size_t n) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T)));
    c->dev_bytes += (int64_t)(std::max<size_t>(n, 1) * sizeof(T));
    return FX_OK;
}

// per-step obstacle masks: one 64-bit word per 64 obstacles, word-major ([word][step]) so that the first word is the whole
// table for K <= 64 (the only case the grid kernel's staged obstacle path handles)
inline int mask_words(int K) { return K > 0 ? (K + 63) / 64 : 1; }

size_t input_bytes_for(int64_t cand, int S, int M, int K, int Pn, bool matrix) {
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
int fetch_slot(FxContext *c, FxContext::TimeSlot &t) {
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

int ensure_planes(FxContext *c, size_t bytes) {
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

int validate(const FxProblem *p) {
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
double pack_obstacle_tables(int S, int K, int P, const double *obs_pos, const double *obs_cov_inv, const int32_t *obs_npred,
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
void hot_origin_of(const double *knots, int M, double s0, double *origin) {
    int lo = 0, hi = M;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (knots[(size_t)mid * FX_REF_FIELDS] > s0) hi = mid; else lo = mid + 1;
    }
    const int ko = std::min(std::max(lo - 1, 0), M - 1);
    origin[0] = knots[(size_t)ko * FX_REF_FIELDS + 4];
    origin[1] = knots[(size_t)ko * FX_REF_FIELDS + 5];
}

}  // namespace

extern "C" {

int32_t fx_abi_version(void) { return FX_ABI_VERSION; }
const char *fx_last_error(void) { return g_err; }

int32_t fx_device_count(int32_t *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return set_err(FX_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return FX_OK;
}

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
extern "C" hipError_t fx_launch_probe_read(const void *src, void *dst, int blocks, hipStream_t stream);
static bool probe_host_writes(FxContext *c, int device, char *d_in, size_t bytes) {
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
static void host_stage(FxContext *c, size_t lo, size_t hi) {
    memcpy(c->d_in + lo, c->h_in + lo, hi - lo);
    __builtin_ia32_sfence();
    *c->hdp_flush = 1u; (void)*c->hdp_flush;   // flush the GPU's host data path behind the stores (what the driver does behind its own)
    c->stage_path = 3;
}
// host writes only while NOTHING of this context's stream is pending: not the evaluation, not a top-k or publication kernel queued
// behind it, not work a caller put on a stream handed in with fx_set_stream -- any of them may still read the arena
static bool host_stage_allowed(FxContext *c, size_t bytes) {
    // (hipStreamQuery is no help here: it reports hipErrorNotReady for a stream whose last kernel ended milliseconds ago until somebody
    // synchronises -- measured on ROCm 7.2 -- so the library keeps its own account of what may still read the arena)
    return c->bar_ok && !c->in_flight && !c->tail_work && bytes <= FX_STAGE_HOST_MAX;
}

int32_t fx_create_batch(FxContext **out, int32_t device, int32_t max_agents, int64_t max_candidates_total,
                        int32_t max_steps, int32_t max_ref_knots, int32_t max_obstacles, int32_t max_pred_steps) {
    if (!out || max_candidates_total < 1 || max_steps < 1 || max_steps + 1 > FX_MAX_SAMPLES || max_ref_knots < 2 ||
        max_agents < 1 || max_obstacles < 0 || max_pred_steps < 0)
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_create: bad capacity arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return set_err(FX_ERR_NO_DEVICE, "no HIP device visible");
    }
    if (device < 0 || device >= ndev) return set_err(FX_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    FxContext *c = new (std::nothrow) FxContext();
    if (!c) return set_err(FX_ERR_HIP, "out of host memory");
    c->device = device;
    c->max_cand = max_candidates_total;
    c->max_steps = max_steps;
    c->max_knots = max_ref_knots;
    c->max_obs = max_obstacles;
    c->max_pred = std::max(max_pred_steps, 2);
    c->max_agents = max_agents;
    *out = c;
    {   // FX_STREAM_PRIORITY=low|high: experiments with two contexts sharing the device (tools/ns_two_streams.py)
        const char *pr = getenv("FX_STREAM_PRIORITY");
        int least = 0, greatest = 0;
        if (pr && *pr && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pr[0] == 'l' ? least : greatest));
        else
            HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    c->own_stream = true;
    for (auto &t : c->ring) {
        HIP_TRY(hipEventCreate(&t.e0));
        HIP_TRY(hipEventCreate(&t.e_eval));
        HIP_TRY(hipEventCreate(&t.e_end));
        HIP_TRY(hipEventCreate(&t.e_obs0));
        HIP_TRY(hipEventCreate(&t.e_obs1));
    }
    const int S = max_steps + 1;
    // every agent's leading dimension is rounded up to 64 candidates
    c->total_ld = (int64_t)align_up((size_t)max_candidates_total, 64) + 64 * (int64_t)max_agents;
    c->max_blocks_total = c->total_ld / 2 + max_agents + 1;  // 64-lane workgroups at G = 32: 2 candidates each
    c->in_bytes = (size_t)max_agents * input_bytes_for(0, S, max_ref_knots, max_obstacles, c->max_pred, false) +
                  align_up(sizeof(double) * 13 * (size_t)max_candidates_total, 256) + 4096;
    c->probs_bytes = align_up(sizeof(DevProblem) * (size_t)max_agents, 256);
    c->in_bytes += c->probs_bytes;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_in), c->in_bytes, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_in_dev), c->h_in, 0));
    if (const char *sm = getenv("FX_STAGE")) c->stage_mode = !strcmp(sm, "dma") ? 1 : (!strcmp(sm, "kernel") ? 2 : (!strcmp(sm, "bar") ? 3 : 0));
    if (const char *of = getenv("FX_OBST_STAGE")) c->obst_force = std::max(0, std::min(2, atoi(of)));   // experiments: fx_set_obstacle_stage's first argument
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_in), c->in_bytes));
    c->dev_bytes += (int64_t)c->in_bytes;
    if (c->stage_mode == 3) {   // opt-in (see probe_host_writes)
        c->bar_ok = probe_host_writes(c, device, c->d_in, c->in_bytes);
        if (!c->bar_ok)
            return set_err(FX_ERR_HIP, "FX_STAGE=bar: the device memory of GPU %d is not host-writable from this process, has no HDP flush "
                           "register, or a kernel did not see a rewritten line", device);
    }
    c->h_probs = reinterpret_cast<DevProblem *>(c->h_in);
    c->d_probs = reinterpret_cast<DevProblem *>(c->d_in);
    int rc;
    if ((rc = dev_alloc(c, &c->d_cost, c->total_ld))) return rc;
    if ((rc = dev_alloc(c, &c->d_cost_tail, c->total_ld))) return rc;
    if ((rc = dev_alloc(c, &c->d_flags, c->total_ld))) return rc;
    if ((rc = dev_alloc(c, &c->d_costmap, (size_t)FX_NUM_COSTS * c->total_ld))) return rc;
    if ((rc = dev_alloc(c, &c->d_coeffs, (size_t)FX_COEFF_ROWS * c->total_ld))) return rc;
    if ((rc = dev_alloc(c, &c->d_trajlen, c->total_ld))) return rc;
    if ((rc = dev_alloc(c, &c->d_bstep, c->total_ld))) return rc;
    if ((rc = dev_alloc(c, &c->d_part_cost, c->max_blocks_total))) return rc;
    if ((rc = dev_alloc(c, &c->d_part_idx, c->max_blocks_total))) return rc;
    if ((rc = dev_alloc(c, &c->d_counters, (size_t)max_agents * FX_CNT_COUNT))) return rc;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_counters), sizeof(unsigned long long) * max_agents * (FX_CNT_COUNT + 1),
                          hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_counters, 0, sizeof(unsigned long long) * max_agents * (FX_CNT_COUNT + 1));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_counters_dev), c->h_counters, 0));
    // ON THE CONTEXT'S STREAM: hipMemset on device memory runs on the null stream and may return before it has executed, and
    // a non-blocking stream does not wait for the null stream -- on a busy GPU (a second process) the zeroing landed behind the
    // first step's counter atomics and the step published zeros (tests/test_soak_parity.py, once in a few hundred contexts)
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(unsigned long long) * max_agents * FX_CNT_COUNT, c->stream));
    if ((rc = dev_alloc(c, &c->d_bar, 2 * 16 * 65))) return rc;   // (two barrier blocks: fx_step_kernel.h, FX_BAR_WORDS)
    HIP_TRY(hipMemsetAsync(c->d_bar, 0, sizeof(unsigned long long) * 2 * 16 * 65, c->stream));
    if (const char *e = getenv("FX_STEP_KERNEL")) c->step_kernel_force = atoi(e) ? 2 : 1;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_pub), sizeof(double) * (FX_PUB_MAX + 1), hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_pub, 0, sizeof(double) * (FX_PUB_MAX + 1));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_pub_dev), c->h_pub, 0));
    if ((rc = dev_alloc(c, &c->d_topk_cost, (size_t)max_agents * 64))) return rc;
    if ((rc = dev_alloc(c, &c->d_topk_idx, (size_t)max_agents * 64))) return rc;
    if ((rc = dev_alloc(c, &c->d_topk_scr_cost, (size_t)max_agents * 64 * 64))) return rc;
    if ((rc = dev_alloc(c, &c->d_topk_scr_idx, (size_t)max_agents * 64 * 64))) return rc;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_topk_cost), sizeof(double) * max_agents * 64, hipHostMallocDefault));
    c->pkg_plane_rows = FX_NUM_PLANES * (max_steps + 1);
    c->pkg_stride = (c->pkg_plane_rows + FX_PKG_TAIL + 1 + 7) & ~7;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_pkg), sizeof(double) * (size_t)max_agents * c->pkg_stride,
                          hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_pkg, 0, sizeof(double) * (size_t)max_agents * c->pkg_stride);
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&c->h_pkg_dev), c->h_pkg, 0));
    if ((rc = dev_alloc(c, &c->d_winner_own, (size_t)max_agents * 2))) return rc;
    c->h_cand_doubles = (size_t)FX_NUM_PLANES * (max_steps + 1) + FX_COEFF_ROWS + FX_NUM_COSTS + 4;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_cand), sizeof(double) * c->h_cand_doubles, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_topk_idx), sizeof(long long) * max_agents * 64, hipHostMallocDefault));
    c->slots.resize(max_agents);
    return FX_OK;
}

int32_t fx_create(FxContext **out, int32_t device, int64_t max_candidates, int32_t max_steps, int32_t max_ref_knots,
                  int32_t max_obstacles, int32_t max_pred_steps) {
    return fx_create_batch(out, device, 1, max_candidates, max_steps, max_ref_knots, max_obstacles, max_pred_steps);
}

int32_t fx_destroy(FxContext *c) {
    if (!c) return FX_OK;
    // a context whose wait timed out may hold work that never finishes (a collective a peer never joined): freeing its buffers
    // or its communicator would wait for that work.  It is abandoned as it is; the process is expected to end.
    if (c->timed_out) return FX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *dev[] = {c->d_in, c->d_cost, c->d_cost_tail, c->d_flags, c->d_costmap, c->d_coeffs, c->d_trajlen, c->d_planes,
                   c->d_part_cost, c->d_part_idx, c->d_counters, c->d_topk_cost, c->d_topk_idx, c->d_topk_scr_cost,
                   c->d_topk_scr_idx};
    for (void *p : dev) if (p) (void)hipFree(p);
    if (c->d_bstep) (void)hipFree(c->d_bstep);
    if (c->d_obs_part) (void)hipFree(c->d_obs_part);
    if (c->d_obs_colm) (void)hipFree(c->d_obs_colm);
    if (c->d_obs_ticket) (void)hipFree(c->d_obs_ticket);
    if (c->d_obs_list) (void)hipFree(c->d_obs_list);
    if (c->d_bar) (void)hipFree(c->d_bar);
    if (c->d_bound) (void)hipFree(c->d_bound);
    if (c->h_bound) (void)hipHostFree(c->h_bound);
    if (c->comm) (void)fx_comm_destroy(c);
    if (c->d_winner_own) (void)hipFree(c->d_winner_own);
    if (c->d_xsend) (void)hipFree(c->d_xsend);
    void *host[] = {c->h_in, c->h_counters, c->h_topk_cost, c->h_topk_idx, c->h_pub, c->h_cand, c->h_pkg};
    for (void *p : host) if (p) (void)hipHostFree(p);
    for (auto &t : c->ring) {
        if (t.e0) (void)hipEventDestroy(t.e0);
        if (t.e_eval) (void)hipEventDestroy(t.e_eval);
        if (t.e_end) (void)hipEventDestroy(t.e_end);
        if (t.e_obs0) (void)hipEventDestroy(t.e_obs0);
        if (t.e_obs1) (void)hipEventDestroy(t.e_obs1);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FX_OK;
}

// Publish n doubles of a device buffer (e.g. the all-gathered survivors) to the host through pinned memory;
// enqueued on the context stream.  fx_wait_published copies them out once they have arrived.
int32_t fx_publish(FxContext *c, const void *d_src, int32_t n) {
    if (!c || !d_src) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_publish: NULL argument");
    if (n < 1 || n > FX_PUB_MAX) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_publish: n=%d outside [1,%d]", n, FX_PUB_MAX);
    HIP_TRY(hipSetDevice(c->device));
    c->pub_seq++;
    c->pub_n = n;
    HIP_TRY(fx_launch_publish(reinterpret_cast<const double *>(d_src), n, c->h_pub_dev,
                              reinterpret_cast<unsigned long long *>(c->h_pub_dev + FX_PUB_MAX), c->pub_seq, c->stream));
    c->in_flight = true; c->tail_work = true;
    return FX_OK;
}

int32_t fx_wait_published(FxContext *c, double *out) {
    if (!c || !out) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_wait_published: NULL argument");
    if (c->pub_n < 1) return set_err(FX_ERR_NOT_READY, "nothing published");
    int rc = wait_seq(c, reinterpret_cast<const unsigned long long *>(c->h_pub + FX_PUB_MAX), c->pub_seq);
    if (rc) return rc;
    memcpy(out, c->h_pub, sizeof(double) * c->pub_n);
    return FX_OK;
}

static int64_t max_candidates_of(const FxContext *c) {
    int64_t m = 0;
    for (int a = 0; a < c->n_agents; a++) m = std::max(m, c->slots[a].C);
    return m;
}

int32_t fx_set_timeout_ms(FxContext *c, int32_t timeout_ms) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (timeout_ms < 1) return set_err(FX_ERR_INVALID_ARGUMENT, "timeout must be at least 1 ms");
    c->timeout_ms = timeout_ms;
    return FX_OK;
}

int32_t fx_set_winner_buffer(FxContext *c, void *d_winner) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    c->dev_winner = reinterpret_cast<double *>(d_winner);
    return FX_OK;
}

int32_t fx_set_store_mode(FxContext *c, int32_t store_mode) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (store_mode < 0 || store_mode > 2) return set_err(FX_ERR_INVALID_ARGUMENT, "store mode must be 0 (auto), 1 (write-back) or 2 (write-through)");
    c->store_force = store_mode;
    return FX_OK;
}

int32_t fx_set_part_mapping(FxContext *c, int32_t mapping) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (mapping < 0 || mapping > 2) return set_err(FX_ERR_INVALID_ARGUMENT, "mapping must be 0 (auto), 1 (lane split) or 2 (wave split)");
    c->wsplit_force = mapping;
    return FX_OK;
}

int32_t fx_set_block_size(FxContext *c, int32_t block_size) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (block_size != 0 && block_size != 64 && block_size != 128 && block_size != 256)
        return set_err(FX_ERR_INVALID_ARGUMENT, "block_size must be 0 (auto), 64, 128 or 256");
    c->block_force = block_size;
    return FX_OK;
}

int32_t fx_set_tuning(FxContext *c, int32_t lanes_per_candidate, int32_t waves_per_simd, int32_t kernel_variant) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (lanes_per_candidate != 0 && lanes_per_candidate != 1 && lanes_per_candidate != 2 && lanes_per_candidate != 4 &&
        lanes_per_candidate != 8 && lanes_per_candidate != 16 && lanes_per_candidate != 32)
        return set_err(FX_ERR_INVALID_ARGUMENT, "lanes_per_candidate must be 0 (auto), 1, 2, 4, 8, 16 or 32");
    if (waves_per_simd != 0 && (waves_per_simd < 2 || waves_per_simd > 4))
        return set_err(FX_ERR_INVALID_ARGUMENT, "waves_per_simd must be 0 (auto), 2, 3 or 4");
    if (kernel_variant < 0 || kernel_variant > 2) return set_err(FX_ERR_INVALID_ARGUMENT, "kernel_variant must be 0 (auto), 1 (generic) or 2 (grid)");
    c->G_force = lanes_per_candidate;
    c->wpe_force = waves_per_simd;
    c->variant_force = kernel_variant;
    return FX_OK;
}

int32_t fx_set_stream(FxContext *c, void *hip_stream) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (c->own_stream && c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
    c->user_stream = c->tail_work = true;   // (whatever else the caller queues on it may read the arena: no host writes, ever)
    return FX_OK;
}

// Stage n_agents problems.  Agent a's candidates occupy [cand_off, cand_off + ld) of every per-candidate array.
int32_t fx_upload_batch(FxContext *c, int32_t n_agents, const FxProblem *probs) {
    if (!c || !probs) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_upload: NULL argument");
    if (n_agents < 1 || n_agents > c->max_agents) return set_err(FX_ERR_CAPACITY, "n_agents=%d exceeds capacity %d", n_agents, c->max_agents);
    if (c->timed_out) return set_err(FX_ERR_TIMEOUT, "an earlier wait on this context timed out (its stream may never drain): destroy it");
    HIP_TRY(hipSetDevice(c->device));
    if (c->in_flight) {  // the pinned staging block is about to be rewritten: earlier copies must have landed
        { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
        c->in_flight = false;
    }
    c->uploaded = c->evaluated = false;
    Arena ar{c->h_in, c->d_in, c->probs_bytes, c->in_bytes};
    // road boundary: its own staging block, grown on demand (maps differ by orders of magnitude in size)
    size_t bound_need = 0;
    for (int a = 0; a < n_agents; a++) {
        const FxProblem *p = &probs[a];
        if ((p->mode & FX_MODE_ROAD_BOUNDARY) && p->n_bound > 0 && p->bound_bin && p->M > 0)
            bound_need += align_up(sizeof(double) * 4 * (size_t)p->n_bound, 256) + align_up(sizeof(int32_t) * ((size_t)p->M + 1), 256) +
                          align_up(sizeof(int32_t) * (size_t)std::max(p->bound_bin[p->M], 0), 256);
        if (p->n_lane > 0 && p->lane_poly_off && p->lane_ctr_off)   // the lanelets of the lane_center_offset cost live in the same block
            bound_need += align_up(sizeof(double) * 4 * (size_t)p->n_lane, 256) + 2 * align_up(sizeof(int32_t) * ((size_t)p->n_lane + 1), 256) +
                          align_up(sizeof(double) * 2 * (size_t)std::max(p->lane_poly_off[p->n_lane], 0), 256) +
                          align_up(sizeof(double) * 2 * (size_t)std::max(p->lane_ctr_off[p->n_lane], 0), 256);
    }
    if (bound_need > c->bound_cap) {
        if (c->h_bound) (void)hipHostFree(c->h_bound);
        if (c->d_bound) { (void)hipFree(c->d_bound); c->dev_bytes -= (int64_t)c->bound_cap; }
        c->h_bound = c->d_bound = nullptr;
        c->bound_cap = 0;
        const size_t cap = std::max<size_t>(2 * bound_need, 64 * 1024);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->h_bound), cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_bound), cap));
        c->bound_cap = cap;
        c->dev_bytes += (int64_t)cap;
    }
    Arena br{c->h_bound, c->d_bound, 0, c->bound_cap};
    // lanes per candidate: split the horizon over G lanes while the step has too few candidates to give every
    // SIMD of the chip (256 CUs x 4) a few waves; windowed (Simpson) costs need the whole horizon in one lane
    {
        int64_t waves1 = 0;
        bool extra_any = false;
        for (int a = 0; a < n_agents; a++) {
            const FxProblem *p = &probs[a];
            const int64_t Cg = p->shard_count > 0 ? p->shard_count : (p->sampling_matrix ? p->n_rows : (int64_t)p->nT * p->nV * p->nD);
            waves1 += (Cg + 63) / 64;
            for (int n = 0; n < p->n_cost && p->cost_id; n++) {
                const int id = p->cost_id[n];
                extra_any |= id == FX_COST_ACCELERATION || id == FX_COST_JERK || id == FX_COST_ORIENTATION_OFFSET ||
                             id == FX_COST_PATH_LENGTH || id == FX_COST_DISTANCE_TO_OBSTACLES || id == FX_COST_LANE_CENTER_OFFSET;
            }
        }
        // measured on MI355X (tools/quick.py): one lane per candidate once the grid gives >= 3 waves per SIMD,
        // two lanes per candidate below that, four for tiny grids (a single wave's 31-step chain is pure latency)
        int G = 1;
        if (waves1 < 3072) G = 2;
        if (waves1 < 200) G = 4;
        // planner-sized grids are one dependent chain per lane on a mostly idle chip: spread the horizon until every lane walks
        // one or two steps (plus its carry-in step) -- tools/sweep_small.py, 5 obstacles, kernel time at 4 / 8 / 16 / 32 lanes:
        // 630 candidates 45 / 31 / 21 / 19 us; 1 260 x 51 samples 69 / 47 / 33 / 27; 3 060: 44 / 31 / 25 / 28; 4 200: 46 / 34 / 27 / 45
        if (waves1 < 100) G = 16;
        if (waves1 < 32) G = 32;
        if (c->G_force) G = c->G_force;
        if (extra_any) G = 1;
        c->G_step = G;
        // large grids: 4 waves per SIMD (128 VGPRs, a few spills) beats 2 at full VGPR budget; small grids are
        // latency-bound with 1-2 waves per SIMD anyway and run faster unspilled
        // with the obstacle stage the walk needs ~220 VGPRs: three waves per SIMD (168 VGPRs, few spills) is the best
        // trade at scale, four spill inside the obstacle loop (tools/obst_sweep.py)
        bool obst_any = false;
        for (int a = 0; a < n_agents; a++) obst_any |= probs[a].K > 0 || ((probs[a].mode & FX_MODE_ROAD_BOUNDARY) && probs[a].n_bound > 0);
        // Obstacle stage as its own (candidate x step)-parallel kernel behind the walk (fx_obstacle_kernel.h): for grids whose
        // walk leaves most of the chip's issue slots idle (two lanes per candidate: 200 ... 3 072 waves) the K x S visits of a
        // candidate run at the walk's one or two waves per SIMD when fused; on their own they fill every SIMD.  Needs the
        // materialised planes (x, y, theta are read back), at most 64 obstacles and no road-boundary stage (that one stays in
        // the walk).  tools/c3_split.py: config 3 94.7 vs 98.6 - 105 us per step, config 5's agent with a bundle 240 vs 280 us,
        // 10 000 candidates equal, 3 060 and 1 M candidates slower.
        {
            const int CH = c->obst_CH ? c->obst_CH : 3;
            bool any_k = false, ok = !extra_any;
            size_t lds = 0;
            for (int a = 0; a < n_agents; a++) {
                const FxProblem *p = &probs[a];
                if ((p->mode & FX_MODE_ROAD_BOUNDARY) && p->n_bound > 0) ok = false;
                if (p->K <= 0) continue;
                any_k = true;
                if (p->K > 64 || !(p->mode & FX_MODE_WRITE_BUNDLE)) ok = false;
                lds = std::max(lds, sizeof(double) * 6 * (size_t)CH * (size_t)p->K);
            }
            if (c->obst_force == 2 && any_k && !ok)
                return set_err(FX_ERR_INVALID_ARGUMENT, "obstacle kernel forced but not applicable (needs FX_MODE_WRITE_BUNDLE, K <= 64, no road "
                               "boundary, no windowed cost term)");
            // (a forced work decomposition -- fx_set_tuning -- runs as asked: the automatic choice only follows the automatic G)
            c->split_step = any_k && ok && (c->obst_force == 2 || (c->obst_force == 0 && G == 2 && !c->G_force));
            c->split_CH = CH; c->obs_lds_step = lds;
            if (c->split_step) obst_any = false;   // the walk is tuned and built without the stage
        }
        // a materialised bundle makes the walk store-bound: more resident waves only add spills (1 M candidates, Mode B:
        // 651 us at 2 waves per SIMD, 697 us at 4 -- tools/sweep_1m_modeB.py)
        bool bundle_any = false;
        for (int a = 0; a < n_agents; a++) bundle_any |= (probs[a].mode & FX_MODE_WRITE_BUNDLE) != 0;
        // ... but with the obstacle stage in the walk as well (the north star as written) the kernel is bound by what a wave issues: one
        // FP64 instruction per ~16 cycles (tools/micro/clockrate.hip), so the third wave per SIMD pays (168 registers, no vector
        // spill): 1 062 -> 1 004 us same-box, tools/ns_wpe.py
        c->wpe_step = c->wpe_force ? c->wpe_force : (waves1 >= 3072 ? (bundle_any ? (obst_any ? 3 : 2) : (obst_any ? 3 : 4)) : 2);
        // grid kernel: sampling ranges, no windowed costs, and the longitudinal rows of a workgroup fit in LDS.
        // Workgroup size: the smallest of 64/128/256 lanes whose LDS footprint still lets a CU hold the target
        // number of waves (small workgroups balance small grids at wave granularity).
        bool grid_ok = !extra_any;
        for (int a = 0; a < n_agents && grid_ok; a++)
            if (probs[a].sampling_matrix || probs[a].nD < 1 || probs[a].K > 64) grid_ok = false;   // > 64 obstacles: multi-word masks, generic kernel
        size_t lds_need = 0;
        int block = FX_BLOCK;
        if (grid_ok) {
            size_t hot_block = 0;
            for (int a = 0; a < n_agents && !c->split_step; a++)   // (no staging blocks when the obstacle stage is its own kernel)
                hot_block = std::max(hot_block, align_up(sizeof(double) * FX_HOT_STRIDE * (size_t)std::max(probs[a].K, 0), 16));
            // whether a workgroup of blk lanes will run the wave split (the rule further down: G in {2, 4}, whole waves per part)
            auto ws_expected = [&](int blk) { return (G == 2 || G == 4) && (blk / G) % 64 == 0 && c->wsplit_force != 1; };
            auto lds_for = [&](int blk) {
                size_t need = 0;
                for (int a = 0; a < n_agents; a++) {
                    const FxProblem *p = &probs[a];
                    const size_t n_pairs = (size_t)(blk / G + p->nD - 2) / p->nD + 1;
                    const size_t S = (size_t)p->N + 1;
                    // time table + rows + wave-split exchange block (5 f64 + 5 u32 per slot) + tail: the knots' arc lengths
                    // during the prologue, one staging block of the step's hot obstacle table per wave during the walk
                    // (fx_eval_grid_kernel.h: the two share the bytes)
                    // lane split with the obstacle stage in the kernel: the record table + the two step masks behind the arc lengths
                    // (fx_eval_grid_kernel.h, LSTAGE; same rule there)
                    const bool lane_split = G > 1 && !(ws_expected(blk));
                    const size_t rec_bytes = sizeof(double) * (size_t)S_rec_doubles((int)S, std::max(p->K, 0));
                    const size_t rec_lds = (lane_split && obst_any && p->K > 0 && rec_bytes <= FX_REC_LDS_MAX) ? rec_bytes + 16 * S : 0;
                    need = std::max(need, sizeof(double) * FX_TP * S + 128 * n_pairs * S + (G > 1 ? (size_t)64 * blk : 0) +
                                              std::max(sizeof(double) * (((size_t)p->M + 1) & ~(size_t)1), (size_t)(blk / 64) * hot_block) + rec_lds);
                }
                return need;
            };
            const int want_waves = 4 * c->wpe_step;
            const size_t lds_static = 256;  // static LDS of the kernels (reductions)
            const size_t lds_cap = (160 * 1024) / 2 - 2 * lds_static;  // two workgroups per CU
            block = 0;
            int best_waves = 0;
            const int order_big[3] = {256, 128, 64}, order_small[3] = {128, 256, 64};
            for (int bi = 0; bi < 3; bi++) {
                // two parts on two waves (G = 2, wave split) with the obstacle stage: 128-lane workgroups -- one wave per part --
                // finish 3 - 8 % earlier than 256-lane ones (config 3: 88 - 95 vs 96 us); without obstacles they are slower
                // (config 2: 48.6 vs 42.1 us, select-only 38.4 vs 29.1) -- tools/sweep_tuning.py, tools/c3.py
                const int blk = (G >= 8 || (G == 2 && obst_any) ? order_small : order_big)[bi];
                if (c->wsplit_force == 2 && (G == 2 || G == 4) && (blk / G) % 64 != 0) continue;   // a forced wave split needs whole waves per part
                const size_t need = lds_for(blk);
                const int by_lds = (int)((160 * 1024) / (need + lds_static));
                const int waves = by_lds * (blk / 64);
                if (waves >= want_waves && need <= lds_cap) { block = blk; lds_need = need; break; }
                // nothing reaches the target (few lateral samples per pair -> many rows): keep the workgroup size that
                // holds the most waves per CU among those whose rows fit at all
                if (need <= lds_cap && waves > best_waves) { best_waves = waves; block = blk; lds_need = need; }
            }
            if (c->block_force) { block = c->block_force; lds_need = lds_for(block); }
            if (!block) { block = FX_BLOCK; lds_need = lds_for(block); }
            if (lds_need > lds_cap) grid_ok = false;   // at least two workgroups per CU
        }
        if (c->variant_force == 1) grid_ok = false;
        if (c->variant_force == 2 && !grid_ok) return set_err(FX_ERR_INVALID_ARGUMENT, "grid kernel forced but not applicable (G=%d block=%d rows+tables need %zu B of LDS per workgroup)", G, block, lds_need);
        c->use_grid = grid_ok;
        if (const char *pad = getenv("FX_LDS_PAD")) lds_need = std::max(lds_need, (size_t)atol(pad));  // experiments: occupancy cap through LDS
        c->lds_step = lds_need;
        c->block_step = grid_ok ? block : FX_BLOCK;
        // wave split needs whole waves per part (CPB % 64 == 0) and G in {2, 4}
        const bool ws_possible = grid_ok && (G == 2 || G == 4) && (c->block_step / G) % 64 == 0;
        c->wsplit_step = ws_possible && c->wsplit_force != 1;
        if (c->wsplit_force == 2 && !ws_possible && G > 1) return set_err(FX_ERR_INVALID_ARGUMENT, "wave split forced but not applicable");
    }
    // lane-split kernels with the obstacle stage inside: which agents' record tables ride in LDS (FX_MODE_INT_REC_LDS).  Grid
    // kernel: lds_for above has made room by the same rule; generic kernel (>= 4 lanes per candidate): behind the knots and the time
    // table where everything still fits a CU
    const bool rec_rule_grid = c->use_grid && c->G_step > 1 && !c->wsplit_step;
    bool rec_rule_gen = false;
    c->gen_rec_lds = 0;
    if (!c->use_grid && c->G_step >= 4) {
        size_t base_max = 0, need = 0;
        bool any_obst_in = false;
        for (int a = 0; a < n_agents; a++) {
            const FxProblem *p = &probs[a];
            const size_t S = (size_t)p->N + 1;
            base_max = std::max(base_max, sizeof(double) * ((size_t)p->M * (FX_REF_FIELDS + 1) + 2 + FX_TP * S));
            const size_t rb = sizeof(double) * (size_t)S_rec_doubles((int)S, std::max(p->K, 0));
            if (p->K > 0 && p->K <= 64 && rb <= FX_REC_LDS_MAX) need = std::max(need, rb + 16 * S);
            any_obst_in |= p->K > 0;
        }
        if (any_obst_in && need && base_max + need <= (size_t)160 * 1024 - 2048) { rec_rule_gen = true; c->gen_rec_lds = need; }
    }
    const int CPB = c->block_step / c->G_step;
    int64_t cand_off = 0, block_off = 0;
    size_t planes_need = 0, obs_part_need = 0, obs_colm_need = 0, obs_tick_need = 0;
    c->any_bundle = c->any_obst = c->any_extra = false;
    c->fusable_step = true;
    c->count_step = false;
    c->wt_step = false;
    // candidates per agent up to which the agent's last workgroup counts the collisions in front of the winner itself (it re-reads
    // the agent's flag words); larger steps keep fx_select_kernel's slices.  Measured (tools/probe_timeline.py, closed_loop_timing.py):
    // the tail costs ~6.5 us at 630 candidates and 8 - 10 us at 11 000, the selection kernel + gather behind a launch gap ~8.5 - 10 us
    // whatever the size -- plan() 70 -> 63 us at 630 candidates, 81 -> 84 us at 11 220: the bound sits between them
    static const int64_t tail_max_c = [] { const char *e = getenv("FX_TAIL_MAX_C"); return e ? (int64_t)atoll(e) : (int64_t)8192; }();
    bool all_deferred = n_agents > 0;
    c->max_blocks_step = 0;
    c->obs_blocks_step = 0;
    c->obs_tiles_step = 0;
    c->obs_wg_waves = 0;
    c->M_max_step = 0;
    c->K_max_step = 0;
    c->S_max_step = 0;
    for (int a = 0; a < n_agents; a++) {
        const FxProblem *p = &probs[a];
        int rc = validate(p);
        if (rc) return rc;
        const int S = p->N + 1;
        const int64_t C_global = p->sampling_matrix ? p->n_rows : (int64_t)p->nT * p->nV * p->nD;
        if (p->shard_count < 0 || p->shard_begin < 0 || (p->shard_count > 0 && p->shard_begin + p->shard_count > C_global))
            return set_err(FX_ERR_INVALID_ARGUMENT, "shard [%lld, +%lld) outside the grid of %lld candidates",
                           (long long)p->shard_begin, (long long)p->shard_count, (long long)C_global);
        const int64_t C = p->shard_count > 0 ? p->shard_count : C_global;
        const int64_t g_base = p->shard_count > 0 ? p->shard_begin : 0;
        if (p->N > c->max_steps) return set_err(FX_ERR_CAPACITY, "N=%d exceeds context capacity %d", p->N, c->max_steps);
        if (p->M > c->max_knots) return set_err(FX_ERR_CAPACITY, "M=%d reference knots exceed capacity %d", p->M, c->max_knots);
        if (p->K > c->max_obs || (p->K > 0 && p->P > c->max_pred))
            return set_err(FX_ERR_CAPACITY, "obstacles K=%d P=%d exceed capacity %d x %d", p->K, p->P, c->max_obs, c->max_pred);
        // only the generic kernel stages the whole knot records (64 B each) in LDS; the grid kernel keeps 8 B per knot and its
        // LDS need was checked when it was chosen above
        if (!c->use_grid && ((size_t)p->M * (FX_REF_FIELDS + 1) + 2 + FX_TP * (size_t)S) * sizeof(double) > 160 * 1024 - 1024)
            return set_err(FX_ERR_CAPACITY, "reference with %d knots does not fit the 160 KiB LDS of the generic kernel (sampling matrix / "
                           "windowed costs); resample the reference or use sampling ranges", p->M);
        const int64_t ld = (int64_t)align_up((size_t)std::max<int64_t>(C, 1), 64);
        if (cand_off + ld > c->total_ld) return set_err(FX_ERR_CAPACITY, "candidates exceed context capacity %lld", (long long)c->max_cand);
        DevProblem &d = c->h_probs[a];
        memset(&d, 0, sizeof(d));
        d.N = p->N; d.S = S; d.mode = p->mode; d.low_vel_mode = p->low_vel_mode; d.dt = p->dt;
        memcpy(d.x0_lon, p->x0_lon, sizeof(d.x0_lon));
        memcpy(d.x0_lat, p->x0_lat, sizeof(d.x0_lat));
        d.x0_orientation = p->x0_orientation; d.v_des = p->v_des; d.veh = p->veh;
        d.nT = p->nT; d.nV = p->nV; d.nD = p->nD; d.has_matrix = p->sampling_matrix != nullptr;
        d.lon_mode = p->lon_mode;
        d.C = C; d.g_base = g_base; d.ld = ld; d.M = p->M; d.K = p->K; d.P = p->P; d.n_cost = p->n_cost; d.n_dto = p->n_dto;
        bool extra = false;
        for (int n = 0; n < p->n_cost; n++) {
            d.cost_id[n] = p->cost_id[n];
            d.cost_w[n] = p->cost_w[n];
            const int id = p->cost_id[n];
            extra |= id == FX_COST_ACCELERATION || id == FX_COST_JERK || id == FX_COST_ORIENTATION_OFFSET ||
                     id == FX_COST_PATH_LENGTH || id == FX_COST_DISTANCE_TO_OBSTACLES || id == FX_COST_LANE_CENTER_OFFSET;
        }
        memcpy(d.simpson_corr, p->simpson_corr, sizeof(d.simpson_corr));
        bool ok = true;
        FxAgentSlot &sl = c->slots[a];
        sl = FxAgentSlot();
        sl.nT = p->nT; sl.nV = p->nV; sl.nD = p->nD; sl.K = p->K; sl.P = p->P; sl.M = p->M;
        sl.want_collision = (p->mode & FX_MODE_COLLISION) != 0;
        // what may change from step to step (fx_update_state) comes first, right behind the problems, so that an update is
        // one copy of the front of the block; the per-reference constants follow
        if (d.has_matrix) {
            d.matrix = ar.put(p->sampling_matrix, (size_t)13 * C_global, &ok);
        } else {
            sl.off_t = ar.off; d.t_samp = ar.put(p->t_samp, p->nT, &ok);
            sl.off_v = ar.off; d.v_samp = ar.put(p->v_samp, p->nV, &ok);
            sl.off_d = ar.off; d.d_samp = ar.put(p->d_samp, p->nD, &ok);
        }
        double *rec = nullptr, *hot = nullptr;
        unsigned long long *pm = nullptr, *hm = nullptr;
        const bool have_hull = p->K > 0 && p->obs_hull && p->obs_nhull;
        if (p->K > 0) {
            sl.have_hull = have_hull;
            if (!have_hull) d.mode &= ~FX_MODE_COLLISION;
            // step-major packed records + per-step obstacle masks + hot table (filled below, once the knots are staged): what the
            // walk reads every step comes first, so that a state update stages this range only
            const double *dev = nullptr, *dhot = nullptr;
            const unsigned long long *dpm = nullptr, *dhm = nullptr;
            sl.off_rec = ar.off; rec = ar.host_slot<double>((size_t)S * p->K * 12, &dev, &ok);
            sl.off_pm = ar.off; pm = ar.host_slot<unsigned long long>((size_t)S * mask_words(p->K), &dpm, &ok);
            sl.off_hm = ar.off; hm = ar.host_slot<unsigned long long>((size_t)S * mask_words(p->K), &dhm, &ok);
            sl.off_hot = ar.off; hot = ar.host_slot<double>((size_t)S * p->K * FX_HOT_STRIDE, &dhot, &ok);
            d.obs_rec = dev; d.obs_pmask = dpm; d.obs_hmask = dhm; d.obs_hot = dhot;
            sl.dyn_end = ar.off;
            // the raw predictions: read on the device only by the generic kernel's windowed costs, kept for re-packing
            sl.off_pos = ar.off; d.obs_pos = ar.put(p->obs_pos, (size_t)2 * p->K * p->P, &ok);
            sl.off_cov = ar.off; d.obs_cov_inv = ar.put(p->obs_cov_inv, (size_t)4 * p->K * p->P, &ok);
            sl.off_npred = ar.off; d.obs_npred = ar.put(p->obs_npred, p->K, &ok);
            sl.raw_end = ar.off;
            if (have_hull) {  // kept in the staging block for re-packing; the kernels read the hulls from `rec`
                sl.off_hull = ar.off; d.obs_hull = ar.put(p->obs_hull, (size_t)6 * p->K * (p->P - 1), &ok);
                sl.off_nhull = ar.off; d.obs_nhull = ar.put(p->obs_nhull, p->K, &ok);
            }
        } else {
            d.mode &= ~FX_MODE_COLLISION;
            sl.dyn_end = ar.off;
        }
        d.tpow = ar.put(p->tpow, (size_t)5 * S, &ok);
        {   // reference knots, AoS: pos, theta, curv, curv_d, x, y, nx, ny
            const double *dev = nullptr;
            sl.off_ref = ar.off;
            double *h = ar.host_slot<double>((size_t)p->M * FX_REF_FIELDS, &dev, &ok);
            if (h) {
                for (int k = 0; k < p->M; k++) {
                    double *q = h + (size_t)k * FX_REF_FIELDS;
                    q[0] = p->ref_pos[k]; q[1] = p->ref_theta[k]; q[2] = p->ref_curv[k]; q[3] = p->ref_curv_d[k];
                    q[4] = p->ref_x[k]; q[5] = p->ref_y[k]; q[6] = p->ref_nx[k]; q[7] = p->ref_ny[k];
                }
                if (rec && pm && hm && hot) {
                    hot_origin_of(h, p->M, p->x0_lon[0], d.hot_origin);
                    d.hot_gap_margin = pack_obstacle_tables(S, p->K, p->P, p->obs_pos, p->obs_cov_inv, p->obs_npred, p->obs_hull,
                                                            p->obs_nhull, have_hull, d.hot_origin[0], d.hot_origin[1], rec, pm, hm, hot);
                }
            }
            d.ref = dev;
        }
        if (p->n_dto > 0) d.dto_pos = ar.put(p->dto_pos, (size_t)2 * p->n_dto, &ok);
        if ((p->mode & FX_MODE_ROAD_BOUNDARY) && p->n_bound > 0) {
            d.n_bound = p->n_bound;
            d.bound_piece = br.put(p->bound_piece, (size_t)4 * p->n_bound, &ok);
            d.bound_bin = br.put(p->bound_bin, (size_t)p->M + 1, &ok);
            d.bound_item = br.put(p->bound_item, (size_t)p->bound_bin[p->M], &ok);
            d.bound_d_reach = p->bound_d_reach;
        } else {
            d.mode &= ~FX_MODE_ROAD_BOUNDARY;
        }
        if (p->n_lane > 0) {
            d.n_lane = p->n_lane;
            d.lane_bbox = br.put(p->lane_bbox, (size_t)4 * p->n_lane, &ok);
            d.lane_poly_off = br.put(p->lane_poly_off, (size_t)p->n_lane + 1, &ok);
            d.lane_poly = br.put(p->lane_poly, (size_t)2 * p->lane_poly_off[p->n_lane], &ok);
            d.lane_ctr_off = br.put(p->lane_ctr_off, (size_t)p->n_lane + 1, &ok);
            d.lane_ctr = br.put(p->lane_ctr, (size_t)2 * p->lane_ctr_off[p->n_lane], &ok);
        }
        if (!ok) return set_err(FX_ERR_CAPACITY, "input arena too small (%zu bytes)", c->in_bytes);
        d.cost = c->d_cost + cand_off;
        d.cost_tail = c->d_cost_tail + cand_off;
        d.flags = c->d_flags + cand_off;
        d.costmap = c->d_costmap + (size_t)FX_NUM_COSTS * cand_off;  // [n_cost][ld] inside this agent's slab
        d.coeffs = c->d_coeffs + (size_t)FX_COEFF_ROWS * cand_off;
        d.traj_len = c->d_trajlen + cand_off;
        d.bound_step = c->d_bstep + cand_off;
        const int walk_blocks = (int)((C + CPB - 1) / CPB);
        d.n_blocks = walk_blocks;
        const bool deferred = c->split_step && p->K > 0;
        {
            const size_t rb = sizeof(double) * (size_t)S_rec_doubles(S, std::max(p->K, 0));
            if (!deferred && p->K > 0 && p->K <= 64 && rb <= FX_REC_LDS_MAX && (rec_rule_grid || rec_rule_gen)) d.mode |= FX_MODE_INT_REC_LDS;
        }
        all_deferred = all_deferred && deferred;
        if (deferred) {   // the obstacle kernel writes this agent's arg-min partials: one per tile of 64 candidates
            d.mode |= FX_MODE_INT_DEFER_OBST;
            const int n_tiles = (int)((C + 63) / 64), NC = (S - 1 + c->split_CH - 1) / c->split_CH;
            const int NC_alloc = std::max(NC, (S - 1 + 2) / 3);   // (the one-launch step picks its own steps per item: 3, 5 or 8)
            d.n_blocks = n_tiles;
            c->obs_blocks_step = std::max(c->obs_blocks_step, n_tiles * NC);
            c->obs_tiles_step = std::max(c->obs_tiles_step, n_tiles);
            c->obs_wg_waves = std::max(c->obs_wg_waves, NC);
            d.obs_part = reinterpret_cast<double *>(obs_part_need);      // offsets for now, patched below
            d.obs_colm = reinterpret_cast<unsigned long long *>(obs_colm_need);
            d.obs_ticket = reinterpret_cast<unsigned int *>(obs_tick_need);
            obs_part_need += (size_t)NC_alloc * (size_t)ld;
            obs_colm_need += (size_t)NC_alloc * (size_t)n_tiles;
            obs_tick_need += (size_t)n_tiles;
        }
        if (block_off + d.n_blocks > c->max_blocks_total)
            return set_err(FX_ERR_CAPACITY, "agent %d: %lld workgroups exceed the partial-result capacity %lld", a,
                           (long long)(block_off + d.n_blocks), (long long)c->max_blocks_total);
        d.part_cost = c->d_part_cost + block_off;
        d.part_idx = c->d_part_idx + block_off;
        d.counters = c->d_counters + (size_t)a * FX_CNT_COUNT;
        d.pkg_out = c->h_pkg_dev + (size_t)a * c->pkg_stride;
        d.pkg_seq = reinterpret_cast<unsigned long long *>(d.pkg_out + c->pkg_stride - 1);
        d.pkg_plane_rows = c->pkg_plane_rows;
        if (d.mode & FX_MODE_WRITE_BUNDLE) {
            if ((uint64_t)ld * 8u >= (1ull << 32))  // the walk addresses a row with a 32-bit byte offset per lane
                return set_err(FX_ERR_CAPACITY, "agent %d: %lld candidates with a materialised bundle (rows are limited to 4 GiB)", a, (long long)C);
            d.planes = reinterpret_cast<double *>(planes_need);  // offset for now, patched below
            planes_need += sizeof(double) * FX_NUM_PLANES * (size_t)S * (size_t)ld;
            c->any_bundle = true;
        }
        c->any_obst |= (p->K > 0 && !deferred) || (d.mode & FX_MODE_ROAD_BOUNDARY);
        if (d.n_blocks == 0 || deferred) c->fusable_step = false;
        if (d.mode & FX_MODE_COLLISION) {
            c->count_step = true;
            if (C > tail_max_c && !c->fuse_any_size) c->fusable_step = false;
        }
        c->any_extra |= extra;
        c->max_blocks_step = std::max(c->max_blocks_step, walk_blocks);
        c->M_max_step = std::max(c->M_max_step, p->M);
        c->K_max_step = std::max(c->K_max_step, std::max(p->K, 0));
        c->S_max_step = std::max(c->S_max_step, S);
        sl.C = C; sl.ld = ld; sl.cand_off = cand_off; sl.S = S; sl.n_cost = p->n_cost; sl.n_blocks = d.n_blocks; sl.mode = d.mode;
        cand_off += ld;
        block_off += d.n_blocks;
    }
    if (planes_need) {
        int rc = ensure_planes(c, planes_need);
        if (rc) return rc;
        // plane stores: write-through while the step's whole bundle is small (FX_STORE_WT_MAX_BYTES, measured), else write-back
        const bool wt = c->store_force == 2 || (c->store_force == 0 && planes_need <= FX_STORE_WT_MAX_BYTES);
        for (int a = 0; a < n_agents; a++) {
            c->h_probs[a].mode &= ~FX_MODE_INT_STORE_WT;
            if (wt) c->h_probs[a].mode |= FX_MODE_INT_STORE_WT;
        }
        c->wt_step = wt;
        for (int a = 0; a < n_agents; a++)
            if (c->h_probs[a].mode & FX_MODE_WRITE_BUNDLE)
                c->h_probs[a].planes = reinterpret_cast<double *>(reinterpret_cast<char *>(c->d_planes) +
                                                                  reinterpret_cast<size_t>(c->h_probs[a].planes));
    }
    if (obs_part_need) {   // scratch of the obstacle kernel: grown on demand, tickets start (and are left) zeroed
        if (obs_part_need > c->obs_part_cap || obs_colm_need > c->obs_colm_cap) {
            { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
            if (c->d_obs_part) { (void)hipFree(c->d_obs_part); c->dev_bytes -= (int64_t)(sizeof(double) * c->obs_part_cap); }
            if (c->d_obs_colm) { (void)hipFree(c->d_obs_colm); c->dev_bytes -= (int64_t)(sizeof(unsigned long long) * c->obs_colm_cap); }
            c->d_obs_part = nullptr; c->d_obs_colm = nullptr;
            c->obs_part_cap = c->obs_colm_cap = 0;
            int rc;
            if ((rc = dev_alloc(c, &c->d_obs_part, obs_part_need))) return rc;
            if ((rc = dev_alloc(c, &c->d_obs_colm, obs_colm_need))) return rc;
            c->obs_part_cap = obs_part_need; c->obs_colm_cap = obs_colm_need;
        }
        if (!c->d_obs_ticket) {
            int rc;
            const size_t n_tick = (size_t)(c->total_ld / 64) + (size_t)c->max_agents;
            if ((rc = dev_alloc(c, &c->d_obs_ticket, n_tick))) return rc;
            HIP_TRY(hipMemsetAsync(c->d_obs_ticket, 0, sizeof(unsigned int) * n_tick, c->stream));   // (in order with the step's kernels)
        }
        if (!c->d_obs_list) {
            int rc;
            if ((rc = dev_alloc(c, &c->d_obs_list, (size_t)c->total_ld))) return rc;
        }
        for (int a = 0; a < n_agents; a++) {
            DevProblem &d = c->h_probs[a];
            if (!(d.mode & FX_MODE_INT_DEFER_OBST)) continue;
            d.obs_part = c->d_obs_part + reinterpret_cast<size_t>(d.obs_part);
            d.obs_colm = c->d_obs_colm + reinterpret_cast<size_t>(d.obs_colm);
            d.obs_ticket = c->d_obs_ticket + reinterpret_cast<size_t>(d.obs_ticket);
            d.obs_list = c->d_obs_list + (d.cost - c->d_cost);   // the agent's slab of the per-candidate arrays
        }
    }
    // the whole step in one launch (fx_step_kernel.h): the split step of the tuned two-lanes-per-candidate walk with a write-through
    // bundle, every agent's obstacle stage deferred; whether the device holds the launch is asked when it is sized (fx_evaluate)
    // (opt-in: measured slower than the three launches on config 3, fx_step_kernel.h -- `force` 2 or FX_STEP_KERNEL=1)
    c->step_kernel_ok = c->step_kernel_force == 2 && c->split_step && all_deferred && c->use_grid && c->G_step == 2 && c->wsplit_step &&
                        c->block_step == FX_BLOCK && c->wpe_step == 2 && c->any_bundle && !c->any_obst && !c->any_extra && c->wt_step &&
                        c->obs_blocks_step > 0 && c->K_max_step <= 64;
    c->last_live = -1;
    c->n_agents = n_agents;
    c->in_used = ar.off;
    c->dirty_lo = (size_t)-1; c->dirty_hi = 0; c->probs_dirty = false;
    // problems + inputs: small uploads through the staging kernel as well (the DMA engine's submission latency dominates below ~1 MiB)
    {
        const size_t up = (ar.off + 15) & ~(size_t)15;
        if (up <= c->in_bytes && host_stage_allowed(c, up))
            host_stage(c, 0, up);
        else if (c->stage_mode == 2 || (c->stage_mode != 1 && up <= FX_STAGE_KERNEL_MAX && up <= c->in_bytes)) {
            HIP_TRY(fx_launch_stage(c->h_in_dev, c->d_in, up, c->stream));
            c->stage_path = 2;
        } else {
            HIP_TRY(hipMemcpyAsync(c->d_in, c->h_in, ar.off, hipMemcpyHostToDevice, c->stream));
            c->stage_path = 1;
        }
    }
    if (br.off) HIP_TRY(hipMemcpyAsync(c->d_bound, c->h_bound, br.off, hipMemcpyHostToDevice, c->stream));
    c->uploaded = true;
    c->in_flight = true;
    return FX_OK;
}

int32_t fx_upload(FxContext *c, const FxProblem *prob) { return fx_upload_batch(c, 1, prob); }

int32_t fx_evaluate(FxContext *c) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (!c->uploaded) return set_err(FX_ERR_NOT_READY, "fx_evaluate before fx_upload");
    if (c->timed_out) return set_err(FX_ERR_TIMEOUT, "an earlier wait on this context timed out: destroy it (its stream may never drain)");
    HIP_TRY(hipSetDevice(c->device));
    if (c->probs_dirty || c->dirty_hi > c->dirty_lo) {
        // inputs rewritten by fx_update_state since the last evaluation: ONE copy of the front of the staging block (the
        // problems, then whatever changed behind them)
        const size_t lo = c->probs_dirty ? 0 : c->dirty_lo;
        const size_t hi = std::max(c->dirty_hi > c->dirty_lo ? c->dirty_hi : 0, c->probs_dirty ? sizeof(DevProblem) * (size_t)c->n_agents : 0);
        // (offsets inside the block are multiples of 256, so the 16-byte lanes of the staging kernel line up)
        const size_t lo16 = lo & ~(size_t)15, hi16 = (hi + 15) & ~(size_t)15;
        // host writes only while nothing of this context is in flight: fx_update_state drained the stream (or fx_finish saw the
        // previous step's last word) before the block was rewritten, so no kernel still reads the arena
        if (host_stage_allowed(c, hi16 - lo16))
            host_stage(c, lo16, hi16);
        else if (c->stage_mode == 2 || (c->stage_mode != 1 && hi16 - lo16 <= FX_STAGE_KERNEL_MAX)) {
            HIP_TRY(fx_launch_stage(c->h_in_dev + lo16, c->d_in + lo16, hi16 - lo16, c->stream));
            c->stage_path = 2;
        } else {
            HIP_TRY(hipMemcpyAsync(c->d_in + lo, c->h_in + lo, hi - lo, hipMemcpyHostToDevice, c->stream));
            c->stage_path = 1;
        }
        c->dirty_lo = (size_t)-1; c->dirty_hi = 0;
        c->probs_dirty = false;
    }
    // timing (every timing_every-th step): FX_TIMING_KERNEL attaches start/stop events to the evaluation kernel
    // itself (hipExtLaunchKernel), so its duration is the kernel's, not launch latency; FX_TIMING_STREAM brackets
    // with stream events instead (includes the dispatch gap before the kernel).  Events live in a ring and are
    // only read on request.
    const bool timed = c->timing != FX_TIMING_OFF && (c->n_steps % c->timing_every) == 0;
    c->n_steps++;
    c->eval_launched = c->max_blocks_step > 0;
    const bool attached = timed && c->timing == FX_TIMING_KERNEL && c->eval_launched;
    FxContext::TimeSlot *ts = nullptr;
    if (timed) {
        ts = &c->ring[c->n_timed % FxContext::kTimeRing];
        ts->fetched = false;
        ts->eval_launched = c->eval_launched;
    }
    hipEvent_t k0 = attached ? ts->e0 : nullptr, k1 = attached ? ts->e_eval : nullptr;
    if (timed && !attached) HIP_TRY(hipEventRecord(ts->e0, c->stream));
    // one launch when no agent needs the collision-ordered count of the selection kernel: the evaluation kernel's
    // last workgroup reduces and publishes (fx_eval_kernel.h, "fused selection")
    c->seq++;
    c->fused_step = c->fuse_enabled && c->fusable_step && c->eval_launched;
    c->pkg_step = c->package_enabled && c->any_bundle;
    double *winner = c->dev_winner ? c->dev_winner : (c->pkg_step ? c->d_winner_own : nullptr);
    // the agent's last workgroup ends the step (fx_tail.h): collision count where a collision stage ran in this kernel, winner
    // package where the bundle is stored write-through -- a planner-sized step with everything on is ONE launch
    // (the tail is compiled into the planner-sized decompositions only, FX_TAIL_IN_KERNEL: a step of another decomposition that
    // needs the collision count keeps the selection kernel, one that only needs the package keeps the package kernel)
    const bool tail_kernel = FX_TAIL_IN_KERNEL(c->G_step, c->any_extra);
    if (c->fused_step && c->count_step && !tail_kernel) c->fused_step = false;
    c->tail_step = 0;
    if (c->fused_step && tail_kernel) {
        if (c->count_step) c->tail_step |= FX_TAIL_COUNT;
        if (c->pkg_step && c->wt_step) c->tail_step |= FX_TAIL_PACKAGE;
    }
    const bool pkg_in_tail = (c->tail_step & FX_TAIL_PACKAGE) != 0;
    FuseArgs fuse{c->fused_step ? c->h_counters_dev : nullptr, c->seq, winner, (int32_t)((uint32_t)c->K_max_step | (c->tail_step << 16))};
    // ---- the whole step in ONE launch (fx_step_kernel.h) where the upload qualifies and the device holds the launch at once ----
    c->step_kernel_step = false;
    if (c->step_kernel_ok && !c->fused_step && c->eval_launched) {
        // sizing: (tile, chunk of CH steps) items over all waves of the launch, in as few rounds as the resident workgroups allow --
        // the list's length is the previous step's (a planner's consecutive steps differ little), two thirds of the grid at first
        int64_t c_max = 0;
        for (int a = 0; a < c->n_agents; a++) c_max = std::max(c_max, c->slots[a].C);
        const int64_t live_est = c->last_live >= 0 ? std::min(c->last_live, c_max) : (2 * c_max + 2) / 3;
        const int tiles_est = (int)std::max<int64_t>(1, (live_est + 63) / 64);
        int best_CH = 0, best_blocks = 0, best_score = 1 << 30;
        size_t best_lds = 0;
        static const int chs[3] = {3, 5, 8};
        for (int q = 0; q < 3; q++) {
            const int CH = chs[q];
            if (c->step_kernel_CH && c->step_kernel_CH != CH) continue;
            const size_t lds = std::max(c->lds_step, (size_t)(FX_BLOCK / 64) * sizeof(double) * 16 * (size_t)CH * (size_t)c->K_max_step);   // FX_OBST_LDS_DOUBLES(.., true)
            int cap = 0;
            {   // (one occupancy query per (CH, lds) of this process and device)
                static std::mutex mu;
                static std::map<std::tuple<int, int, size_t>, int> seen;
                std::lock_guard<std::mutex> lk(mu);
                const auto key = std::make_tuple(c->device, CH, lds);
                auto it = seen.find(key);
                if (it == seen.end()) {
                    int v = 0;
                    if (fx_step_kernel_capacity(CH, lds, &v) != hipSuccess) { (void)hipGetLastError(); v = 0; }
                    it = seen.emplace(key, v).first;
                }
                cap = it->second;
            }
            const int cap_agent = cap / std::max(c->n_agents, 1);
            if (cap_agent < c->max_blocks_step) continue;   // the walk alone does not fit at once: three launches
            const int NC = (c->S_max_step - 1 + CH - 1) / CH;
            const int64_t items = (int64_t)((tiles_est + FX_STEP_T - 1) / FX_STEP_T) * NC;   // (T tiles x one chunk per wave)
            const int blocks = (int)std::min<int64_t>(cap_agent, std::max<int64_t>(c->max_blocks_step, (items + FX_BLOCK / 64 - 1) / (FX_BLOCK / 64)));
            const int rounds = (int)((items + (int64_t)blocks * (FX_BLOCK / 64) - 1) / ((int64_t)blocks * (FX_BLOCK / 64)));
            const int score = rounds * CH;
            if (score < best_score) { best_score = score; best_CH = CH; best_blocks = blocks; best_lds = lds; }
        }
        if (best_CH) {
            StepArgs sa{};
            sa.bar = c->d_bar; sa.bar_base = c->bar_base;
            sa.host_result = c->h_counters_dev; sa.seq = c->seq; sa.dev_winner = winner;
            sa.host_pkg = c->pkg_step ? c->h_pkg_dev : nullptr; sa.pkg_stride = c->pkg_stride; sa.pkg_plane_rows = c->pkg_plane_rows;
            sa.walk_blocks = c->max_blocks_step;
            HIP_TRY(fx_launch_step(c->d_probs, c->n_agents, best_blocks, best_lds, best_CH, k0, k1, fuse, sa, c->stream));
            c->bar_base += (unsigned long long)best_blocks * (unsigned long long)c->n_agents;
            c->step_kernel_step = true;
            c->step_blocks = best_blocks; c->step_CH = best_CH; c->step_lds = best_lds;
            if (timed && !attached) HIP_TRY(hipEventRecord(ts->e_eval, c->stream));
            if (timed) { ts->obst_timed = false; ts->fused = true; c->n_timed++; }
            c->timed_step = timed;
            c->evaluated = true;
            c->in_flight = true;
            return FX_OK;
        }
    }
    if (c->eval_launched)
    {
        if (c->use_grid)
            HIP_TRY(fx_launch_eval_grid(c->d_probs, c->n_agents, c->max_blocks_step, c->block_step, c->lds_step, c->G_step,
                                        c->any_bundle, c->any_obst, c->wpe_step, c->wsplit_step, k0, k1, fuse, c->stream));
        else
            HIP_TRY(fx_launch_eval(c->d_probs, c->n_agents, c->max_blocks_step,
                                   sizeof(double) * ((size_t)c->M_max_step * FX_REF_FIELDS + FX_TP * (size_t)c->S_max_step +
                                                     (((size_t)c->M_max_step + 1) & ~(size_t)1)) + c->gen_rec_lds,
                                   c->G_step, c->any_bundle, c->any_obst, c->any_extra, c->wpe_step, k0, k1, fuse, c->stream));
    }
    if (timed && !attached) HIP_TRY(hipEventRecord(ts->e_eval, c->stream));
    if (timed) ts->obst_timed = false;
    if (c->split_step && c->obs_blocks_step > 0) {
        const bool t_obs = timed && c->timing == FX_TIMING_KERNEL;
        // one workgroup per tile (the chunks meet in LDS) where the horizon's chunks fit a workgroup; else single-wave items
        // (measured, tools/c3_split.py: config 3 34.2 -> 32.5 us, config 4's batch 15.6 -> 13.2 us; a config-5 agent with a bundle
        // -- 1 617 tiles -- 102 -> 159 us and the 1 M grid 374 -> 524 us: ten-wave workgroups schedule badly once there are more
        // tiles than the chip holds at once, so the automatic choice takes them up to 1 024 tiles per launch)
        const char *wg_env = getenv("FX_OBST_WG");   // experiments: 0 / 1 force single-wave items / workgroups
        const int wg_mode = wg_env ? (atoi(wg_env) ? 2 : 1) : ((int64_t)c->obs_tiles_step * c->n_agents <= 1024 ? 2 : 1);
        const size_t lds_wg = align_up((size_t)c->obs_wg_waves * (c->obs_lds_step + 64 * sizeof(double) + sizeof(unsigned long long)), 16);
        // the workgroup's waves each keep their slice of the staging area: five steps per item with 64 obstacles and eleven or more
        // chunks would ask for more than a CU's 160 KB (minus the kernel's static LDS) -- such a step runs as single-wave items
        const bool wg = wg_mode == 2 && c->obs_wg_waves >= 1 && c->obs_wg_waves <= 16 && lds_wg <= (size_t)160 * 1024 - 1024;
        c->obs_wg_step = wg ? c->obs_wg_waves : 0;
        HIP_TRY(fx_launch_obstacle(c->d_probs, c->n_agents, c->obs_blocks_step, wg ? lds_wg : c->obs_lds_step, c->split_CH,
                                   t_obs ? ts->e_obs0 : nullptr, t_obs ? ts->e_obs1 : nullptr, c->stream, wg ? c->obs_wg_waves : 0,
                                   c->obs_tiles_step));
        if (timed) ts->obst_timed = t_obs;
    }
    if (!c->fused_step) {
        // with a package the selection's publishing workgroup gathers the winner's arrays itself (no further launch)
        int64_t c_max = 0;
        for (int a = 0; a < c->n_agents; a++) c_max = std::max(c_max, c->slots[a].C);
        HIP_TRY(fx_launch_select(c->d_probs, c->n_agents, c_max, c->h_counters_dev, c->seq, winner, c->pkg_step ? c->h_pkg_dev : nullptr,
                                 c->pkg_stride, c->pkg_plane_rows, c->stream));
    } else if (c->pkg_step && !pkg_in_tail) {
        // fused selection without the tail's write-through hand-off (forced write-back plane stores) publishes while other waves'
        // plane stores may still be in flight: the gather runs as its own small kernel behind the evaluation; fx_finish waits for
        // its sequence word
        HIP_TRY(fx_launch_package(c->d_probs, c->n_agents, winner, c->h_pkg_dev, c->pkg_stride, c->pkg_plane_rows, c->seq, c->stream));
    }
    const bool one_launch = c->fused_step && (!c->pkg_step || pkg_in_tail);
    if (timed && !one_launch) HIP_TRY(hipEventRecord(ts->e_end, c->stream));
    if (timed) { ts->fused = one_launch; c->n_timed++; }
    c->timed_step = timed;
    c->evaluated = true;
    c->in_flight = true;
    return FX_OK;
}

int32_t fx_finish_batch(FxContext *c, FxResult *res) {
    if (!c || !res) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_finish: NULL argument");
    if (!c->evaluated) return set_err(FX_ERR_NOT_READY, "fx_finish before fx_evaluate");
    // wait for the sequence words the selection kernel publishes (bounded in TIME: fx_set_timeout_ms)
    for (int a = 0; a < c->n_agents; a++) {
        // with a winner package the last word to arrive is the package's (its kernel runs behind the selection)
        const volatile unsigned long long *sq = c->pkg_step
            ? reinterpret_cast<const unsigned long long *>(c->h_pkg + (size_t)a * c->pkg_stride + c->pkg_stride - 1)
            : c->h_counters + (size_t)a * (FX_CNT_COUNT + 1) + FX_CNT_COUNT;
        int rc = wait_seq(c, sq, c->seq);
        if (rc) return rc;
    }
    c->in_flight = false;
    // device time of this step: only if its events have already completed (a timed step never waits for them here;
    // fx_last_kernel_ms / fx_read_kernel_times do)
    double step_ms = -1.0;
    if (c->timed_step) {
        FxContext::TimeSlot &t = c->ring[(c->n_timed - 1) % FxContext::kTimeRing];
        if (t.fetched || hipEventQuery(t.fused ? t.e_eval : t.e_end) == hipSuccess) {
            int rc = fetch_slot(c, t);
            if (rc) return rc;
            step_ms = t.step_ms;
        } else (void)hipGetLastError();  // hipErrorNotReady is not an error
    }
    for (int a = 0; a < c->n_agents; a++) {
        const unsigned long long *cn = c->h_counters + (size_t)a * (FX_CNT_COUNT + 1);
        FxResult &r = res[a];
        memset(&r, 0, sizeof(r));
        r.n_candidates = c->slots[a].C;
        r.n_returned = (int64_t)cn[FX_CNT_RETURNED];
        r.n_feasible = (int64_t)cn[FX_CNT_FEASIBLE];
        r.n_infeasible = r.n_returned - r.n_feasible;
        for (int k = 0; k < FX_NUM_REASONS; k++) r.reason_hist[k] = (int64_t)cn[FX_CNT_HIST0 + k];
        r.best_index = cn[FX_CNT_BEST_IDX] == ~0ULL ? -1 : (int64_t)cn[FX_CNT_BEST_IDX];
        double bc;
        memcpy(&bc, &cn[FX_CNT_BEST_COST], sizeof(bc));
        r.best_cost = r.best_index < 0 ? 0.0 : bc;
        r.n_collisions = (int64_t)cn[FX_CNT_COLLISIONS];
        r.feasible_percentage = r.n_returned ? 100.0 * ((double)r.n_feasible / (double)r.n_returned) : 0.0;
        r.kernel_ms = step_ms;
        // (costed candidates of the step: the feasible ones, with draw_traj_set everything returned -- sizes the next one-launch step)
        const int64_t live = (c->slots[a].mode & FX_MODE_DRAW_TRAJ_SET) ? r.n_returned : r.n_feasible;
        c->last_live = a == 0 ? live : std::max(c->last_live, live);
    }
    return FX_OK;
}

int32_t fx_finish(FxContext *c, FxResult *res) { return fx_finish_batch(c, res); }

int32_t fx_step(FxContext *c, FxResult *res) {
    const int rc = fx_evaluate(c);
    return rc ? rc : fx_finish_batch(c, res);
}

// Per-step state of one agent of the uploaded batch (header: fxplan.h).
int32_t fx_update_state(FxContext *c, int32_t agent, const FxStateUpdate *u) {
    if (!c || !u) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_update_state: NULL argument");
    if (!c->uploaded) return set_err(FX_ERR_NOT_READY, "fx_update_state before fx_upload");
    if (agent < 0 || agent >= c->n_agents) return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d out of range", agent);
    if (c->in_flight) {  // a copy out of the staging block may still be running: let it land before rewriting its source
        HIP_TRY(hipSetDevice(c->device));
        { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
        c->in_flight = false;
    }
    FxAgentSlot &sl = c->slots[agent];
    DevProblem &d = c->h_probs[agent];
    // every argument is checked BEFORE anything is rewritten: an update that is refused leaves the context as it was
    if ((u->t_samp || u->v_samp || u->d_samp) && sl.off_t == (size_t)-1)
        return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d was uploaded with a sampling matrix: upload again", agent);
    if ((u->obs_pos || u->obs_cov_inv || u->obs_npred || u->obs_hull || u->obs_nhull) && sl.K <= 0)
        return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d was uploaded without obstacles: upload again", agent);
    if ((u->obs_hull || u->obs_nhull) && !sl.have_hull)
        return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d was uploaded without obstacle hulls: upload again", agent);
    if ((u->t_samp && u->nT && u->nT != sl.nT) || (u->v_samp && u->nV && u->nV != sl.nV) || (u->d_samp && u->nD && u->nD != sl.nD))
        return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d: sampling arrays of %d x %d x %d values, uploaded %d x %d x %d: upload again", agent,
                       u->nT, u->nV, u->nD, sl.nT, sl.nV, sl.nD);
    if ((u->obs_pos || u->obs_cov_inv || u->obs_npred || u->obs_hull || u->obs_nhull) && ((u->K && u->K != sl.K) || (u->P && u->P != sl.P)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d: obstacle arrays for K = %d, P = %d, uploaded K = %d, P = %d: upload again", agent,
                       u->K, u->P, sl.K, sl.P);
    auto touch = [&](size_t off, size_t bytes) {
        c->dirty_lo = std::min(c->dirty_lo, off);
        c->dirty_hi = std::max(c->dirty_hi, off + bytes);
    };
    bool origin_moved = false;
    if (u->x0_lon) {
        origin_moved = d.x0_lon[0] != u->x0_lon[0];
        memcpy(d.x0_lon, u->x0_lon, sizeof(d.x0_lon));
    }
    if (u->x0_lat) memcpy(d.x0_lat, u->x0_lat, sizeof(d.x0_lat));
    if (u->x0_orientation == u->x0_orientation) d.x0_orientation = u->x0_orientation;
    if (u->v_des == u->v_des) d.v_des = u->v_des;
    if (u->low_vel_mode >= 0) d.low_vel_mode = u->low_vel_mode;
    if (u->t_samp || u->v_samp || u->d_samp) {
        if (u->t_samp) { memcpy(c->h_in + sl.off_t, u->t_samp, sizeof(double) * sl.nT); touch(sl.off_t, sizeof(double) * sl.nT); }
        if (u->v_samp) { memcpy(c->h_in + sl.off_v, u->v_samp, sizeof(double) * sl.nV); touch(sl.off_v, sizeof(double) * sl.nV); }
        if (u->d_samp) { memcpy(c->h_in + sl.off_d, u->d_samp, sizeof(double) * sl.nD); touch(sl.off_d, sizeof(double) * sl.nD); }
    }
    const bool new_obs = u->obs_pos || u->obs_cov_inv || u->obs_npred || u->obs_hull || u->obs_nhull;
    if (new_obs || (origin_moved && sl.K > 0)) {
        const int K = sl.K, P = sl.P, S = sl.S;
        double *pos = reinterpret_cast<double *>(c->h_in + sl.off_pos), *cov = reinterpret_cast<double *>(c->h_in + sl.off_cov);
        int32_t *npred = reinterpret_cast<int32_t *>(c->h_in + sl.off_npred);
        if (u->obs_pos) memcpy(pos, u->obs_pos, sizeof(double) * 2 * K * P);
        if (u->obs_cov_inv) memcpy(cov, u->obs_cov_inv, sizeof(double) * 4 * K * P);
        if (u->obs_npred) memcpy(npred, u->obs_npred, sizeof(int32_t) * K);
        double *hull = nullptr;
        int32_t *nhull = nullptr;
        if (sl.have_hull) {
            hull = reinterpret_cast<double *>(c->h_in + sl.off_hull);
            nhull = reinterpret_cast<int32_t *>(c->h_in + sl.off_nhull);
            if (u->obs_hull) memcpy(hull, u->obs_hull, sizeof(double) * 6 * K * (P - 1));
            if (u->obs_nhull) memcpy(nhull, u->obs_nhull, sizeof(int32_t) * K);
        }
        hot_origin_of(reinterpret_cast<const double *>(c->h_in + sl.off_ref), sl.M, d.x0_lon[0], d.hot_origin);
        d.hot_gap_margin = pack_obstacle_tables(S, K, P, pos, cov, npred, hull, nhull, sl.have_hull, d.hot_origin[0], d.hot_origin[1],
                                                reinterpret_cast<double *>(c->h_in + sl.off_rec),
                                                reinterpret_cast<unsigned long long *>(c->h_in + sl.off_pm),
                                                reinterpret_cast<unsigned long long *>(c->h_in + sl.off_hm),
                                                reinterpret_cast<double *>(c->h_in + sl.off_hot));
        // the generic kernel also reads the raw predictions (windowed costs); the grid kernel only the packed tables
        touch(sl.off_rec, (c->use_grid ? sl.dyn_end : sl.raw_end) - sl.off_rec);
    }
    c->probs_dirty = true;
    return FX_OK;
}

int32_t fx_update_step(FxContext *c, const FxStateUpdate *u, FxResult *res) {
    int rc = fx_update_state(c, 0, u);
    if (rc) return rc;
    return fx_step(c, res);
}

static int check_agent(FxContext *c, int a);

// ---- survivor exchange inside the library (header: fxplan.h) ----
// RCCL is bound at run time (dlopen of librccl.so.1: the copy the process already has -- torch's -- or the system one), so the
// library loads and plans on a single GPU without it.
namespace {
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
            r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));   // optional
            r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
        }
    });
    return &r;
}
#define RCCL_TRY(expr)                                                                                             \
    do {                                                                                                            \
        const int e_ = (expr);                                                                                      \
        if (e_ != 0) return set_err(FX_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(e_));              \
    } while (0)
}  // namespace

int32_t fx_comm_unique_id(uint8_t *id128) {
    if (!id128) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_comm_unique_id: NULL argument");
    if (!rccl()->ok) return set_err(FX_ERR_NOT_READY, "librccl.so.1 not available");
    Rccl::UniqueId id;
    RCCL_TRY(rccl()->GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return FX_OK;
}

// Local preconditions of fx_comm_init, WITHOUT entering anything collective: every rank calls this first and the ranks agree
// (e.g. an all-reduce MIN over the host program's own group) before any of them calls fx_comm_init -- a rank that would fail
// there never reaches ncclCommInitRank, and its peers would wait for it forever.
int32_t fx_comm_check(const FxContext *c, int32_t world) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_comm_check: NULL argument");
    if (world < 1) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_comm_check: world %d", world);
    if ((size_t)world * c->max_agents * 2 > FX_PUB_MAX) return set_err(FX_ERR_CAPACITY, "%d ranks x %d agents exceed the publication block", world, c->max_agents);
    if (!rccl()->ok) return set_err(FX_ERR_NOT_READY, "librccl.so.1 not available");
    if (c->comm) return set_err(FX_ERR_INVALID_ARGUMENT, "this context already has a communicator");
    if (c->comm_init_failed)
        return set_err(FX_ERR_TIMEOUT, "an earlier fx_comm_init on this context timed out (its helper thread may still be inside "
                       "ncclCommInitRank): no second attempt -- use another exchange and leave the process through its exit path");
    return FX_OK;
}

int32_t fx_comm_init(FxContext *c, const uint8_t *id128, int32_t rank, int32_t world) {
    if (!c || !id128) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_comm_init: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_comm_init: rank %d of %d", rank, world);
    int rc = fx_comm_check(c, world);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    // everything that can fail locally comes BEFORE the collective call
    const size_t need = (size_t)world * c->max_agents * 2;
    if (c->d_gather && c->gather_cap < need) {   // (left by an earlier communicator of this context)
        (void)hipFree(c->d_gather);
        c->dev_bytes -= (int64_t)(sizeof(double) * c->gather_cap);
        c->d_gather = nullptr; c->gather_cap = 0;
    }
    if (!c->d_gather) {
        if ((rc = dev_alloc(c, &c->d_gather, need))) return rc;
        c->gather_cap = need;
    }
    if (!c->d_winner_own && (rc = dev_alloc(c, &c->d_winner_own, (size_t)c->max_agents * 2))) return rc;
    if (!c->d_xsend && (rc = dev_alloc(c, &c->d_xsend, (size_t)c->max_agents * 2 * 64))) return rc;
    // ncclCommInitRank is a blocking collective without a time bound of its own: a peer that never arrives (or a fabric that
    // never answers) would hold this thread forever.  It runs on a helper thread; this one waits for it with the context's time
    // bound and, past it, gives the communicator up (the helper is left behind, detached, with its state) -- the caller falls
    // back to another exchange instead of hanging the job.
    struct InitState {
        std::atomic<int> done{0};
        int rc = 0;
        void *comm = nullptr;
        Rccl::UniqueId id;
    };
    auto st = std::make_shared<InitState>();
    memcpy(st->id.internal, id128, 128);
    const int device = c->device;
    std::thread([st, world, rank, device] {
        (void)hipSetDevice(device);
        st->rc = rccl()->CommInitRank(&st->comm, world, st->id, rank);
        st->done.store(1, std::memory_order_release);
    }).detach();
    const auto t0 = std::chrono::steady_clock::now();
    const auto limit = std::chrono::milliseconds(c->timeout_ms > 0 ? c->timeout_ms : 20000);
    while (!st->done.load(std::memory_order_acquire)) {
        if (std::chrono::steady_clock::now() - t0 >= limit) {
            // the helper thread stays inside ncclCommInitRank (it may even finish later: that communicator is never used and
            // never destroyed).  No second attempt on this context; the process should leave through distributed.exit_on_timeout
            // / os._exit rather than a normal interpreter teardown that would wait for RCCL's threads.
            c->comm_init_failed = true;
            return set_err(FX_ERR_TIMEOUT, "fx_comm_init: ncclCommInitRank did not return within %d ms (rank %d of %d)",
                           (int)limit.count(), rank, world);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (st->rc != 0) return set_err(FX_ERR_HIP, "ncclCommInitRank failed: %s (rank %d of %d)", rccl()->GetErrorString(st->rc), rank, world);
    c->comm = st->comm;
    c->comm_rank = rank; c->comm_world = world;
    c->comm_agents = c->max_agents;
    c->comm_rows_clean = c->comm_agents; c->comm_k_clean = -1;   // nothing known about the send buffers yet
    return FX_OK;
}

// The number of agent rows every rank contributes to an exchange.  The element count of the all-gather must be the same on
// every rank whatever a rank's own step does, so it is a property of the communicator, fixed here (default: the context's
// max_agents) -- not of the rank's current upload.  The ranks agree on it before they call this (distributed.py).
int32_t fx_comm_set_agents(FxContext *c, int32_t n_agents) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_comm_set_agents: NULL argument");
    if (!c->comm) return set_err(FX_ERR_NOT_READY, "fx_comm_set_agents before fx_comm_init");
    if (n_agents < 1 || n_agents > c->max_agents) return set_err(FX_ERR_CAPACITY, "fx_comm_set_agents: %d outside [1, %d]", n_agents, c->max_agents);
    if ((size_t)c->comm_world * n_agents * 2 > FX_PUB_MAX) return set_err(FX_ERR_CAPACITY, "%d ranks x %d agents exceed the publication block", c->comm_world, n_agents);
    c->comm_agents = n_agents;
    c->comm_k_clean = -1;
    return FX_OK;
}

// out[0] rank, [1] world, [2] ranks RCCL itself reports for the communicator (ncclCommCount; -1 if unavailable), [3] agent rows per rank
int32_t fx_comm_info(const FxContext *c, int32_t *out4) {
    if (!c || !out4) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_comm_info: NULL argument");
    if (!c->comm) return set_err(FX_ERR_NOT_READY, "fx_comm_info before fx_comm_init");
    int n = -1;
    if (rccl()->CommCount && rccl()->CommCount(c->comm, &n) != 0) n = -1;
    out4[0] = c->comm_rank; out4[1] = c->comm_world; out4[2] = n; out4[3] = c->comm_agents;
    return FX_OK;
}

int32_t fx_comm_destroy(FxContext *c) {
    if (!c || !c->comm) return FX_OK;
    if (c->timed_out) { c->comm = nullptr; return FX_OK; }   // (ncclCommDestroy would wait for the collective that never finishes)
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)rccl()->CommDestroy(c->comm);
    c->comm = nullptr;
    if (c->d_gather) { (void)hipFree(c->d_gather); c->d_gather = nullptr; c->gather_cap = 0; }
    return FX_OK;
}

// "No survivor" -- (inf, -1) -- in rows [first, comm_agents) of the exchange's send buffer (k = 0: the winner buffer
// [agents][2]; k > 0: [cost agents x k | index agents x k]).  Enqueued on the context's stream; returns a HIP status.
static hipError_t fill_no_survivor(FxContext *c, int first, int k) {
    const int A = c->comm_agents;
    if (first >= A) return hipSuccess;
    double *h = c->h_topk_cost;        // pinned [max_agents][64]
    long long *hi = c->h_topk_idx;
    if (k == 0) {
        for (int a = first; a < A; a++) { h[2 * a] = INFINITY; const long long m1 = -1; memcpy(&h[2 * a + 1], &m1, sizeof(m1)); }
        return hipMemcpyAsync(c->d_winner_own + 2 * first, h + 2 * first, sizeof(double) * 2 * (size_t)(A - first), hipMemcpyHostToDevice, c->stream);
    }
    const size_t e0 = (size_t)first * k, e1 = (size_t)A * k;
    for (size_t e = e0; e < e1; e++) { h[e] = INFINITY; hi[e] = -1; }
    hipError_t e = hipMemcpyAsync(c->d_xsend + e0, h + e0, sizeof(double) * (e1 - e0), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(reinterpret_cast<long long *>(c->d_xsend + e1) + e0, hi + e0, sizeof(long long) * (e1 - e0), hipMemcpyHostToDevice, c->stream);
}

// Where the all-gather lands and how its arrival is signalled.  Mode 0: receive buffer in device memory, then fx_publish_kernel
// copies it into the pinned block and releases the sequence word (one more launch: + 7.5 us at one rank).  Mode 1: the receive
// buffer IS the pinned, mapped block (its device address), and the sequence word behind it is written by a stream-ordered memory
// operation (hipStreamWriteValue64) -- no launch.  Mode 1 is only used after it has agreed with the torch.distributed exchange
// on every rank (distributed.ShardedEvaluator.crosscheck_exchange tries it first and falls back to mode 0, then to torch).
int32_t fx_set_exchange_mode(FxContext *c, int32_t mode) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (mode != 0 && mode != 1) return set_err(FX_ERR_INVALID_ARGUMENT, "exchange mode must be 0 (device receive + publication kernel) or 1 (receive in the pinned block)");
    c->exchange_mode = mode;
    return FX_OK;
}
static double *exchange_recv(FxContext *c) { return c->exchange_mode == 1 ? c->h_pub_dev : c->d_gather; }
static int exchange_signal(FxContext *c, int32_t total) {
    if (c->exchange_mode != 1) return fx_publish(c, c->d_gather, total);
    c->pub_seq++;
    c->pub_n = total;
    HIP_TRY(hipStreamWriteValue64(c->stream, c->h_pub_dev + FX_PUB_MAX, c->pub_seq, 0));
    c->in_flight = true; c->tail_work = true;
    return FX_OK;
}

// One plan step of every rank: evaluation (+ selection), ONE all-gather of the ranks' winners (cost f64, global index i64 per
// agent row; 16 B per rank and row) on the context's stream, publication to pinned host memory -- enqueued back to back, then the
// host takes the local result block while the collective runs and waits (bounded in time) for the gathered winners.
// The element count of the collective is the communicator's (fx_comm_set_agents), the same on every rank whatever this rank's
// step does: a rank whose evaluation fails, or whose upload does not fit the agreed rows, STILL enters the all-gather -- with
// (inf, -1) in its rows -- and returns its error afterwards; nothing that can fail locally returns ahead of the collective.
int32_t fx_step_exchange(FxContext *c, FxResult *res, double *cost, int64_t *index) {
    if (!c || !res || !cost || !index) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_step_exchange: NULL argument");
    if (!c->comm) return set_err(FX_ERR_NOT_READY, "fx_step_exchange before fx_comm_init");
    if (c->timed_out) return set_err(FX_ERR_TIMEOUT, "an earlier wait on this context timed out: destroy it");
    const int A = c->comm_agents;
    int rc_local = FX_OK;
    char err_local[sizeof(g_err)];
    auto keep = [&](int rc) { if (rc && !rc_local) { rc_local = rc; memcpy(err_local, g_err, sizeof(err_local)); } };
    auto keep_hip = [&](hipError_t e, const char *what) { if (e != hipSuccess) keep(set_err(FX_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e))); };
    keep_hip(hipSetDevice(c->device), "hipSetDevice");
    if (c->n_agents > A) keep(set_err(FX_ERR_CAPACITY, "%d uploaded agents but the communicator exchanges %d rows per rank (fx_comm_set_agents)", c->n_agents, A));
    bool evaluated = false;
    if (!rc_local) {
        double *saved = c->dev_winner;
        c->dev_winner = c->d_winner_own;   // the selection leaves (cost, index) of every agent here
        const int rc = fx_evaluate(c);
        c->dev_winner = saved;
        keep(rc);
        evaluated = rc == FX_OK;
        if (evaluated && saved)   // a caller-owned winner buffer (fx_set_winner_buffer) gets its copy
            keep_hip(hipMemcpyAsync(saved, c->d_winner_own, sizeof(double) * 2 * (size_t)c->n_agents, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
    }
    const int n_mine = evaluated ? c->n_agents : 0;   // rows this rank fills; the others say "no survivor"
    if (c->comm_k_clean != 0 || c->comm_rows_clean > n_mine) {
        keep_hip(fill_no_survivor(c, n_mine, 0), "hipMemcpyAsync");
        c->comm_k_clean = 0;
    }
    c->comm_rows_clean = n_mine;
    const int n = A * 2, total = n * c->comm_world;
    RCCL_TRY(rccl()->AllGather(c->d_winner_own, exchange_recv(c), (size_t)n, /*ncclDouble*/ 8, c->comm, c->stream));
    int rc;
    if ((rc = exchange_signal(c, total))) return rc;
    if (evaluated && (rc = fx_finish_batch(c, res))) keep(rc);
    if ((rc = wait_seq(c, reinterpret_cast<const unsigned long long *>(c->h_pub + FX_PUB_MAX), c->pub_seq))) return rc;
    for (int r = 0; r < c->comm_world; r++)
        for (int a = 0; a < A; a++) {
            const double *q = c->h_pub + (size_t)r * n + 2 * a;
            cost[(size_t)r * A + a] = q[0];
            memcpy(&index[(size_t)r * A + a], &q[1], sizeof(int64_t));
        }
    if (rc_local) { memcpy(g_err, err_local, sizeof(err_local)); return rc_local; }
    return FX_OK;
}

// The same for the k best survivors per agent (BASELINE config 5: per-agent top-32 over 8 GPUs): evaluation, selection, the two
// top-k launches writing [cost A x k | index A x k] into the send buffer, ONE all-gather of 16 k bytes per rank and agent row,
// publication, results -- no host code between the launches.  cost / index: [world][A][k] (A = the communicator's agent rows),
// index -1 where a rank has fewer than k survivors.  k must be the same on every rank (the ranks agree on it beforehand).
int32_t fx_step_exchange_topk(FxContext *c, int32_t k, FxResult *res, double *cost, int64_t *index) {
    if (!c || !res || !cost || !index) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_step_exchange_topk: NULL argument");
    if (k < 1 || k > 64) return set_err(FX_ERR_INVALID_ARGUMENT, "k=%d outside [1,64]", k);
    if (!c->comm) return set_err(FX_ERR_NOT_READY, "fx_step_exchange_topk before fx_comm_init");
    if (c->timed_out) return set_err(FX_ERR_TIMEOUT, "an earlier wait on this context timed out: destroy it");
    const int A = c->comm_agents;
    const size_t n = (size_t)A * 2 * k, total = n * c->comm_world;
    // what depends only on (communicator, k) is the same on every rank: these refusals happen everywhere or nowhere
    if (total > FX_PUB_MAX) return set_err(FX_ERR_CAPACITY, "%d ranks x %d agents x %d survivors exceed the publication block", c->comm_world, A, k);
    int rc_local = FX_OK;
    char err_local[sizeof(g_err)];
    auto keep = [&](int rc) { if (rc && !rc_local) { rc_local = rc; memcpy(err_local, g_err, sizeof(err_local)); } };
    auto keep_hip = [&](hipError_t e, const char *what) { if (e != hipSuccess) keep(set_err(FX_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e))); };
    keep_hip(hipSetDevice(c->device), "hipSetDevice");
    if (total > c->gather_cap) {   // (a function of (communicator, k) as well; a failed allocation leaves the old buffer in place)
        double *bigger = nullptr;
        if (hipStreamSynchronize(c->stream) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&bigger), sizeof(double) * (size_t)FX_PUB_MAX) == hipSuccess) {
            if (c->d_gather) { (void)hipFree(c->d_gather); c->dev_bytes -= (int64_t)(sizeof(double) * c->gather_cap); }
            c->d_gather = bigger; c->gather_cap = FX_PUB_MAX; c->dev_bytes += (int64_t)(sizeof(double) * (size_t)FX_PUB_MAX);
        } else {
            (void)hipGetLastError();
            // without a receive buffer of the agreed size this rank cannot take part: the one failure that cannot be carried
            // through the collective (its peers run into their time bound)
            return set_err(FX_ERR_HIP, "fx_step_exchange_topk: no memory for the %zu-element receive buffer", (size_t)FX_PUB_MAX);
        }
    }
    if (c->n_agents > A) keep(set_err(FX_ERR_CAPACITY, "%d uploaded agents but the communicator exchanges %d rows per rank (fx_comm_set_agents)", c->n_agents, A));
    long long *send_idx = reinterpret_cast<long long *>(c->d_xsend + (size_t)A * k);
    bool evaluated = false;
    if (!rc_local) {
        const int rc = fx_evaluate(c);
        keep(rc);
        if (rc == FX_OK) {
            const hipError_t e = fx_launch_topk(c->d_probs, c->n_agents, max_candidates_of(c), k, c->d_topk_scr_cost, c->d_topk_scr_idx, c->d_xsend, send_idx, c->stream);
            keep_hip(e, "fx_launch_topk");
            evaluated = e == hipSuccess;
        }
    }
    const int n_mine = evaluated ? c->n_agents : 0;
    if (c->comm_k_clean != k || c->comm_rows_clean > n_mine) {
        keep_hip(fill_no_survivor(c, n_mine, k), "hipMemcpyAsync");
        c->comm_k_clean = k;
    }
    c->comm_rows_clean = n_mine;
    RCCL_TRY(rccl()->AllGather(c->d_xsend, exchange_recv(c), n, /*ncclDouble*/ 8, c->comm, c->stream));
    int rc;
    if ((rc = exchange_signal(c, (int32_t)total))) return rc;
    if (evaluated && (rc = fx_finish_batch(c, res))) keep(rc);
    if ((rc = wait_seq(c, reinterpret_cast<const unsigned long long *>(c->h_pub + FX_PUB_MAX), c->pub_seq))) return rc;
    const size_t nk = (size_t)A * k;
    for (int r = 0; r < c->comm_world; r++) {
        const double *q = c->h_pub + (size_t)r * n;
        memcpy(cost + (size_t)r * nk, q, sizeof(double) * nk);
        memcpy(index + (size_t)r * nk, q + nk, sizeof(int64_t) * nk);
    }
    if (rc_local) { memcpy(g_err, err_local, sizeof(err_local)); return rc_local; }
    return FX_OK;
}

// ---- winner package (header: fxplan.h) ----
int32_t fx_set_package(FxContext *c, int32_t enabled) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    c->package_enabled = enabled != 0;
    return FX_OK;
}

int32_t fx_read_package(FxContext *c, int32_t agent, double yaw_rate0, FxPackage *pkg, double *block) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    if (!pkg) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_read_package: NULL argument");
    if (!c->pkg_step) return set_err(FX_ERR_NOT_READY, "the last step ran without a winner package (fx_set_package, FX_MODE_WRITE_BUNDLE)");
    if (c->in_flight) {  // fx_finish has not been called for this step: wait for the package word here
        rc = wait_seq(c, reinterpret_cast<const unsigned long long *>(c->h_pkg + (size_t)agent * c->pkg_stride + c->pkg_stride - 1), c->seq);
        if (rc) return rc;
    }
    const FxAgentSlot &sl = c->slots[agent];
    const DevProblem &d = c->h_probs[agent];
    const double *src = c->h_pkg + (size_t)agent * c->pkg_stride, *tail = src + c->pkg_plane_rows;
    const int S = sl.S;
    memset(pkg, 0, sizeof(*pkg));
    pkg->S = S;
    pkg->n_cost = sl.n_cost;
    pkg->index = -1;
    pkg->found = tail[16 + FX_NUM_COSTS] != 0.0;
    if (!pkg->found) return FX_OK;
    memcpy(pkg->coeff_lon, tail, sizeof(double) * 6);
    memcpy(pkg->coeff_lat, tail + 6, sizeof(double) * 6);
    memcpy(pkg->raw_costs, tail + 12, sizeof(double) * FX_NUM_COSTS);
    pkg->cost = tail[12 + FX_NUM_COSTS];
    pkg->traj_len = (int32_t)tail[13 + FX_NUM_COSTS];
    pkg->flags = (uint32_t)tail[14 + FX_NUM_COSTS];
    pkg->index = (int64_t)tail[15 + FX_NUM_COSTS];
    pkg->tau_lat = tail[17 + FX_NUM_COSTS];
    if (!block) return FX_OK;
    memcpy(block, src, sizeof(double) * FX_NUM_PLANES * S);
    // the derived columns of planner.py:394-447 (_compute_trajectory_pair): yaw rate by backward differences of the heading,
    // steering angle of the kinematic single-track model, heading shifted into [x0_orientation - pi, x0_orientation + pi]
    const double *theta = block + 2 * (size_t)S, *kappa = block + 5 * (size_t)S;
    double *yaw = block + (size_t)FX_NUM_PLANES * S, *steer = yaw + S, *orient = steer + S;
    const double lo = d.x0_orientation - M_PI, hi = d.x0_orientation + M_PI, wb = d.veh.wheelbase;
    for (int i = 0; i < S; i++) {
        yaw[i] = i == 0 ? yaw_rate0 : (theta[i] - theta[i - 1]) / d.dt;
        steer[i] = std::atan2(wb * kappa[i], 1.0);
        double o = theta[i];
        for (int r = 0; r < 4; r++) {
            if (o < lo) o += 2 * M_PI;
            if (o > hi) o -= 2 * M_PI;
        }
        orient[i] = o;
    }
    return FX_OK;
}

int32_t fx_plan_and_package(FxContext *c, const FxStateUpdate *upd, double yaw_rate0, FxResult *res, FxPackage *pkg, double *block) {
    if (!c || !res || !pkg) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_plan_and_package: NULL argument");
    int rc;
#ifdef FX_HOST_PROBE   // probe builds: where the host side of a planner step goes (tools/probe_build)
    static double acc[4]; static int n_acc;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
#endif
    if (upd && (rc = fx_update_state(c, 0, upd))) return rc;
#ifdef FX_HOST_PROBE
    const double t1 = now();
#endif
    const bool was = c->package_enabled;
    c->package_enabled = true;
    rc = fx_evaluate(c);
    c->package_enabled = was;
    if (rc) return rc;
#ifdef FX_HOST_PROBE
    const double t2 = now();
#endif
    if ((rc = fx_finish_batch(c, res))) return rc;
#ifdef FX_HOST_PROBE
    const double t3 = now();
    rc = fx_read_package(c, 0, yaw_rate0, pkg, block);
    const double t4 = now();
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3;
    if (++n_acc == 200) {
        fprintf(stderr, "fx_plan_and_package: update_state %.1f us, evaluate (launches) %.1f us, finish (wait) %.1f us, read_package %.1f us\n",
                acc[0] / n_acc, acc[1] / n_acc, acc[2] / n_acc, acc[3] / n_acc);
        acc[0] = acc[1] = acc[2] = acc[3] = 0; n_acc = 0;
    }
    return rc;
#else
    return fx_read_package(c, 0, yaw_rate0, pkg, block);
#endif
}

int32_t fx_plan_batch_begin(FxContext *c, int32_t n_agents, const FxStateUpdate *const *upd) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_plan_batch_begin: NULL argument");
    if (!c->uploaded) return set_err(FX_ERR_NOT_READY, "fx_plan_batch_begin before fx_upload");
    if (n_agents != c->n_agents)
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_plan_batch_begin: %d agents, the uploaded batch has %d", n_agents, c->n_agents);
    int rc;
    if (upd)
        for (int a = 0; a < n_agents; a++)
            if (upd[a] && (rc = fx_update_state(c, a, upd[a]))) return rc;
    const bool was = c->package_enabled;
    c->package_enabled = true;
    rc = fx_evaluate(c);
    c->package_enabled = was;
    return rc;
}

int32_t fx_plan_batch_end(FxContext *c, int32_t n_agents, const double *yaw_rate0, FxResult *res, FxPackage *pkg, double *const *blocks) {
    if (!c || !res || !pkg) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_plan_batch_end: NULL argument");
    if (n_agents != c->n_agents)
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_plan_batch_end: %d agents, the uploaded batch has %d", n_agents, c->n_agents);
    int rc;
    if ((rc = fx_finish_batch(c, res))) return rc;
    for (int a = 0; a < n_agents; a++)
        if ((rc = fx_read_package(c, a, yaw_rate0 ? yaw_rate0[a] : 0.0, pkg + a, blocks ? blocks[a] : nullptr))) return rc;
    return FX_OK;
}

int32_t fx_plan_batch_packaged(FxContext *c, int32_t n_agents, const FxStateUpdate *const *upd, const double *yaw_rate0, FxResult *res,
                               FxPackage *pkg, double *const *blocks) {
    if (!c || !res || !pkg) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_plan_batch_packaged: NULL argument");
#ifdef FX_HOST_PROBE   // probe builds: where the host side of a batched planner step goes (tools/probe_build)
    static double acc[2]; static int n_acc;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
#endif
    int rc = fx_plan_batch_begin(c, n_agents, upd);
    if (rc) return rc;
#ifdef FX_HOST_PROBE
    const double t1 = now();
#endif
    rc = fx_plan_batch_end(c, n_agents, yaw_rate0, res, pkg, blocks);
#ifdef FX_HOST_PROBE
    acc[0] += t1 - t0; acc[1] += now() - t1;
    if (++n_acc == 20) {
        fprintf(stderr, "fx_plan_batch_packaged: begin (state updates, launches) %.1f us, end (wait, packages) %.1f us\n", acc[0] / n_acc,
                acc[1] / n_acc);
        acc[0] = acc[1] = 0; n_acc = 0;
    }
#endif
    return rc;
}

// ---- host geometry of the callers either side of the path ----
// (s, d) of a Cartesian point along the reference polyline (planner.py:574-578 convert_to_curvilinear_coords): on every
// segment k the foot point P_k + lam b and the interpolated normal n_k + lam dn are collinear with the point where
// cross(a + lam b, n_k + lam dn) = 0, a quadratic in lam; of all roots in [0, 1] the one with the smallest |d| wins.
int32_t fx_cs_to_curvilinear(int32_t M, const double *ref_xy, const double *normals, const double *ref_pos, double x, double y, double *sd) {
    return fx_cs_to_curvilinear_ex(M, ref_xy, normals, ref_pos, x, y, 0, sd);
}
int32_t fx_cs_to_curvilinear_ex(int32_t M, const double *ref_xy, const double *normals, const double *ref_pos, double x, double y,
                                int32_t pseudo_normal, double *sd) {
    if (M < 2 || !ref_xy || !normals || !ref_pos || !sd) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_cs_to_curvilinear: bad argument");
    bool have = false;
    double best_s = 0, best_d = 0;
    for (int k = 0; k + 1 < M; k++) {
        const double ax = ref_xy[2 * k] - x, ay = ref_xy[2 * k + 1] - y;
        const double bx = ref_xy[2 * k + 2] - ref_xy[2 * k], by = ref_xy[2 * k + 3] - ref_xy[2 * k + 1];
        const double n0x = normals[2 * k], n0y = normals[2 * k + 1];
        const double dnx = normals[2 * k + 2] - n0x, dny = normals[2 * k + 3] - n0y;
        const double c2 = bx * dny - by * dnx;
        const double c1 = ax * dny - ay * dnx + bx * n0y - by * n0x;
        const double c0 = ax * n0y - ay * n0x;
        double roots[2];
        int nr = 0;
        if (std::fabs(c2) < 1e-14) {
            if (!(std::fabs(c1) < 1e-300)) roots[nr++] = -c0 / c1;
        } else {
            const double disc = c1 * c1 - 4 * c2 * c0;
            if (disc < 0) continue;
            const double sq = std::sqrt(disc);
            roots[nr++] = (-c1 + sq) / (2 * c2);
            roots[nr++] = (-c1 - sq) / (2 * c2);
        }
        for (int r = 0; r < nr; r++) {
            double lam = roots[r];
            if (!(lam >= -1e-12 && lam <= 1 + 1e-12)) continue;
            lam = std::fmin(std::fmax(lam, 0.0), 1.0);
            const double fx = ref_xy[2 * k] + lam * bx, fy = ref_xy[2 * k + 1] + lam * by;
            double nx = n0x + lam * dnx, ny = n0y + lam * dny;
            // (x, y) = foot + d n / |n|  (or foot + d n: the pseudo-distance variant)  =>  d = (p - foot) . n / |n|  (/ |n|^2)
            const double nn = pseudo_normal ? nx * nx + ny * ny : std::sqrt(nx * nx + ny * ny);
            nx = nx / nn; ny = ny / nn;
            const double dd = (x - fx) * nx + (y - fy) * ny;
            if (!have || std::fabs(dd) < std::fabs(best_d)) {
                have = true;
                best_d = dd;
                best_s = ref_pos[k] + lam * (ref_pos[k + 1] - ref_pos[k]);
            }
        }
    }
    if (!have) return set_err(FX_ERR_INVALID_ARGUMENT, "point outside projection domain");
    sd[0] = best_s; sd[1] = best_d;
    return FX_OK;
}

// Inverses of n 2x2 matrices with the arithmetic of np.linalg.inv (LAPACK gesv on the identity as OpenBLAS executes it:
// partial pivoting, the multiplier and both divisions through reciprocals, one fused multiply-add in the back substitution)
// -- bit-identical to NumPy's result (collision_probability.py:281 inverts the prediction covariances with it; the CPU
// suite compares the two on random matrices).  Returns FX_ERR_INVALID_ARGUMENT for a singular matrix (NumPy: LinAlgError).
int32_t fx_invert_cov2(int32_t n, const double *m, double *out) {
    if (n < 0 || (n > 0 && (!m || !out))) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_invert_cov2: bad argument");
    for (int i = 0; i < n; i++) {
        const double a = m[4 * i], b = m[4 * i + 1], c = m[4 * i + 2], d = m[4 * i + 3];
        const bool sw = std::fabs(c) > std::fabs(a);
        const double p0a = sw ? c : a, p0b = sw ? d : b, p1a = sw ? a : c, p1b = sw ? b : d;
        if (p0a == 0.0) return set_err(FX_ERR_INVALID_ARGUMENT, "singular matrix (%d)", i);
        const double rp = 1.0 / p0a;
        const double l = p1a * rp;
        const double u11 = p1b - l * p0b;
        if (u11 == 0.0) return set_err(FX_ERR_INVALID_ARGUMENT, "singular matrix (%d)", i);
        const double ru = 1.0 / u11;
        for (int col = 0; col < 2; col++) {
            const double r0 = sw ? (col == 1) : (col == 0), r1 = sw ? (col == 0) : (col == 1);
            const double x1 = (r1 - l * r0) * ru;
            out[4 * i + col] = std::fma(-p0b, x1, r0) * rp;
            out[4 * i + 2 + col] = x1;
        }
    }
    return FX_OK;
}

// Packing of K predicted obstacles (prediction_helpers.py:209-261 dict entries) into the arrays FxProblem / FxStateUpdate take,
// in one call: obstacle k has n[k] predictions at pos[k] ([n][2]), cov[k] ([n][4]) and -- when yaw[k] is not NULL -- headings
// yaw[k] ([n]) with the box length[k] x width[k].  Outputs with stride P (zero-filled here): pos_out [K][P][2], cov_inv_out
// [K][P][4] (fx_invert_cov2), npred [K] = n[k] (the real length decides which ego steps see the obstacle,
// collision_probability.py:287), hull [K][P-1][6], nhull [K] (hulls over the first min(n_samples, n[k], P) boxes,
// collision_check.py:150).
int32_t fx_pack_predictions(int32_t K, int32_t P, int32_t n_samples, const int32_t *n, const double *const *pos, const double *const *cov,
                            const double *const *yaw, const double *length, const double *width, double *pos_out, double *cov_inv_out,
                            int32_t *npred, double *hull, int32_t *nhull) {
    if (K < 0 || P < 2 || (K > 0 && (!n || !pos || !cov || !yaw || !length || !width || !pos_out || !cov_inv_out || !npred || !hull || !nhull)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_pack_predictions: bad argument");
    memset(pos_out, 0, sizeof(double) * 2 * (size_t)K * P);
    memset(cov_inv_out, 0, sizeof(double) * 4 * (size_t)K * P);
    memset(hull, 0, sizeof(double) * 6 * (size_t)K * (P - 1));
    for (int k = 0; k < K; k++) {
        npred[k] = n[k];
        nhull[k] = 0;
        const int m = std::min(n[k], P);
        if (m <= 0) continue;
        if (!pos[k] || !cov[k]) return set_err(FX_ERR_INVALID_ARGUMENT, "obstacle %d: NULL arrays", k);
        memcpy(pos_out + (size_t)2 * P * k, pos[k], sizeof(double) * 2 * m);
        int rc = fx_invert_cov2(m, cov[k], cov_inv_out + (size_t)4 * P * k);
        if (rc) return rc;
        if (yaw[k]) {
            rc = fx_build_obstacle_hulls(std::min(n_samples, m), pos_out + (size_t)2 * P * k, yaw[k], length[k], width[k],
                                         hull + (size_t)6 * (P - 1) * k, nhull + k);
            if (rc) return rc;
        }
    }
    return FX_OK;
}

// fx_build_obstacle_hulls for K obstacles in one call: pos [K][P][2], yaw [K][P], n_use [K] predictions that count,
// length / width [K]; hull [K][P-1][6], n_hull [K].
int32_t fx_build_obstacle_hulls_batch(int32_t K, int32_t P, const int32_t *n_use, const double *pos, const double *yaw,
                                      const double *length, const double *width, double *hull, int32_t *n_hull) {
    if (K < 0 || P < 2 || (K > 0 && (!n_use || !pos || !yaw || !length || !width || !hull || !n_hull)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_build_obstacle_hulls_batch: bad argument");
    for (int k = 0; k < K; k++) {
        if (n_use[k] > P) return set_err(FX_ERR_INVALID_ARGUMENT, "obstacle %d: %d predictions, stride %d", k, n_use[k], P);
        int rc = fx_build_obstacle_hulls(n_use[k], pos + (size_t)2 * P * k, yaw + (size_t)P * k, length[k], width[k],
                                         hull + (size_t)6 * (P - 1) * k, n_hull + k);
        if (rc) return rc;
    }
    return FX_OK;
}

int32_t fx_plan_step(FxContext *c, const FxProblem *prob, FxResult *res) {
    int rc = fx_upload(c, prob);
    if (rc) return rc;
    if ((rc = fx_evaluate(c))) return rc;
    return fx_finish(c, res);
}

// ---- read-back ----
static int check_agent(FxContext *c, int a) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (!c->evaluated) return set_err(FX_ERR_NOT_READY, "no evaluated plan step");
    if (a < 0 || a >= c->n_agents) return set_err(FX_ERR_INVALID_ARGUMENT, "agent %d out of range", a);
    return FX_OK;
}

int32_t fx_read_costs_agent(FxContext *c, int32_t agent, double *cost, uint32_t *flags) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    if (cost) HIP_TRY(hipMemcpy(cost, c->d_cost + s.cand_off, sizeof(double) * s.C, hipMemcpyDeviceToHost));
    if (flags) HIP_TRY(hipMemcpy(flags, c->d_flags + s.cand_off, sizeof(uint32_t) * s.C, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_costs(FxContext *c, double *cost, uint32_t *flags) { return fx_read_costs_agent(c, 0, cost, flags); }

int32_t fx_read_costmap_agent(FxContext *c, int32_t agent, double *raw) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_COSTMAP)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_COSTMAP");
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    HIP_TRY(hipMemcpy2D(raw, sizeof(double) * s.C, c->d_costmap + (size_t)FX_NUM_COSTS * s.cand_off, sizeof(double) * s.ld,
                        sizeof(double) * s.C, s.n_cost, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_costmap(FxContext *c, double *raw) { return fx_read_costmap_agent(c, 0, raw); }

static int32_t read_coeff_rows(FxContext *c, int32_t agent, int64_t index, double *lon6, double *lat6, double *tau_lat, int32_t *traj_len) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_BUNDLE)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (index < 0 || index >= s.C) return set_err(FX_ERR_INVALID_ARGUMENT, "candidate %lld out of range", (long long)index);
    double tmp[FX_COEFF_ROWS];
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    HIP_TRY(hipMemcpy2D(tmp, sizeof(double), c->d_coeffs + (size_t)FX_COEFF_ROWS * s.cand_off + index, sizeof(double) * s.ld,
                        sizeof(double), FX_COEFF_ROWS, hipMemcpyDeviceToHost));
    if (lon6) memcpy(lon6, tmp, 6 * sizeof(double));
    if (lat6) memcpy(lat6, tmp + 6, 6 * sizeof(double));
    if (tau_lat) *tau_lat = tmp[12];
    if (traj_len) HIP_TRY(hipMemcpy(traj_len, c->d_trajlen + s.cand_off + index, sizeof(int32_t), hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_coeffs_agent(FxContext *c, int32_t agent, int64_t index, double *lon6, double *lat6, int32_t *traj_len) {
    return read_coeff_rows(c, agent, index, lon6, lat6, nullptr, traj_len);
}
int32_t fx_read_lat_tau_agent(FxContext *c, int32_t agent, int64_t index, double *tau_lat) {
    if (!tau_lat) return set_err(FX_ERR_INVALID_ARGUMENT, "tau_lat is NULL");
    return read_coeff_rows(c, agent, index, nullptr, nullptr, tau_lat, nullptr);
}
int32_t fx_read_boundary_steps_agent(FxContext *c, int32_t agent, int32_t *steps) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    if (!steps) return set_err(FX_ERR_INVALID_ARGUMENT, "steps is NULL");
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_ROAD_BOUNDARY)) return set_err(FX_ERR_NOT_READY, "the step ran without FX_MODE_ROAD_BOUNDARY");
    HIP_TRY(hipMemcpyAsync(steps, c->d_bstep + s.cand_off, sizeof(int32_t) * s.C, hipMemcpyDeviceToHost, c->stream));
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    return FX_OK;
}
int32_t fx_read_boundary_steps(FxContext *c, int32_t *steps) { return fx_read_boundary_steps_agent(c, 0, steps); }

int32_t fx_read_coeffs(FxContext *c, int64_t index, double *lon6, double *lat6, int32_t *traj_len) {
    return fx_read_coeffs_agent(c, 0, index, lon6, lat6, traj_len);
}

int32_t fx_read_sample_agent(FxContext *c, int32_t agent, int64_t index, double *planes) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_BUNDLE)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (index < 0 || index >= s.C) return set_err(FX_ERR_INVALID_ARGUMENT, "candidate %lld out of range", (long long)index);
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    // one strided gather: 14*S elements, pitch = ld doubles
    HIP_TRY(hipMemcpy2D(planes, sizeof(double), c->h_probs[agent].planes + index, sizeof(double) * s.ld, sizeof(double),
                        (size_t)FX_NUM_PLANES * s.S, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_sample(FxContext *c, int64_t index, double *planes) { return fx_read_sample_agent(c, 0, index, planes); }

int32_t fx_read_candidate_agent(FxContext *c, int32_t agent, int64_t index, double *planes, double *coeffs13, int32_t *traj_len,
                                double *raw_costs, double *cost, uint32_t *flags) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (index < 0 || index >= s.C) return set_err(FX_ERR_INVALID_ARGUMENT, "candidate %lld out of range", (long long)index);
    const bool bundle = (s.mode & FX_MODE_WRITE_BUNDLE) != 0, cmap = (s.mode & FX_MODE_WRITE_COSTMAP) != 0;
    if ((planes || coeffs13 || traj_len) && !bundle) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (raw_costs && !cmap) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_COSTMAP");
    // all pieces go to one pinned block with asynchronous copies; ONE synchronisation
    double *hp = c->h_cand;
    const size_t n_pl = (size_t)FX_NUM_PLANES * s.S;
    double *h_co = hp + n_pl, *h_rc = h_co + FX_COEFF_ROWS, *h_c = h_rc + FX_NUM_COSTS;
    int32_t *h_tl = reinterpret_cast<int32_t *>(h_c + 1);
    uint32_t *h_fl = reinterpret_cast<uint32_t *>(h_c + 2);
    if (n_pl + FX_COEFF_ROWS + FX_NUM_COSTS + 4 > c->h_cand_doubles) return set_err(FX_ERR_CAPACITY, "candidate staging block too small");
    if (planes)
        HIP_TRY(hipMemcpy2DAsync(hp, sizeof(double), c->h_probs[agent].planes + index, sizeof(double) * s.ld, sizeof(double), n_pl,
                                 hipMemcpyDeviceToHost, c->stream));
    if (coeffs13)
        HIP_TRY(hipMemcpy2DAsync(h_co, sizeof(double), c->d_coeffs + (size_t)FX_COEFF_ROWS * s.cand_off + index, sizeof(double) * s.ld,
                                 sizeof(double), FX_COEFF_ROWS, hipMemcpyDeviceToHost, c->stream));
    if (traj_len) HIP_TRY(hipMemcpyAsync(h_tl, c->d_trajlen + s.cand_off + index, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (raw_costs && s.n_cost > 0)
        HIP_TRY(hipMemcpy2DAsync(h_rc, sizeof(double), c->d_costmap + (size_t)FX_NUM_COSTS * s.cand_off + index, sizeof(double) * s.ld,
                                 sizeof(double), s.n_cost, hipMemcpyDeviceToHost, c->stream));
    if (cost) HIP_TRY(hipMemcpyAsync(h_c, c->d_cost + s.cand_off + index, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (flags) HIP_TRY(hipMemcpyAsync(h_fl, c->d_flags + s.cand_off + index, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    if (planes) memcpy(planes, hp, sizeof(double) * n_pl);
    if (coeffs13) memcpy(coeffs13, h_co, sizeof(double) * FX_COEFF_ROWS);
    if (traj_len) *traj_len = *h_tl;
    if (raw_costs) memcpy(raw_costs, h_rc, sizeof(double) * s.n_cost);
    if (cost) *cost = *h_c;
    if (flags) *flags = *h_fl;
    return FX_OK;
}

int32_t fx_read_plane_agent(FxContext *c, int32_t agent, int32_t plane, double *out) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_BUNDLE)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (plane < 0 || plane >= FX_NUM_PLANES) return set_err(FX_ERR_INVALID_ARGUMENT, "plane %d out of range", plane);
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    HIP_TRY(hipMemcpy2D(out, sizeof(double) * s.C, c->h_probs[agent].planes + (size_t)plane * s.S * s.ld, sizeof(double) * s.ld,
                        sizeof(double) * s.C, s.S, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_plane(FxContext *c, int32_t plane, double *out) { return fx_read_plane_agent(c, 0, plane, out); }

int32_t fx_topk_to_device(FxContext *c, int32_t k, void *d_cost, void *d_index) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (!c->evaluated) return set_err(FX_ERR_NOT_READY, "no evaluated plan step");
    if (k < 1 || k > 64) return set_err(FX_ERR_INVALID_ARGUMENT, "k=%d outside [1,64]", k);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(fx_launch_topk(c->d_probs, c->n_agents, max_candidates_of(c), k, c->d_topk_scr_cost, c->d_topk_scr_idx, reinterpret_cast<double *>(d_cost),
                           reinterpret_cast<long long *>(d_index), c->stream));
    c->in_flight = true; c->tail_work = true;
    return FX_OK;
}

int32_t fx_read_topk_batch(FxContext *c, int32_t k, double *cost, int64_t *index) {
    int rc = fx_topk_to_device(c, k, c->d_topk_cost, c->d_topk_idx);
    if (rc) return rc;
    const size_t n = (size_t)k * c->n_agents;
    HIP_TRY(hipMemcpyAsync(c->h_topk_cost, c->d_topk_cost, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_topk_idx, c->d_topk_idx, sizeof(long long) * n, hipMemcpyDeviceToHost, c->stream));
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    memcpy(cost, c->h_topk_cost, sizeof(double) * n);
    for (size_t i = 0; i < n; i++) index[i] = (int64_t)c->h_topk_idx[i];
    return FX_OK;
}

int32_t fx_read_topk(FxContext *c, int32_t k, double *cost, int64_t *index, int32_t *n_out) {
    if (c && c->n_agents != 1) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_read_topk is single-agent; use fx_read_topk_batch");
    int rc = fx_read_topk_batch(c, k, cost, index);
    if (rc) return rc;
    int n = 0;
    while (n < k && index[n] >= 0) n++;
    if (n_out) *n_out = n;
    return FX_OK;
}

int32_t fx_build_obstacle_hulls(int32_t n_pred, const double *pos, const double *yaw, double length, double width,
                                double *hull, int32_t *n_hull) {
    if (!n_hull || (n_pred > 0 && (!pos || !yaw))) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_build_obstacle_hulls: NULL argument");
    // collision_check.py:165-168: an obstacle with <= 2 predicted steps is skipped
    if (n_pred <= 2) { *n_hull = 0; return FX_OK; }
    if (!hull) return set_err(FX_ERR_INVALID_ARGUMENT, "hull output is NULL");
    const double hl = length / 2, hw = width / 2;
    double u1x, u1y;   // box j + 1's heading is box j's of the next hull: one cos / sin per box
    ::sincos(yaw[0], &u1y, &u1x);
    for (int j = 0; j + 1 < n_pred; j++) {
        const double c0x = pos[2 * j], c0y = pos[2 * j + 1], c1x = pos[2 * j + 2], c1y = pos[2 * j + 3];
        const double u0x = u1x, u0y = u1y;
        ::sincos(yaw[j + 1], &u1y, &u1x);
        double mx = u0x + u1x, my = u0y + u1y;
        double mn = std::sqrt(mx * mx + my * my);
        double ex, ey;
        if (mn < 1e-12) { ex = u0x; ey = u0y; } else { ex = mx / mn; ey = my / mn; }
        const double fx = -ey, fy = ex;
        double lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
        for (int b = 0; b < 2; b++) {
            const double cx = b ? c1x : c0x, cy = b ? c1y : c0y, ux = b ? u1x : u0x, uy = b ? u1y : u0y;
            const double p1 = cx * ex + cy * ey, p2 = cx * fx + cy * fy;
            const double r1 = hl * std::fabs(ux * ex + uy * ey) + hw * std::fabs(-uy * ex + ux * ey);
            const double r2 = hl * std::fabs(ux * fx + uy * fy) + hw * std::fabs(-uy * fx + ux * fy);
            if (b == 0) { lo1 = p1 - r1; hi1 = p1 + r1; lo2 = p2 - r2; hi2 = p2 + r2; }
            else {
                lo1 = std::fmin(lo1, p1 - r1); hi1 = std::fmax(hi1, p1 + r1);
                lo2 = std::fmin(lo2, p2 - r2); hi2 = std::fmax(hi2, p2 + r2);
            }
        }
        const double m1 = 0.5 * (lo1 + hi1), m2 = 0.5 * (lo2 + hi2);
        double *o = hull + 6 * j;
        o[0] = m1 * ex + m2 * fx;
        o[1] = m1 * ey + m2 * fy;
        o[2] = ex;
        o[3] = ey;
        o[4] = 0.5 * (hi1 - lo1);
        o[5] = 0.5 * (hi2 - lo2);
    }
    *n_hull = n_pred - 1;
    return FX_OK;
}

// self-test hook: atan / sin / cos of the device math kernels for n host values (synchronous)
int32_t fx_math_selftest(int32_t n, const double *x, double *atan_out, double *sin_out, double *cos_out) {
    if (n < 1 || !x || !atan_out || !sin_out || !cos_out) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_math_selftest: bad argument");
    double *d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * 4 * n));
    HIP_TRY(hipMemcpy(d, x, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(fx_launch_math_test(n, d, d + n, d + 2 * n, d + 3 * n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(atan_out, d + n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sin_out, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cos_out, d + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return FX_OK;
}

// Road-boundary geometry prep (host only): split segments into pieces, bin them by reference knot.
int32_t fx_build_boundary_bins(int32_t M, const double *ref_x, const double *ref_y, int32_t n_seg, const double *seg,
                               double max_len, double reach, int32_t piece_cap, double *piece_out, int32_t *n_piece,
                               int32_t *bin_out, int32_t item_cap, int32_t *item_out, int32_t *n_item) {
    if (M < 1 || !ref_x || !ref_y || n_seg < 0 || (n_seg && !seg) || !(max_len > 0.0) || !(reach >= 0.0) || !n_piece || !n_item ||
        !bin_out)
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_build_boundary_bins: bad argument");
    int64_t np_ = 0;
    for (int i = 0; i < n_seg; i++) {
        const double *q = seg + 4 * (size_t)i;
        const double len = std::sqrt((q[2] - q[0]) * (q[2] - q[0]) + (q[3] - q[1]) * (q[3] - q[1]));
        const int n = std::max(1, (int)std::ceil(len / max_len));
        for (int k = 0; k < n; k++, np_++) {
            if (np_ >= piece_cap || !piece_out) continue;
            const double t0 = (double)k / n, t1 = (double)(k + 1) / n;
            const double ax = q[0] + t0 * (q[2] - q[0]), ay = q[1] + t0 * (q[3] - q[1]);
            const double bx = q[0] + t1 * (q[2] - q[0]), by = q[1] + t1 * (q[3] - q[1]);
            double *o = piece_out + 4 * (size_t)np_;
            o[0] = 0.5 * (ax + bx); o[1] = 0.5 * (ay + by); o[2] = 0.5 * (bx - ax); o[3] = 0.5 * (by - ay);
        }
    }
    *n_piece = (int32_t)np_;
    if (np_ > piece_cap || !piece_out) { *n_item = 0; return set_err(FX_ERR_CAPACITY, "boundary needs %lld pieces", (long long)np_); }
    int64_t ni = 0;
    bin_out[0] = 0;
    for (int k = 0; k < M; k++) {
        for (int64_t j = 0; j < np_; j++) {
            const double *o = piece_out + 4 * (size_t)j;
            const double dx = o[0] - ref_x[k], dy = o[1] - ref_y[k];
            if (std::sqrt(dx * dx + dy * dy) <= reach + std::sqrt(o[2] * o[2] + o[3] * o[3])) {
                if (ni < item_cap && item_out) item_out[ni] = (int32_t)j;
                ni++;
            }
        }
        bin_out[k + 1] = (int32_t)std::min<int64_t>(ni, INT32_MAX);
    }
    *n_item = (int32_t)std::min<int64_t>(ni, INT32_MAX);
    if (ni > item_cap || !item_out) return set_err(FX_ERR_CAPACITY, "boundary bins need %lld items", (long long)ni);
    return FX_OK;
}

int64_t fx_device_bytes(const FxContext *c) { return c ? c->dev_bytes : 0; }
double fx_last_kernel_ms(const FxContext *cc) {
    FxContext *c = const_cast<FxContext *>(cc);
    if (!c || !c->evaluated || c->n_timed == 0) return 0.0;
    FxContext::TimeSlot &t = c->ring[(c->n_timed - 1) % FxContext::kTimeRing];
    return fetch_slot(c, t) ? 0.0 : (double)t.step_ms;
}
double fx_last_eval_kernel_ms(const FxContext *cc) {
    FxContext *c = const_cast<FxContext *>(cc);
    if (!c || !c->evaluated || c->n_timed == 0) return 0.0;
    FxContext::TimeSlot &t = c->ring[(c->n_timed - 1) % FxContext::kTimeRing];
    return fetch_slot(c, t) ? 0.0 : (double)t.eval_ms;
}
int32_t fx_read_kernel_times(FxContext *c, int32_t max_n, double *eval_ms, double *step_ms, int32_t *n_out) {
    if (!c || max_n < 0 || !n_out) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_read_kernel_times: bad argument");
    const long long have = std::min<long long>(c->n_timed, FxContext::kTimeRing);
    const int n = (int)std::min<long long>(have, max_n);
    for (int k = 0; k < n; k++) {  // oldest of the returned window first
        FxContext::TimeSlot &t = c->ring[(c->n_timed - n + k) % FxContext::kTimeRing];
        int rc = fetch_slot(c, t);
        if (rc) return rc;
        if (eval_ms) eval_ms[k] = t.eval_ms;
        if (step_ms) step_ms[k] = t.step_ms;
    }
    *n_out = n;
    return FX_OK;
}
int32_t fx_set_timing_interval(FxContext *c, int32_t every) {
    if (!c || every < 1) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_set_timing_interval: every=%d", every);
    c->timing_every = every;
    return FX_OK;
}
// How the last evaluation was launched (tests and tools): out[0] grid kernel, [1] lanes per candidate, [2] waves per SIMD,
// [3] workgroup size, [4] wave split, [5] fused selection, [6] workgroups per agent (max), [7] agents, [8] winner package,
// [9] dynamic LDS bytes
int32_t fx_step_info(const FxContext *c, int64_t *out10) {
    if (!c || !out10) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_step_info: NULL argument");
    const int64_t v[10] = {c->use_grid, c->G_step, c->wpe_step, c->block_step, c->wsplit_step, c->fused_step, c->max_blocks_step,
                           c->n_agents, c->pkg_step, (int64_t)c->lds_step};
    memcpy(out10, v, sizeof(v));
    return FX_OK;
}

int32_t fx_set_obstacle_stage(FxContext *c, int32_t stage, int32_t steps_per_item) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (stage < 0 || stage > 2) return set_err(FX_ERR_INVALID_ARGUMENT, "stage must be 0 (auto), 1 (fused into the walk) or 2 (own kernel)");
    if (steps_per_item != 0 && steps_per_item != 2 && steps_per_item != 3 && steps_per_item != 5)
        return set_err(FX_ERR_INVALID_ARGUMENT, "steps_per_item must be 0 (auto), 2, 3 or 5");
    c->obst_force = stage; c->obst_CH = steps_per_item;
    return FX_OK;
}
double fx_last_obstacle_kernel_ms(const FxContext *cc) {
    FxContext *c = const_cast<FxContext *>(cc);
    if (!c || !c->evaluated || c->n_timed == 0) return 0.0;
    FxContext::TimeSlot &t = c->ring[(c->n_timed - 1) % FxContext::kTimeRing];
    return fetch_slot(c, t) ? 0.0 : (double)t.obst_ms;
}
int32_t fx_read_obstacle_kernel_times(FxContext *c, int32_t max_n, double *obst_ms, int32_t *n_out) {
    if (!c || max_n < 0 || !n_out) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_read_obstacle_kernel_times: bad argument");
    const long long have = std::min<long long>(c->n_timed, FxContext::kTimeRing);
    const int n = (int)std::min<long long>(have, max_n);
    for (int k = 0; k < n; k++) {  // oldest of the returned window first
        FxContext::TimeSlot &t = c->ring[(c->n_timed - n + k) % FxContext::kTimeRing];
        int rc = fetch_slot(c, t);
        if (rc) return rc;
        if (obst_ms) obst_ms[k] = t.obst_ms;
    }
    *n_out = n;
    return FX_OK;
}
// fx_step_info, extended: out[0 .. 9] as fx_step_info, [10] obstacle stage as its own kernel, [11] steps per work item, [12] work
// items (waves) per agent (max) and [13] dynamic LDS bytes of that kernel, [14] waves per workgroup when a tile's chunks share one
// workgroup (0: one wave per (tile, chunk) item), [15] what the agent's last workgroup did beyond the arg-min (1: collision count,
// 2: winner package; fx_tail.h)
int32_t fx_step_info_ex(const FxContext *c, int64_t *out16) {
    if (!c || !out16) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_step_info_ex: NULL argument");
    int rc = fx_step_info(c, out16);
    if (rc) return rc;
    out16[10] = c->split_step; out16[11] = c->split_CH; out16[12] = c->obs_blocks_step; out16[13] = (int64_t)c->obs_lds_step;
    out16[14] = c->obs_wg_step; out16[15] = c->tail_step | ((int64_t)c->stage_path << 8);
    if (c->step_kernel_step) {   // the whole step in one launch: steps per item, waves, LDS of THAT kernel; bit 16 says so
        out16[11] = c->step_CH; out16[12] = (int64_t)c->step_blocks * (FX_BLOCK / 64); out16[13] = (int64_t)c->step_lds; out16[14] = 0;
        out16[15] |= 1 << 16;
    }
    return FX_OK;
}

int32_t fx_set_step_kernel(FxContext *c, int32_t mode, int32_t steps_per_item) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (mode < 0 || mode > 2) return set_err(FX_ERR_INVALID_ARGUMENT, "mode must be 0 (auto), 1 (off: three launches) or 2 (on where applicable)");
    if (steps_per_item != 0 && steps_per_item != 3 && steps_per_item != 5 && steps_per_item != 8)
        return set_err(FX_ERR_INVALID_ARGUMENT, "steps_per_item must be 0 (auto), 3, 5 or 8");
    c->step_kernel_force = mode; c->step_kernel_CH = steps_per_item;   // (the mode takes effect at the next upload)
    return FX_OK;
}
int32_t fx_set_fused_selection(FxContext *c, int32_t enabled) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    c->fuse_enabled = enabled != 0;
    c->fuse_any_size = enabled == 2;   // (takes effect at the next upload)
    return FX_OK;
}
int32_t fx_set_timing(FxContext *c, int32_t mode) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (mode < FX_TIMING_OFF || mode > FX_TIMING_KERNEL) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_set_timing: unknown mode");
    c->timing = mode;
    return FX_OK;
}

// device pointers of agent 0's outputs, for callers that keep working on the GPU (torch tensors via
// from_blob-style wrapping or RCCL sends): cost f64[ld], flags u32[ld], planes f64[14][S][ld]
int32_t fx_device_views(FxContext *c, int32_t agent, void **cost, void **flags, void **planes, int64_t *ld) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    if (cost) *cost = c->h_probs[agent].cost;
    if (flags) *flags = c->h_probs[agent].flags;
    if (planes) *planes = (c->slots[agent].mode & FX_MODE_WRITE_BUNDLE) ? c->h_probs[agent].planes : nullptr;
    if (ld) *ld = c->slots[agent].ld;
    return FX_OK;
}

}  // extern "C"
