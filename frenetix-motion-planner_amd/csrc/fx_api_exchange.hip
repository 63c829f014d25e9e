// fx_api_exchange.hip -- the survivor exchange inside the library: RCCL bound at run time, one all-gather per step (header:
// include/fxplan.h; context: fx_context.h).
#include "fx_context.h"

extern "C" {

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

}  // extern "C"
