// fx_api.hip -- host side of libfxplan.so: the C-ABI declared in include/fxplan.h.
//
// A context owns (a) one pinned host staging buffer + one device arena for all per-step inputs, so a plan
// step is ONE H2D copy, (b) the per-candidate outputs (cost, flags, cost map, SoA bundle), (c) a pinned
// read-back block for the counters/winner, so a plan step is ONE small D2H copy.  All work is enqueued on the
// context's HIP stream; nothing synchronises until fx_finish().
#include "fx_context.h"

thread_local char g_err[512] = "";

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
