// fx_api_step.hip -- the plan step through the C-ABI: upload (work decomposition, launch policy), fx_evaluate, results, state
// updates, the winner package and the batched plan calls (header: include/fxplan.h; context: fx_context.h).
#include "fx_context.h"

extern "C" {

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
        // ... but with the obstacle stage in the walk as well (the north star as written) the kernel is bound by the vector unit and
        // by latency as much as by its stores: the third wave per SIMD pays (168 registers, no vector spill): 1 062 -> 1 004 us
        // same-box, tools/ns_wpe.py
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
            const size_t lds = std::max(c->lds_step, (size_t)(FX_BLOCK / 64) * sizeof(double) * 6 * (size_t)CH * (size_t)c->K_max_step);   // FX_STEP_ITEM_DOUBLES
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
            const int64_t items = (int64_t)tiles_est * NC;   // (one tile x one chunk per wave)
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

int32_t fx_plan_step(FxContext *c, const FxProblem *prob, FxResult *res) {
    int rc = fx_upload(c, prob);
    if (rc) return rc;
    if ((rc = fx_evaluate(c))) return rc;
    return fx_finish(c, res);
}

}  // extern "C"
