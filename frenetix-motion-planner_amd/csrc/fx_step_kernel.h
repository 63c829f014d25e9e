// fx_step_kernel.h -- a whole plan step of a mid-sized grid in ONE launch: walk | grid barrier | obstacle stage | grid barrier |
// selection.  OPT-IN (fx_set_step_kernel(ctx, 2, ..) / FX_STEP_KERNEL=1): measured slower than the three launches it replaces.
//
// Grids of 200 ... 3 072 waves (BASELINE config 3: 50 388 candidates, config 4's batch) run the obstacle stage with its own
// occupancy (fx_obstacle_kernel.h) and end with the sliced selection (fx_select.h): three launches.  Measured on the MI355X
// (tools/c3_streams.py): the step of config 3 takes 84 us, HALF of its candidates 65 us -- 46 us of the step are fixed costs, paid
// per launch: dispatch ramp and drain, the dependency gap in front of the next kernel, and every kernel's own chain of entry
// round trips (problem -> tables -> rows ...).  Here the three kernels are three PHASES of one launch:
//
//   phase 1  the walk (fx_eval_grid_kernel.h's body, MEGA: everything a later phase reads is stored write-through) on the first
//            `walk_blocks` workgroups; the launch's other workgroups ("helpers") go straight to the barrier;
//   barrier  every wave's stores acknowledged, one arrival per workgroup on a device counter, 64 release flags (grid_barrier);
//   phase 2  the obstacle stage: (T tiles of 64 listed candidates, chunk of CH steps) items dealt to ALL waves of the launch
//            (obstacle_items: the chunks of a tile meet through global memory, agent-scope loads of what phase 1 left); CH is
//            picked so that the items fit the launch's waves in one round where possible;
//   barrier
//   phase 3  the selection (fx_select_body<true>): every workgroup reduces the tiles' partials to the winner, counts the colliding
//            candidates in front of it in ITS slice of the candidates and takes a ticket; the last one publishes (and gathers the
//            winner package).
//
// Results are those of the three launches bit for bit at three steps per item (the phases are the kernels' bodies and expression
// trees; tests/test_step_kernel.py).  Measured (tools/c3_step_kernel.py, tools/probe_step_kernel.py; profiles/r6/NOTES.md): walk
// done at 38 us, each barrier ~1 us, selection 5 - 9 us -- and the obstacle phase 40 - 45 us against the obstacle kernel's 32:
// config 3 takes 94 us this way, 86 us as three launches.  Why: the launch runs with the walk's 165 registers -- three waves per SIMD,
// 3 072 waves for 5 030 three-step items --, so every wave runs TWO items' latency chains back to back (entry round trips, visits with
// their LDS waits, hand-off, closing), where the obstacle kernel's 73 registers give every item its own wave at six per SIMD and
// overlap those latencies across waves.  (The vector unit itself is not the limit: two waves per SIMD already reach its FP64 rate,
// tools/micro/clockrate.hip.)  What the fixed costs return (two launches and gaps, ~12 us) the obstacle phase loses twice over; it
// wins at no grid size (profiles/r6/NOTES.md).  The kernel stays as the measured alternative.
//
// The grid barrier needs every workgroup of the launch resident at once: the host launches at most what the occupancy query says
// the device holds (fx_api.hip) and otherwise keeps the three launches.  A barrier that does not complete within FX_BAR_TIMEOUT
// (another process holding the CUs for seconds) gives up -- the launch returns without publishing and the host's own wait times
// out (FX_ERR_TIMEOUT) instead of a hung GPU.
#pragma once

#include "fx_eval_grid_kernel.h"
#include "fx_obstacle_kernel.h"
#include "fx_select.h"

#define FX_BAR_TIMEOUT_TICKS 200000000ULL   // 2 s of the 100 MHz wall clock

// (StepArgs: fx_device.h)

namespace fxk {

// Grid barrier.  Layout of one barrier's block (FX_BAR_WORDS 64-bit words, device memory, never reset): word 0 the arrival counter,
// words 16 * (1 + k), k = 0 .. 63, the release flags -- 64 copies on 64 cache lines.  A workgroup arrives with ONE agent-scope
// atomic; the one that completes the count writes the launch's target value into all 64 flags (one store instruction), everybody
// else polls flag (workgroup % 64).  (First version: every waiting workgroup polled the counter itself -- up to 768 pollers on ONE
// line, four requests per microsecond each: the walk's slowest workgroups took 63 us instead of 36 and the obstacle items
// 20 - 100 us, tools/probe_step_kernel.py -- the channel that holds the line served nothing else.)  Values only grow (the host
// advances `target` by the launch's workgroups per step), so nothing is ever reset.
// Returns false when the barrier timed out (the caller leaves the kernel).
#define FX_BAR_WORDS (16 * 65)
__device__ __forceinline__ bool grid_barrier(unsigned long long *bar, const unsigned long long target) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores are performed
    __syncthreads();
    if (threadIdx.x < 64) {
        FX_GLOBAL unsigned long long *b = as_global(bar);
        unsigned long long seen = 0ULL;
        if (threadIdx.x == 0) seen = __hip_atomic_fetch_add(b, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ULL;
        seen = uniform_u64(seen);
        int good = 1;
        if (seen == target) {
            st_agent(b + 16 * (1 + threadIdx.x), target);   // the last arrival releases: 64 flags, one instruction
        } else if (threadIdx.x == 0) {
            FX_GLOBAL unsigned long long *f = b + 16 * (1 + (blockIdx.x + blockIdx.y * gridDim.x) % 64);
            if (ld_agent(f) < target) {
                const unsigned long long t0 = wall_clock64();
                while (ld_agent(f) < target) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t0 > FX_BAR_TIMEOUT_TICKS) { good = 0; break; }
                }
            }
        }
        if (threadIdx.x == 0) ok = good;
    }
    __syncthreads();
    return ok != 0;
}

}  // namespace fxk

namespace fxk {

// the 64-bit value lane `l` holds (l a compile-time constant after unrolling)
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffULL), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// Dynamic LDS of one wave's obstacle item: per (step of the chunk, obstacle) 16 doubles -- (cu, cw) | hull circle (hx2, hy2, hr2,
// ck) | (l11, l12 | l22, -) | obstacle hull of the exact test (6)
#define FX_STEP_ITEM_DOUBLES(CH, K) ((size_t)16 * (size_t)(CH) * (size_t)(K))

// One obstacle item of the one-launch step: T tiles of 64 listed candidates x one chunk of CH steps, by one wave.  The arithmetic per
// candidate is fx_obstacle_kernel.h's (same expression trees, same order: the results agree bit for bit with that kernel's
// single-wave items); what differs is where the operands come from and how often they are used:
//   * this phase runs with the walk's register allocation -- three waves per SIMD, not the obstacle kernel's six --, so a
//     dependent scalar load per pair of visits is not hidden by other waves (measured 4.2 us per STEP of one item,
//     tools/probe_step_kernel.py): EVERY operand of the visit loops -- multipliers, addends, hull circles, the obstacle hulls of the
//     exact test, the chunk's step masks -- is requested at the item's entry, in one round trip, and parked in the wave's LDS block;
//   * a wave-uniform operand read from LDS costs the LDS as much as a per-lane one (a wave's ds_read_b128 moves 1 KB): 24 cycles of the
//     CU's LDS per visit against ~41 of one SIMD's vector unit, i.e. LDS-bound with the CU's four SIMDs busy (measured 2.1 us per
//     step and item).  So one wave walks T tiles: every operand it reads is used for T candidates per lane.
// What the walk left (list, flags, rows, cost sums) is read with agent-scope loads -- other workgroups of the launch that is still
// running wrote it (write-through, acknowledged before the grid barrier); what the selection phase reads is stored the same way.
template <int CH, int T>
__device__ __forceinline__ void obstacle_items(const DevProblem &P, const int tile0, const int chunk, const long long n_live,
                                               double *__restrict__ lds_dyn) {
    const uint32_t mode = P.mode;
    const int S = P.S, K = P.K;
    const int NC = (S - 1 + CH - 1) / CH;
    const int64_t C = P.C, ld = P.ld;
    const int lane = threadIdx.x & 63;
    const int i_a = 1 + chunk * CH, i_b = min(S, i_a + CH);
    const int64_t ps = (int64_t)S * ld;
    const FX_GLOBAL double *__restrict__ pl = as_global(P.planes);
    const bool col_mode = (mode & FX_MODE_COLLISION) != 0 && K > 0;
    FX_OSTAMP(0);
    // ---- everything the item needs, requested together ----
    // tables: entry e = (step of the chunk, obstacle) is handled by lane e, e + 64 (two in flight), beyond that in a loop
    const FX_GLOBAL double *__restrict__ hot_a = as_global(P.obs_hot) + (size_t)i_a * K * FX_HOT_STRIDE;
    const FX_GLOBAL double *__restrict__ rec_a = as_global(P.obs_rec) + (size_t)i_a * K * 12;
    fx_d2 *__restrict__ cc_tab = reinterpret_cast<fx_d2 *>(lds_dyn);                            // [CH][K]
    double *__restrict__ circ_tab = lds_dyn + 2 * (size_t)CH * K;                               // [CH][K][4]
    fx_d2 *__restrict__ mul_tab = reinterpret_cast<fx_d2 *>(lds_dyn + 6 * (size_t)CH * K);      // [CH][K][2]
    double *__restrict__ hull_tab = lds_dyn + 10 * (size_t)CH * K;                              // [CH][K][6]
    const int n_e = max((i_b - i_a) * K, 0);
    auto load_entry = [&](int e, double *v) {
        const FX_GLOBAL double *q = hot_a + (size_t)e * FX_HOT_STRIDE;
        const FX_GLOBAL double *r = rec_a + (size_t)e * 12 + 6;
        v[0] = q[FX_HOT_CU]; v[1] = q[FX_HOT_CW];
        v[2] = q[FX_HOT_HX2]; v[3] = q[FX_HOT_HY2]; v[4] = q[FX_HOT_HR2]; v[5] = q[FX_HOT_CK];
        v[6] = q[FX_HOT_L11]; v[7] = q[FX_HOT_L12]; v[8] = q[FX_HOT_L22];
#pragma unroll
        for (int w = 0; w < 6; w++) v[9 + w] = r[w];
    };
    auto park_entry = [&](int e, const double *v) {
        cc_tab[e] = fx_d2{v[0], v[1]};
        circ_tab[4 * e + 0] = v[2]; circ_tab[4 * e + 1] = v[3]; circ_tab[4 * e + 2] = v[4]; circ_tab[4 * e + 3] = v[5];
        mul_tab[2 * e] = fx_d2{v[6], v[7]}; mul_tab[2 * e + 1] = fx_d2{v[8], 0.0};
#pragma unroll
        for (int w = 0; w < 6; w++) hull_tab[6 * e + w] = v[9 + w];
    };
    double hv[2][15];
#pragma unroll
    for (int u = 0; u < 2; u++) load_entry(max(min(lane + 64 * u, n_e - 1), 0), hv[u]);   // (K > 0: the host defers only agents with obstacles)
    // the chunk's step masks: lane j holds step i_a + j's
    unsigned long long pm_lane, hm_lane;
    {
        const int il = min(i_a + lane, S - 1);
        pm_lane = as_global(P.obs_pmask)[il];
        hm_lane = as_global(P.obs_hmask)[il];
    }
    // the tiles' candidates (list entries of the walk), their flag words, rows i_a - 1 .. i_b - 1 of x, y (theta with the collision
    // stage) and the two operands of the cost sum's continuation (the wave that closes a tile needs them)
    int64_t g[T];
    uint32_t f[T];
    bool act[T];
    double xs[T][CH + 1], ys[T][CH + 1], ts[T][CH + 1], pre_cost[T], pre_tail[T];
    int32_t g_listed[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        const int64_t slot = min((int64_t)(tile0 + t) * 64 + lane, C - 1);   // (past the list's end: some candidate, loaded from, never stored for)
        g_listed[t] = ld_agent(as_global(P.obs_list) + slot);
    }
#pragma unroll
    for (int t = 0; t < T; t++) {
        g[t] = (uint32_t)g_listed[t] < (uint64_t)C ? (int64_t)g_listed[t] : 0;
        act[t] = (int64_t)(tile0 + t) * 64 + lane < n_live;
        f[t] = ld_agent(as_global(P.flags) + g[t]);
#pragma unroll
        for (int j = 0; j <= CH; j++) {
            const int i = min(i_a - 1 + j, S - 1);
            xs[t][j] = ld_agent(pl + ((int64_t)FX_PL_X * ps + (int64_t)i * ld + g[t]));
            ys[t][j] = ld_agent(pl + ((int64_t)FX_PL_Y * ps + (int64_t)i * ld + g[t]));
            ts[t][j] = col_mode ? ld_agent(pl + ((int64_t)FX_PL_THETA * ps + (int64_t)i * ld + g[t])) : 0.0;
        }
        pre_cost[t] = ld_agent(as_global(P.cost) + g[t]);
        pre_tail[t] = ld_agent(as_global(P.cost_tail) + g[t]);
    }
    // ---- tables -> LDS ----
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (lane + 64 * u < n_e) park_entry(lane + 64 * u, hv[u]);
    for (int e0 = lane + 128; e0 < n_e; e0 += 64) {
        double v[15];
        load_entry(e0, v);
        park_entry(e0, v);
    }
    // where the prediction term sits in the (id-sorted) cost function
    int n_pred = -1;
    double w_pred = 0.0;
    bool has_tail = false;
    for (int n = 0; n < P.n_cost; n++)
        if (P.cost_id[n] == FX_COST_PREDICTION) { n_pred = n; w_pred = P.cost_w[n]; has_tail = n + 1 < P.n_cost; }
    // every chunk of a tile looks at the same 64 flag words: these are tile-uniform
    bool do_pred[T], do_col[T], work[T];
    bool any_pred = false, any_col = false;
#pragma unroll
    for (int t = 0; t < T; t++) {
        do_pred[t] = n_pred >= 0 && K > 0 && wave_any_bit(act[t] && (f[t] & FX_FLAG_COSTED));
        do_col[t] = col_mode && wave_any_bit(act[t] && (f[t] & FX_FLAG_SELECTABLE));
        work[t] = do_pred[t] || do_col[t];
        any_pred |= do_pred[t]; any_col |= do_col[t];
    }
    FX_OSTAMP(1);
    double acc[T];
    bool collided[T];
#pragma unroll
    for (int t = 0; t < T; t++) { acc[t] = 0.0; collided[t] = false; }
    if ((any_pred || any_col) && chunk < NC) {
        const FX_GLOBAL double *__restrict__ rec = as_global(P.obs_rec);
        const double ox = P.hot_origin[0], oy = P.hot_origin[1];
        const double wb = P.veh.wb_rear_axle, half_len = P.veh.length / 2, half_wid = P.veh.width / 2;
        const double gap_margin = P.hot_gap_margin;
        // constant part of a hull's bounding radius for the wave-level cull: |wb| + sqrt(2) (|wb| + hd), rounded up
        const float cull_r0 = (float)((fabs(wb) * 2.41422 + 1.41423 * sqrt(half_len * half_len + half_wid * half_wid)) * 1.00001);
        const unsigned long long full = K >= 64 ? ~0ULL : ((1ULL << K) - 1ULL);
        const int kl = min(lane, K - 1);
        // the table writes above are this wave's own: LDS operations of one wave execute in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double sn_prev[T], cs_prev[T];
        bool have_prev[T];   // wave-uniform: (sn_prev, cs_prev) = sin / cos of theta at step i - 1
#pragma unroll
        for (int t = 0; t < T; t++) { sn_prev[t] = 0.0; cs_prev[t] = 1.0; have_prev[t] = false; }
        FX_OSTAMP(2);
#pragma unroll
        for (int j = 1; j <= CH; j++) {
            const int i = i_a + j - 1;
            if (i < i_b) {
                const unsigned long long pm = any_pred ? lane_u64(pm_lane, j - 1) : 0ULL;
                const unsigned long long hm_step = (any_col && i >= 2) ? lane_u64(hm_lane, j - 1) : 0ULL;
                const fx_d2 *__restrict__ cc_i = cc_tab + (size_t)(j - 1) * K;
                const double *__restrict__ circ_i = circ_tab + 4 * (size_t)(j - 1) * K;
                const fx_d2 *__restrict__ mul_i = mul_tab + 2 * (size_t)(j - 1) * K;
                const double *__restrict__ hull_i = hull_tab + 6 * (size_t)(j - 1) * K;
                // ---- prediction cost: sum over the obstacles of 1 / m^2 (collision_probability.py:283-292), T tiles per operand ----
                if (pm) {
                    double xr[T], yr[T], ssum[T];
#pragma unroll
                    for (int t = 0; t < T; t++) { xr[t] = xs[t][j] - ox; yr[t] = ys[t][j] - oy; ssum[t] = 0.0; }
                    auto ld_ = [&](int k) {
                        ObsEntry e;
                        const fx_d2 a = mul_i[2 * k], b = mul_i[2 * k + 1];
                        e.l11 = a.x; e.l12 = a.y; e.l22 = b.x;
                        e.cc = cc_i[k];
                        return e;
                    };
                    if (pm == full) {
                        // four obstacles per iteration, every complete group of four shares one reciprocal (fx_obstacle_kernel.h / fx_walk.h:
                        // same grouping, same expression tree)
                        const int kz = K - 1;
                        ObsEntry a = ld_(0), b = ld_(min(1, kz)), c, d;
                        int k = 0;
                        for (; k + 3 < K; k += 4) {
                            c = ld_(k + 2); d = ld_(k + 3);
                            double qa[T], qb[T];
#pragma unroll
                            for (int t = 0; t < T; t++) { qa[t] = obs_msq(a, xr[t], yr[t]); qb[t] = obs_msq(b, xr[t], yr[t]); }
                            a = ld_(min(k + 4, kz)); b = ld_(min(k + 5, kz));
#pragma unroll
                            for (int t = 0; t < T; t++) ssum[t] += obs_four(qa[t], qb[t], obs_msq(c, xr[t], yr[t]), obs_msq(d, xr[t], yr[t]));
                        }
                        // entries k (a) and k + 1 (b) are loaded where they exist; up to three remain
                        ObsEntry e2 = a;
                        if (k + 2 < K) e2 = ld_(k + 2);
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            double s1 = 0.0;
                            if (k < K) ssum[t] += rcp_pred(obs_msq(a, xr[t], yr[t]));
                            if (k + 1 < K) s1 += rcp_pred(obs_msq(b, xr[t], yr[t]));
                            if (k + 2 < K) ssum[t] += rcp_pred(obs_msq(e2, xr[t], yr[t]));
                            ssum[t] += s1;
                        }
                    } else {
                        unsigned long long m = pm;
                        while (m) {
                            const int k = __builtin_ctzll(m);
                            m &= m - 1;
                            const ObsEntry e = ld_(k);
#pragma unroll
                            for (int t = 0; t < T; t++) ssum[t] += rcp_pred(obs_msq(e, xr[t], yr[t]));
                        }
                    }
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        // anything not finite (an obstacle centre hit to the last bit, a covariance without a Cholesky factor: the
                        // host leaves a zero entry) sends the step to the reference form on the raw records -- rare
                        if (wave_any_bit(!(ssum[t] < 1e300))) {
                            const auto rec_i = rec + (int64_t)i * K * 12;
                            double a = 0.0;
                            unsigned long long m = pm;
                            while (m) {
                                const int k = __builtin_ctzll(m);
                                m &= m - 1;
                                const auto q = rec_i + k * 12;
                                const double e0 = xs[t][j] - q[0], e1 = ys[t][j] - q[1];
                                const double r0 = fma(e1, q[4], e0 * q[2]), r1 = fma(e1, q[5], e0 * q[3]);
                                const double mq = fma(r1, e1, r0 * e0);
                                const double mm = mq * mq;
                                a += mm > 0.0 ? rcp_pred(mm) : 1.0 / mm;
                            }
                            ssum[t] = !(ssum[t] < 1e300) ? a : ssum[t];
                        }
                        acc[t] += ssum[t];
                    }
                }
                if (j == 1) FX_OSTAMP(12);
                // ---- collision: OBB-sum hull of ego boxes (i-1, i) against the obstacle hulls of this step, tile by tile ----
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const unsigned long long hm = do_col[t] ? hm_step : 0ULL;
                    if (hm) {
                        // wave-level cull before anything of the boxes is built (fx_obstacle_kernel.h: the same bound)
                        unsigned long long cand;
                        {
                            const double mxr = fma(0.5, xs[t][j - 1] + xs[t][j], -ox), myr = fma(0.5, ys[t][j - 1] + ys[t][j], -oy);
                            const double tx = xs[t][j] - xs[t][j - 1], ty = ys[t][j] - ys[t][j - 1];
                            const double qd = fma(tx, tx, ty * ty);
                            const double m0x = uniform_f64(mxr), m0y = uniform_f64(myr);
                            const double dx = mxr - m0x, dy = myr - m0y;
                            const double qe = fma(dx, dx, dy * dy);
                            const bool bad = !(qd + qe < 1e300);   // a point that is not finite keeps every obstacle on the exact path
                            const float reach = fmaf(__builtin_amdgcn_sqrtf((float)qe), 1.00001f,
                                                     fmaf(__builtin_amdgcn_sqrtf((float)qd), 0.707115f, cull_r0));
                            const unsigned rb = wave_max_u32(bad ? 0u : __float_as_uint(reach));  // reach >= 0: bit patterns order like values
                            const double Rw = (double)__uint_as_float(rb);
                            const double *qo = circ_i + 4 * (size_t)kl;
                            const double ex = fma(-0.5, qo[0], -m0x), ey = fma(-0.5, qo[1], -m0y);   // h - m0
                            const double rr = fma(-0.5, qo[2], Rw) * 1.00001;                        // r_o + R
                            cand = __builtin_amdgcn_ballot_w64(!(fma(ex, ex, ey * ey) > rr * rr)) & hm;
                            if (wave_any_bit(bad)) cand = hm;
                        }
                        have_prev[t] = have_prev[t] && cand != 0ULL;
                        if (cand) {
                            // the two ego boxes (rear axle + wb along the heading, state.py:30-39) and their OBB-sum hull; the heading
                            // of step i - 1 is carried when that step built a hull too
                            double s0 = sn_prev[t], c0 = cs_prev[t], s1, c1;
                            if (!have_prev[t]) fxm::sincos(ts[t][j - 1], &s0, &c0);   // (wave-uniform: the previous step went the same way)
                            fxm::sincos(ts[t][j], &s1, &c1);
                            sn_prev[t] = s1; cs_prev[t] = c1;
                            const Obb hull = obb_hull(fma(wb, c0, xs[t][j - 1]), fma(wb, s0, ys[t][j - 1]), c0, s0, fma(wb, c1, xs[t][j]),
                                                      fma(wb, s1, ys[t][j]), c1, s1, half_len, half_wid);
                            // per-lane circle test in the expanded form of the table for the survivors, the exact 4-axis test for
                            // what is still near (the walk's sequence: decisions are those of the brute-force definition)
                            const double re = (hull.h1 + hull.h2) * 1.000001;
                            const double cxr = hull.cx - ox, cyr = hull.cy - oy;
                            const double wq = fma(cxr, cxr, fma(cyr, cyr, -re * re));
                            do {
                                const int k = __builtin_ctzll(cand);
                                cand &= cand - 1;
                                const double *q = circ_i + 4 * (size_t)k;
                                const double gq = fma(q[0], cxr, fma(q[1], cyr, fma(q[2], re, q[3] + wq)));
                                if (wave_any_bit(!(gq > gap_margin))) collided[t] |= obb_overlap(hull, hull_i + 6 * k);
                            } while (cand);
                            have_prev[t] = true;
                        }
                    } else {
                        have_prev[t] = false;
                    }
                }
                if (j == 1) FX_OSTAMP(13);
                if (j == 2) FX_OSTAMP(14);
            }
        }
    }
    FX_OSTAMP(4);
    // ---- hand-off: agent-scope stores, acknowledged (vmcnt 0) before the tickets are taken -- whoever draws a tile's last ticket sees
    //      every chunk's partial of that tile ----
    FX_GLOBAL double *__restrict__ part = as_global(P.obs_part);
    FX_GLOBAL unsigned long long *__restrict__ colm = as_global(P.obs_colm);
    FX_GLOBAL unsigned int *__restrict__ ticket = as_global(P.obs_ticket);
    const int64_t n_tiles = (C + 63) / 64;
    bool exists[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        exists[t] = (int64_t)(tile0 + t) * 64 < n_live;
        if (exists[t] && work[t]) {
            if (act[t]) st_agent(part + ((int64_t)chunk * ld + g[t]), acc[t]);
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(collided[t]);
            if (lane == 0) st_agent(colm + ((int64_t)chunk * n_tiles + (tile0 + t)), cm);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int drew = 0;   // lane t draws tile t's ticket: one instruction for the T tiles
    if (lane < T && (int64_t)(tile0 + lane) * 64 < n_live) {
        bool w = false;
#pragma unroll
        for (int t = 0; t < T; t++) w |= (lane == t) && work[t];
        if (w) drew = __hip_atomic_fetch_add(ticket + (tile0 + lane), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    FX_OSTAMP(5);
    // ---- close the tiles whose last ticket this wave drew (a tile without work is closed by its chunk 0 alone) ----
#pragma unroll
    for (int t = 0; t < T; t++) {
        if (!exists[t]) continue;
        const unsigned int tk = (unsigned int)__builtin_amdgcn_readlane((int)drew, t);
        if (work[t] ? tk != (unsigned)(NC - 1) : chunk != 0) continue;
        const int tile = tile0 + t;
        if (work[t] && lane == 0) st_agent(ticket + tile, 0u);
        double pred = 0.0;
        unsigned long long cmask = 0ULL;
        if (work[t]) {
            for (int q = 0; q < NC; q++) {   // chunk order: deterministic, the same sum as fx_obstacle_kernel's
                pred += ld_agent(part + ((int64_t)q * ld + g[t]));
                cmask |= ld_agent(colm + ((int64_t)q * n_tiles + tile));
            }
        }
        uint32_t fl = f[t];
        const bool costed = (fl & FX_FLAG_COSTED) != 0, selectable = (fl & FX_FLAG_SELECTABLE) != 0;
        double total = pre_cost[t];
        if (n_pred >= 0) {
            // the walk left the running sum in front of the prediction term: continue the same sequence of additions
            double sum = total;
            sum += w_pred * pred;
            if (has_tail) sum += pre_tail[t];
            total = 0.0 + sum;
            if (act[t]) {
                st_agent(as_global(P.cost) + g[t], costed ? total : 0.0);
                if (mode & FX_MODE_WRITE_COSTMAP) st_agent(as_global(P.costmap) + ((int64_t)n_pred * ld + g[t]), costed ? pred : 0.0);
            }
        }
        if (selectable && (mode & FX_MODE_COLLISION) && ((cmask >> lane) & 1ULL)) {
            fl |= FX_FLAG_COLLISION;
            if (act[t]) st_agent(as_global(P.flags) + g[t], fl);
        }
        // (cost, index) arg-min of the tile: the lanes hold the list's candidates in no particular order, so among the lanes with
        // the minimum cost the smallest index is reduced as well
        const bool eligible = act[t] && selectable && !(fl & (FX_FLAG_COLLISION | FX_FLAG_BOUNDARY)) && total == total;
        const double bc = eligible ? total : INFINITY;
        double m = bc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off));
        long long bi = (eligible && bc == m) ? (long long)(g[t] + P.g_base) : 0x7fffffffffffffffLL;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const long long o = __shfl_xor(bi, off); bi = o < bi ? o : bi; }
        if (lane == 0) {
            st_agent(as_global(P.part_cost) + tile, m);
            st_agent(as_global(P.part_idx) + tile, (int64_t)bi);
        }
    }
    FX_OSTAMP(15);
}

}  // namespace fxk

template <int CH>
__global__ __launch_bounds__(FX_BLOCK, 3) void fx_step_kernel(const DevProblem *__restrict__ probs, const FuseArgs fuse, const StepArgs sa) {
    using namespace fxk;
    extern __shared__ __attribute__((aligned(16))) double lds_step[];
    // ---- phase 1: the walk (its early returns land here) ----
    FX_MSTAMP(6);
    if ((int)blockIdx.x < sa.walk_blocks) fx_eval_grid_body<2, true, false, 2, true, true>(probs, fuse);
    const unsigned long long n_wg = (unsigned long long)gridDim.x * gridDim.y;
    FX_MSTAMP(7);
    if (!grid_barrier(sa.bar, sa.bar_base + n_wg)) return;
    FX_MSTAMP(8);

    // ---- phase 2: obstacle stage over the list the walk left ----
    const DevProblem &P = probs[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_wave = (int)(blockDim.x >> 6);
    const long long n_live = (long long)ld_agent(as_global(P.counters) + FX_DCNT_LIVE);
    const int n_tiles = (int)((n_live + 63) / 64);
    if (P.mode & FX_MODE_INT_DEFER_OBST) {
        const int NC = (P.S - 1 + CH - 1) / CH;
        const int n_w = (int)gridDim.x * n_wave;
        double *my_lds = lds_step + (size_t)wave * FX_STEP_ITEM_DOUBLES(CH, P.K);
        // items = (T adjacent tiles, chunk), tile-group-major: the waves of a workgroup share the tiles' list entries, flag words and rows
        const int n_groups = (n_tiles + FX_STEP_T - 1) / FX_STEP_T;
        const int n_items = n_groups * NC;
        for (int item = (int)blockIdx.x * n_wave + wave; item < n_items; item += n_w) {
            const int grp = item / NC;
            obstacle_items<CH, FX_STEP_T>(P, grp * FX_STEP_T, item - grp * NC, n_live, my_lds);
        }
    }
    (void)lane;
    FX_MSTAMP(9);
    if (!grid_barrier(sa.bar + FX_BAR_WORDS, sa.bar_base + n_wg)) return;
    FX_MSTAMP(10);

    // ---- phase 3: selection; slices = the launch's workgroups of this agent ----
    fx_select_body<true>(P, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, n_tiles, sa.host_result, sa.seq, sa.dev_winner, sa.host_pkg,
                         sa.pkg_stride, sa.pkg_plane_rows);
    FX_MSTAMP(11);
}

