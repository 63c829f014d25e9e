// fx_obstacle_kernel.h -- the obstacle stage as its own (candidate x step)-parallel kernel.
//
// Prediction cost (collision_probability.py:264-299, called from partial_cost_functions.py:341-356) and the OBB-sum collision
// walk (planner.py:329-392, collision_check.py:110-200) of a candidate have no step-to-step dependency once
// (x, y, theta_gl) exist.  Inside the walk they run with the walk's occupancy -- one or two waves per SIMD, ~230 VGPRs --
// although a visit is ten independent FP64 operations.  For grids whose walk does not fill the chip the host therefore splits
// the step (FX_MODE_INT_DEFER_OBST): the walk kernel (built without the stage) materialises the planes and leaves the cost sum
// open at the prediction term; this kernel reads x / y / theta back from the planes, visits every (candidate, step, obstacle),
// closes the cost sum, sets the collision flag and writes the per-tile (cost, index) arg-min partials the selection reduces.
//
// Work items.  A tile = 64 entries of the agent's list of COSTED candidates (the walk appends them, fx_eval_kernel.h
// finish_candidate: a third of a production step's candidates is infeasible and has neither a prediction cost nor a collision
// check); a chunk = CH steps of the horizon; one wave = one (tile, chunk), with a wave-private operand table in LDS.  The list's
// length, the tile's list entries and the chunk's slice of the hot obstacle table are requested together at entry, the rows of
// x / y / theta behind them: problem -> (count, list, table) -> rows is the whole chain of dependent round trips.  The chunks of a
// tile meet either through global memory (WG = false: every wave stores its partial prediction sum per candidate and its
// collision ballot with agent-scope stores, waits for them and takes the tile's ticket; the wave that draws the last ticket
// closes the tile) or -- launches of at most 1 024 tiles whose chunks fit a workgroup -- in LDS (WG = true: the chunks are the
// waves of one workgroup, one barrier, wave 0 closes).  Either way the partials are added in chunk order: deterministic, and
// the two variants agree bit for bit.
//
// Operands.  The step index of a wave is uniform, so are the obstacle operands.  Of the five per (step, obstacle) -- the
// Cholesky-whitened inverse covariance l11, l12, l22 and the transformed centre cu, cw (fx_walk.h, ObsHot) -- the three
// multipliers come in through SCALAR loads straight from the hot table (a VALU instruction takes one scalar operand: each of
// them has its own FMA) and only the two addends travel through LDS, 16 B per visit.
//
// Collision.  Per step the ego box (rear axle + wb along the heading, state.py:30-39) is rebuilt from (x, y, theta) -- one
// sincos per (candidate, step), the box of the previous step is carried --, then exactly the walk's sequence: OBB-sum hull of
// boxes (i-1, i), one bounding circle for the wave's 64 hulls against which lane k tests obstacle k, the expanded circle test
// per lane for the survivors, the exact 4-axis test on the raw records for what is still near (same obb_hull / obb_overlap as
// the walk: decisions are those of the brute-force definition).
//
// Measured (MI355X, tools/c3_split.py, profiles/r4): config 3 walk 40.9 us + this kernel 31.7 us (+ selection 6.2) = 85 us per
// step against 89 us fused; config 4's batch of five agents 31.6 us; a config-5 agent with a bundle 97 vs 120 us fused; 3 060
// candidates and 1 M candidates slower -- the host picks it where two lanes share a candidate (200 ... 3 072 waves).  Its pure
// issue time is ~14 us; the rest is launch ramp, the three dependent round trips at entry and the closing stores.
#pragma once

#include "fx_eval_kernel.h"

namespace fxk {

// Prediction operands of one (step, obstacle): l11, l12, l22 in scalar registers, (cu, cw) in vector registers
struct ObsEntry {
    double l11, l12, l22;
    fx_d2 cc;
};
template <class HotPtr>
__device__ __forceinline__ ObsEntry obs_load(HotPtr hot_i, const fx_d2 *cc_i, int k) {
    const auto q = hot_i + (size_t)k * FX_HOT_STRIDE;
    ObsEntry e;
    e.l11 = q[FX_HOT_L11]; e.l12 = q[FX_HOT_L12]; e.l22 = q[FX_HOT_L22];
    e.cc = cc_i[k];
    return e;
}
// m^2 of one obstacle for the ego point (xr, yr) relative to the table's origin (fx_walk.h: same expression tree)
__device__ __forceinline__ double obs_msq(const ObsEntry &e, double xr, double yr) {
    const double u = fma(e.l11, xr, fma(e.l12, yr, -e.cc.x));
    const double w = fma(e.l22, yr, -e.cc.y);
    const double m = fma(u, u, w * w);
    return m * m;
}
// 1/a + 1/b + 1/c + 1/d over one reciprocal (fx_walk.h: same expression tree as the fused stage)
__device__ __forceinline__ double obs_four(double a, double b, double c, double d) {
    const double ab = a * b, cd = c * d;
    const double num = fma(a + b, cd, (c + d) * ab);
    return num * rcp_pred(ab * cd);
}

// grid = (max tiles x chunks, n_agents), block = 64, dynamic LDS = CH * K * 48 B.  CH: steps per item (compile time: the rows
// live in registers), WPS: waves per SIMD handed to the register allocator.
// WG (round 4): ALL chunks of a tile in ONE workgroup -- grid = (max tiles, n_agents), block = 64 x (chunks of the longest
// horizon of the launch), wave = chunk.  The chunks then meet in LDS behind one workgroup barrier and wave 0 closes the tile:
// no partial sums and ballots through global memory, no acknowledged stores, no ticket, no device-coherent re-loads -- the
// chain that the last chunk of every tile walked at ~1 us per hop.  Dynamic LDS = waves x (CH * K * 48 B) + waves x 64 x 8 B
// (partial sums) + waves x 8 B (collision ballots).
#ifndef FX_OBST_WG_WPE
#define FX_OBST_WG_WPE 1
#endif
// The kernel's body for one (tile, chunk) item.  COH (the one-launch step, fx_step_kernel.h; single-wave items): the walk ran in the
// SAME launch on other workgroups --
//   * the tables no kernel writes (hot table, records, step masks) are read through the constant address space: behind the grid
//     barrier a wave-uniform read from the global address space is no scalar load any more (fx_device.h, as_const);
//   * the list's count comes from the caller (one agent-scope load per workgroup, not one per item), the tile's list entries are an
//     agent-scope load: every wave of the walk appends to the list, its lines are shared between XCDs;
//   * the walk's per-candidate results -- flag words, rows, cost sums -- are read with PLAIN loads: they were stored write-through and
//     acknowledged before the barrier, no two workgroups of the walk share a cache line of them (a workgroup's candidates are
//     128-aligned), and this XCD's L2 was invalidated at the launch's start and has not read them since -- the load misses and fetches
//     what the walk wrote;
//   * what the selection phase reads is stored write-through.
// The arithmetic is the same instruction sequence either way.
template <int CH, bool WG, bool COH>
__device__ __forceinline__ void fx_obstacle_body(const DevProblem &P, double *__restrict__ lds_all, const int tile, const int chunk,
                                                 const long long n_live_known = -1) {
    static_assert(!COH || !WG, "the one-launch step runs single-wave items");
    auto LD = [](auto p) { return *p; };
    auto ST = [](auto p, auto v) { if (COH) st_agent(p, v); else *p = v; };
    const uint32_t mode = P.mode;
    if (!(mode & FX_MODE_INT_DEFER_OBST)) return;
    const int S = P.S, K = P.K;
    const int NC = (S - 1 + CH - 1) / CH;
    const int64_t C = P.C, ld = P.ld;
    const int n_wave = WG ? (int)(blockDim.x >> 6) : 1;
    const int64_t c0 = (int64_t)tile * 64;
    if (c0 >= C) return;
    const int lane = threadIdx.x & 63;
    // this wave's slice of the dynamic LDS; WG: the meeting area behind the waves' slices
    double *__restrict__ lds_dyn = lds_all + (WG ? (size_t)chunk * 6 * CH * (size_t)K : 0);
    double *__restrict__ sh_part = lds_all + (size_t)n_wave * 6 * CH * (size_t)K;                       // [n_wave][64]
    unsigned long long *__restrict__ sh_colm = reinterpret_cast<unsigned long long *>(sh_part + (size_t)n_wave * 64);   // [n_wave]
    FX_OSTAMP(0);
    // The tile's candidates: 64 entries of the list of costed candidates the walk has left (fx_eval_kernel.h, finish_candidate;
    // the walk is complete, plain loads).  Tiles past the end of the list only leave a neutral arg-min partial.
    // The kernel's floor is a chain of dependent round trips (problem -> count -> list -> rows), so the count, the tile's list
    // entries and the chunk's slice of the hot table are requested TOGETHER: the list slot c0 + lane always exists (c0 < C <= ld);
    // past the list's end it holds whatever an earlier step left -- clamped to a valid candidate, loaded from, never stored for.
    const FX_GLOBAL unsigned long long *cnt_live = as_global(P.counters) + FX_DCNT_LIVE;
    unsigned long long n_live_raw = 0ULL;
    int32_t g_listed = 0;
    if (!COH) {
        n_live_raw = *cnt_live;
        g_listed = as_global(P.obs_list)[c0 + lane];
    } else {
        n_live_raw = (unsigned long long)n_live_known;
        g_listed = ld_agent(as_global(P.obs_list) + (c0 + lane));
    }

    const int i_a = 1 + chunk * CH, i_b = min(S, i_a + CH);
    const int64_t ps = (int64_t)S * ld;
    const FX_GLOBAL double *__restrict__ pl = as_global(P.planes);
    const bool col_mode = (mode & FX_MODE_COLLISION) != 0 && K > 0;
    // the chunk's slice of the hot table -> LDS: per (step, obstacle) the addends (cu, cw) as one 16-byte pair, then the hull
    // circles (hx2, hy2, hr2, ck) as 32 bytes; entry e of the slice is handled by lane e, e + 64, ... (two entries in flight).
    // The first round's loads go out here, its LDS writes follow the row requests below.
    const auto hot_a = as_table<COH>(P.obs_hot) + (size_t)i_a * K * FX_HOT_STRIDE;   // (COH: constant address space, fx_device.h)
    fx_d2 *__restrict__ cc_tab = reinterpret_cast<fx_d2 *>(lds_dyn);   // [CH][K]
    double *__restrict__ circ_tab = lds_dyn + 2 * (size_t)CH * K;       // [CH][K][4]
    const int n_e = max((i_b - i_a) * K, 0);
    double hv[2][6];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int e = max(min(lane + 64 * u, n_e - 1), 0);
        const auto q = hot_a + (size_t)e * FX_HOT_STRIDE;
        const bool on = n_e > 0 && K > 0;
        hv[u][0] = on ? q[FX_HOT_CU] : 0.0; hv[u][1] = on ? q[FX_HOT_CW] : 0.0;
        hv[u][2] = on ? q[FX_HOT_HX2] : 0.0; hv[u][3] = on ? q[FX_HOT_HY2] : 0.0; hv[u][4] = on ? q[FX_HOT_HR2] : 0.0; hv[u][5] = on ? q[FX_HOT_CK] : 0.0;
    }
    auto park_tables = [&]() {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = lane + 64 * u;
            if (e < n_e) {
                cc_tab[e] = fx_d2{hv[u][0], hv[u][1]};
                circ_tab[4 * e + 0] = hv[u][2]; circ_tab[4 * e + 1] = hv[u][3]; circ_tab[4 * e + 2] = hv[u][4]; circ_tab[4 * e + 3] = hv[u][5];
            }
        }
        for (int e0 = lane + 128; e0 < n_e; e0 += 128) {   // more than 128 (step, obstacle) entries: K > 42 at three steps per chunk
            double v[2][6];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = min(e0 + 64 * u, n_e - 1);
                const auto q = hot_a + (size_t)e * FX_HOT_STRIDE;
                v[u][0] = q[FX_HOT_CU]; v[u][1] = q[FX_HOT_CW];
                v[u][2] = q[FX_HOT_HX2]; v[u][3] = q[FX_HOT_HY2]; v[u][4] = q[FX_HOT_HR2]; v[u][5] = q[FX_HOT_CK];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = e0 + 64 * u;
                if (e < n_e) {
                    cc_tab[e] = fx_d2{v[u][0], v[u][1]};
                    circ_tab[4 * e + 0] = v[u][2]; circ_tab[4 * e + 1] = v[u][3]; circ_tab[4 * e + 2] = v[u][4]; circ_tab[4 * e + 3] = v[u][5];
                }
            }
        }
    };
    // ---- everything the chunk needs, requested before the count and the flags are looked at (nearly every tile needs it) ----
    const int64_t g = (uint32_t)g_listed < (uint64_t)C ? (int64_t)g_listed : 0;
    const int64_t g_raw = g;   // (scratch rows are indexed by the candidate: lanes past the list's end hold SOME candidate and never store)
    const uint32_t f = LD(as_global(P.flags) + g);
    // rows i_a - 1 .. i_b - 1 of x, y (and theta with the collision stage)
    double xs[CH + 1], ys[CH + 1], ts[CH + 1];
#pragma unroll
    for (int j = 0; j <= CH; j++) {
        const int i = min(i_a - 1 + j, S - 1);
        xs[j] = LD(pl + ((int64_t)FX_PL_X * ps + (int64_t)i * ld + g));
        ys[j] = LD(pl + ((int64_t)FX_PL_Y * ps + (int64_t)i * ld + g));
        ts[j] = col_mode ? LD(pl + ((int64_t)FX_PL_THETA * ps + (int64_t)i * ld + g)) : 0.0;
    }
    // the closing wave of a workgroup (chunk 0) continues the candidate's cost sum: its two operands come with the rows
    double pre_cost = 0.0, pre_tail = 0.0;
    if (WG && chunk == 0) {
        pre_cost = LD(as_global(P.cost) + g);
        pre_tail = LD(as_global(P.cost_tail) + g);
    }
    const int64_t n_live = (int64_t)n_live_raw;
    if (c0 >= n_live) {
        if (chunk == 0 && lane == 0) {
            ST(as_global(P.part_cost) + tile, (double)INFINITY);
            ST(as_global(P.part_idx) + tile, (int64_t)0x7fffffffffffffffLL);
        }
        return;
    }
    const bool act = c0 + lane < n_live;
    park_tables();

    // where the prediction term sits in the (id-sorted) cost function
    int n_pred = -1;
    double w_pred = 0.0;
    bool has_tail = false;
    for (int n = 0; n < P.n_cost; n++)
        if (P.cost_id[n] == FX_COST_PREDICTION) { n_pred = n; w_pred = P.cost_w[n]; has_tail = n + 1 < P.n_cost; }
    // every chunk of a tile looks at the same 64 flag words: these are tile-uniform
    const bool do_pred = n_pred >= 0 && K > 0 && wave_any_bit(act && (f & FX_FLAG_COSTED));
    const bool do_col = col_mode && wave_any_bit(act && (f & FX_FLAG_SELECTABLE));
    FX_OSTAMP(1);

    double acc = 0.0;
    bool collided = false;
    const bool work = do_pred || do_col;
    if (!work && chunk != 0) return;   // nothing to add: chunk 0 closes the tile on its own (tile-uniform: no barrier is skipped)
    if (work && chunk < NC) {          // (WG: a launch sized for a longer horizon has waves without a chunk of this agent)
        const auto rec = as_table<COH>(P.obs_rec);
        const auto pmask = as_table<COH>(P.obs_pmask);
        const auto hmask = as_table<COH>(P.obs_hmask);
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
        double sn_prev = 0.0, cs_prev = 1.0;
        bool have_prev = false;   // wave-uniform: (sn_prev, cs_prev) = sin / cos of theta at step i - 1
        FX_OSTAMP(2);
#pragma unroll
        for (int j = 1; j <= CH; j++) {
            const int i = i_a + j - 1;
            if (i < i_b) {
                const unsigned long long pm = do_pred ? uniform_u64(pmask[i]) : 0ULL;
                const unsigned long long hm = (do_col && i >= 2) ? uniform_u64(hmask[i]) : 0ULL;
                const auto hot_i = hot_a + (size_t)(j - 1) * K * FX_HOT_STRIDE;
                const fx_d2 *__restrict__ cc_i = cc_tab + (size_t)(j - 1) * K;
                const double *__restrict__ circ_i = circ_tab + 4 * (size_t)(j - 1) * K;
                // ---- prediction cost: sum over the obstacles of 1 / m^2 (collision_probability.py:283-292) ----
                if (pm) {
                    const double xr = xs[j] - ox, yr = ys[j] - oy;
                    double ssum = 0.0;
                    auto ld_ = [&](int k) { return obs_load(hot_i, cc_i, k); };
                    if (pm == full) {
                        // four obstacles per iteration in two alternating register sets: the loads of one pair are in flight
                        // while the other pair is consumed; every complete group of four shares one reciprocal (fx_walk.h: same
                        // grouping, same expression tree)
                        const int kz = K - 1;
                        ObsEntry a = ld_(0), b = ld_(min(1, kz)), c, d;
                        int k = 0;
                        for (; k + 3 < K; k += 4) {
                            c = ld_(k + 2); d = ld_(k + 3);
                            const double qa = obs_msq(a, xr, yr), qb = obs_msq(b, xr, yr);
                            a = ld_(min(k + 4, kz)); b = ld_(min(k + 5, kz));
                            ssum += obs_four(qa, qb, obs_msq(c, xr, yr), obs_msq(d, xr, yr));
                        }
                        // entries k (a) and k + 1 (b) are loaded where they exist; up to three remain
                        double s1 = 0.0;
                        if (k < K) ssum += rcp_pred(obs_msq(a, xr, yr));
                        if (k + 1 < K) s1 += rcp_pred(obs_msq(b, xr, yr));
                        if (k + 2 < K) ssum += rcp_pred(obs_msq(ld_(k + 2), xr, yr));
                        ssum += s1;
                    } else {
                        unsigned long long m = pm;
                        while (m) {
                            const int k = __builtin_ctzll(m);
                            m &= m - 1;
                            ssum += rcp_pred(obs_msq(ld_(k), xr, yr));
                        }
                    }
                    // anything not finite (an obstacle centre hit to the last bit, a covariance without a Cholesky factor: the
                    // host leaves a zero entry) sends the step to the reference form on the raw records -- rare
                    if (wave_any_bit(!(ssum < 1e300))) {
                        const auto rec_i = rec + (int64_t)i * K * 12;
                        double a = 0.0;
                        unsigned long long m = pm;
                        while (m) {
                            const int k = __builtin_ctzll(m);
                            m &= m - 1;
                            const auto q = rec_i + k * 12;
                            const double e0 = xs[j] - q[0], e1 = ys[j] - q[1];
                            const double r0 = fma(e1, q[4], e0 * q[2]), r1 = fma(e1, q[5], e0 * q[3]);
                            const double mq = fma(r1, e1, r0 * e0);
                            const double mm = mq * mq;
                            a += mm > 0.0 ? rcp_pred(mm) : 1.0 / mm;
                        }
                        ssum = !(ssum < 1e300) ? a : ssum;
                    }
                    acc += ssum;
                }
                // ---- collision: OBB-sum hull of ego boxes (i-1, i) against the obstacle hulls of this step ----
                if (hm) {
                    // Wave-level cull before anything of the boxes is built (fx_walk.h has the same cull on the box centres; here
                    // not even the headings exist yet).  Box centre c = p + wb (cos, sin) theta of the rear-axle point p = (x, y):
                    // the midpoint of the two centres lies within |wb| of the midpoint m of the two rear-axle points and
                    // |c1 - c0| <= |p1 - p0| + 2 |wb|, so the hull lies inside the disc around m with radius
                    // |wb| + sqrt(2) (|p1 - p0| / 2 + |wb| + hd), hd = |(L/2, W/2)|.  One bounding disc for the wave's 64 hulls
                    // (around the first lane's m), lane k tests obstacle k's circle against it; for most (tile, step) pairs
                    // nothing survives and neither the two sincos nor the hull are evaluated.
                    unsigned long long cand;
                    {
                        const double mxr = fma(0.5, xs[j - 1] + xs[j], -ox), myr = fma(0.5, ys[j - 1] + ys[j], -oy);
                        const double tx = xs[j] - xs[j - 1], ty = ys[j] - ys[j - 1];
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
                    have_prev = have_prev && cand != 0ULL;
                    if (cand) {
                        // the two ego boxes (rear axle + wb along the heading, state.py:30-39) and their OBB-sum hull; the heading
                        // of step i - 1 is carried when that step built a hull too
                        double s0 = sn_prev, c0 = cs_prev, s1, c1;
                        if (!have_prev) fxm::sincos(ts[j - 1], &s0, &c0);   // (wave-uniform: the previous step went the same way)
                        fxm::sincos(ts[j], &s1, &c1);
                        sn_prev = s1; cs_prev = c1;
                        const Obb hull = obb_hull(fma(wb, c0, xs[j - 1]), fma(wb, s0, ys[j - 1]), c0, s0, fma(wb, c1, xs[j]), fma(wb, s1, ys[j]),
                                                  c1, s1, half_len, half_wid);
                        const auto rec_i = rec + (int64_t)i * K * 12;
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
                            if (wave_any_bit(!(gq > gap_margin))) collided |= obb_overlap(hull, rec_i + k * 12 + 6);
                        } while (cand);
                        have_prev = true;
                    }
                } else {
                    have_prev = false;
                }
            }
        }
    }
    FX_OSTAMP(4);
    FX_GLOBAL double *__restrict__ part = as_global(P.obs_part);
    FX_GLOBAL unsigned long long *__restrict__ colm = as_global(P.obs_colm);
    FX_GLOBAL unsigned int *__restrict__ ticket = as_global(P.obs_ticket);
    const int64_t n_tiles = (C + 63) / 64;
    if (WG) {
        // ---- the chunks meet in LDS: partial sums and ballots, one barrier, wave 0 goes on ----
        if (work) {
            sh_part[chunk * 64 + lane] = acc;
            const unsigned long long cm = __builtin_amdgcn_ballot_w64(collided);
            if (lane == 0) sh_colm[chunk] = cm;
            __syncthreads();
            if (chunk != 0) return;
        }
    } else if (work) {
        // hand-off: agent-scope stores, acknowledged (vmcnt 0) before the ticket is taken -- whoever draws the last ticket
        // sees every chunk's partial
        if (act) __hip_atomic_store(part + (int64_t)chunk * ld + g_raw, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long cm = __builtin_amdgcn_ballot_w64(collided);
        if (lane == 0) __hip_atomic_store(colm + (int64_t)chunk * n_tiles + tile, cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(ticket + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        FX_OSTAMP(5);
        if (t != (unsigned)(NC - 1)) return;
        if (lane == 0) __hip_atomic_store(ticket + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- close the tile: partials in chunk order ----
    double pred = 0.0;
    unsigned long long cmask = 0ULL;
    if (work && WG) {
        for (int q = 0; q < NC; q++) {   // chunk order, as the other variant adds them: the two agree bit for bit
            pred += sh_part[q * 64 + lane];
            cmask |= sh_colm[q];
        }
    } else if (work) {
        for (int q = 0; q < NC; q++) {
            pred += __hip_atomic_load(part + (int64_t)q * ld + g_raw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cmask |= __hip_atomic_load(colm + (int64_t)q * n_tiles + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    uint32_t fl = f;
    const bool costed = (fl & FX_FLAG_COSTED) != 0, selectable = (fl & FX_FLAG_SELECTABLE) != 0;
    double total = WG ? pre_cost : as_global(P.cost)[g];
    if (n_pred >= 0) {
        // the walk left the running sum in front of the prediction term: continue the same sequence of additions
        double sum = total;
        sum += w_pred * pred;
        if (has_tail) sum += WG ? pre_tail : as_global(P.cost_tail)[g];
        total = 0.0 + sum;
        if (act) {
            ST(as_global(P.cost) + g, costed ? total : 0.0);
            if (mode & FX_MODE_WRITE_COSTMAP) ST(as_global(P.costmap) + ((int64_t)n_pred * ld + g), costed ? pred : 0.0);
        }
    }
    if (selectable && (mode & FX_MODE_COLLISION) && ((cmask >> lane) & 1ULL)) {
        fl |= FX_FLAG_COLLISION;
        if (act) ST(as_global(P.flags) + g, fl);
    }
    // (cost, index) arg-min of the tile: the lanes hold the list's candidates in no particular order, so among the lanes with the
    // minimum cost the smallest index is reduced as well
    const bool eligible = act && selectable && !(fl & (FX_FLAG_COLLISION | FX_FLAG_BOUNDARY)) && total == total;
    const double bc = eligible ? total : INFINITY;
    double m = bc;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off));
    long long bi = (eligible && bc == m) ? (long long)(g_raw + P.g_base) : 0x7fffffffffffffffLL;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const long long o = __shfl_xor(bi, off); bi = o < bi ? o : bi; }
    if (lane == 0) {
        ST(as_global(P.part_cost) + tile, m);
        ST(as_global(P.part_idx) + tile, (int64_t)bi);
    }
    FX_OSTAMP(15);
}

template <int CH, int WPS, bool WG = false>
__global__ __launch_bounds__(WG ? 1024 : 64, WG ? FX_OBST_WG_WPE : WPS) void fx_obstacle_kernel(const DevProblem *__restrict__ probs) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];  // per wave: (cu, cw) [CH][K][2] | hull circles [CH][K][4]
    const DevProblem &P = probs[blockIdx.y];
    const int NC = (P.S - 1 + CH - 1) / CH;
    const int tile = WG ? (int)blockIdx.x : (int)(blockIdx.x / NC);
    const int chunk = WG ? (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(blockIdx.x - tile * NC);
    fx_obstacle_body<CH, WG, false>(P, lds_all, tile, chunk);
}

}  // namespace fxk
