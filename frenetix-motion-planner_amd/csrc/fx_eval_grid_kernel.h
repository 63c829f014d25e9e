// fx_eval_grid_kernel.h -- evaluation kernel for the Cartesian-product sampling grid (t x v x d).
//
// The reference builds ONE longitudinal quartic per (t, v) pair and reuses it for every lateral end state d
// (reactive_planner.py:151-158), but check_feasibility then re-derives everything per candidate.  On the grid,
// every per-step quantity that depends only on the longitudinal motion is identical for all nD candidates of a
// pair:  s, s', s'' -- the reference-segment lookup -- theta_ref, kappa_ref, kappa_ref' -- the foot point and the
// unit normal of the projection -- the validity / pre-filter predicates.  That is more than half of the
// arithmetic of a step and all of its divergent LDS traffic.
//
// So each workgroup first computes, cooperatively (one (pair, step) item per lane), the rows of that shared
// "longitudinal table" for the few pairs its 256/G candidates span and keeps them in LDS (128 B per row);
// the per-candidate walk then reads its row with a broadcast LDS read and only does the lateral polynomial,
// the Frenet->Cartesian kinematics, the constraints, the costs and the stores.  Because s(t) comes from the
// table, the horizon extension needs no recurrence in the walk, which also makes the G-lane horizon split
// carry-free except for theta/kappa of step i-1.
//
// Arithmetic is term-by-term the same as in the generic kernel (fx_eval_kernel.h): a value that was computed per
// candidate there is computed per pair here with the same expression, so both kernels produce identical bits.
// Used when sampling ranges (not a matrix) are given, no windowed (Simpson) cost is active and the rows of a
// workgroup fit the LDS budget; otherwise the generic kernel runs.
#pragma once

#include "fx_eval_kernel.h"

// Workgroup size is a launch parameter (64, 128 or 256 lanes); the host guarantees that the rows of a workgroup
// ((CPB + nD - 2) / nD + 1 pairs) fit its dynamic LDS.
// The kernel's body is a function of its own: fx_step_kernel.h runs it as the first phase of a whole plan step in ONE launch
// (MEGA: every per-candidate output a later phase of the same launch reads -- cost sums, flag words, list entries, coefficients,
// cost map -- is stored write-through, agent scope; the returns below are then returns into that kernel, never past a barrier
// some other wave of the workgroup still waits at).
template <int G, bool BUNDLE, bool OBST, int WPE, bool WSPLIT, bool MEGA = false>
__device__ __forceinline__ void fx_eval_grid_body(const DevProblem *__restrict__ probs, const FuseArgs &fuse) {
    using namespace fxk;
    const int BLK = blockDim.x;
    const int CPB = BLK / G;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];  // [5][S] time powers | rows[n_pairs][S]
    __shared__ double red_cost[FX_BLOCK / 64];
    __shared__ long long red_idx[FX_BLOCK / 64];
    __shared__ unsigned int red_cnt[2 + FX_NUM_REASONS + 1];

    __shared__ int32_t sh_cost_id[FX_NUM_COSTS];
    __shared__ double sh_cost_w[FX_NUM_COSTS];
    __shared__ __attribute__((aligned(16))) double sh_atan_k[FX_ATAN_K];   // atan's polynomial coefficients (walk_step with KTAB)

    // every field is fetched by scalar loads issued together at kernel entry (one latency) instead of one
    // dependent s_load wherever a field is first used
    const DevProblem &Pg = probs[blockIdx.y];
    const ProblemRegs P = ProblemRegs::load(Pg, sh_cost_id, sh_cost_w);
    const int tid = threadIdx.x;
    const int64_t C = P.C;
    if ((int64_t)blockIdx.x * CPB >= C) return;
    FX_STAMP(0);
    // parts of a candidate: adjacent lanes (lane split) or the same lane of G lane-groups (wave split, CPB % 64 == 0)
    // wave split: the part is the same for the 64 lanes of a wave -- say so (scalar loop control, scalar row base of the stores)
    const int part = G == 1 ? 0 : (WSPLIT ? __builtin_amdgcn_readfirstlane(tid / CPB) : (tid & (G - 1)));
    const int cand_local = G == 1 ? tid : (WSPLIT ? tid - part * CPB : tid / G);
    const int64_t c0 = (int64_t)blockIdx.x * CPB;  // first local candidate of this workgroup
    const int64_t g_raw = c0 + cand_local;
    const bool active = g_raw < C;
    const int64_t g = active ? g_raw : C - 1;

    const int M = P.M, S = P.S;
    const int nD = P.nD, nV = P.nV;
    // pairs of this workgroup; the sampling values of this lane's first table item and of its own candidate are
    // requested now so that their latency overlaps phase 1
    const int64_t gbase = P.g_base;
    const int64_t pair0 = (c0 + gbase) / nD;
    const int64_t c_last = min(c0 + CPB, C) - 1;
    const int n_pairs = (int)((c_last + gbase) / nD - pair0) + 1;
    const int n_pairs_max = (CPB + nD - 2) / nD + 1;  // what the host sized the row block for
    double T_item0 = 0.0, v_item0 = 0.0;
    if (tid < n_pairs * S) {
        const int64_t pair = pair0 + tid / S;
        const int it = (int)(pair / nV), iv = (int)(pair - (int64_t)it * nV);
        T_item0 = as_global(P.t_samp)[it]; v_item0 = as_global(P.v_samp)[iv];
    }
    const int64_t gg = g + gbase;
    const int64_t pair = gg / nD;
    const int id = (int)(gg - pair * nD);
    const int it = (int)(pair / nV), iv = (int)(pair - (int64_t)it * nV);
    const double T = as_global(P.t_samp)[it], v1 = as_global(P.v_samp)[iv], d1 = as_global(P.d_samp)[id];
    // ---- phase 1: one round of global loads -- time powers, knot arc lengths, cost ids / weights -> LDS ----
    // dynamic LDS: time table | rows | exchange block (G > 1) | tail.  The tail holds the knots' arc lengths during the
    // prologue (binary search of make_lon_row) and the waves' hot obstacle blocks during the walk -- the two never live at the
    // same time (a barrier separates them), and sharing the bytes is what lets a CU hold three 256-lane workgroups of config 5
    // (51 samples, 52 lateral offsets: 54.5 KB apart, 50.5 KB shared)
    double *__restrict__ tpw = lds_dyn;                                // [S][FX_TP] time table
    LonRow *__restrict__ rows = reinterpret_cast<LonRow *>(tpw + FX_TP * S);   // FX_TP * S * 8 is a multiple of 16
    char *__restrict__ lds_tail = reinterpret_cast<char *>(rows + (size_t)n_pairs_max * S) + (G > 1 ? (size_t)64 * BLK : 0);
    double *__restrict__ rpos = reinterpret_cast<double *>(lds_tail);  // [M] arc length of the knots (binary search)
    // Lane split (the planner-sized decompositions: every lane of a wave walks another step): the obstacle loops read one
    // 96-byte record per (step, obstacle) and lane.  From global memory that is a dependent round trip per visited obstacle -- the
    // walk of a 630-candidate step with five obstacles took 7 - 9 us against 3 us without obstacles (tools/probe_timeline.py) --, so
    // the workgroup copies the agent's whole record table and the step masks into LDS with the other tables of phase 1 (where they
    // fit: FX_REC_LDS_MAX; the host sizes the dynamic LDS for them and says so in the agent's mode word, fx_api.hip)
    constexpr bool LSTAGE = OBST && !(G == 1 || WSPLIT);
    const int rec_n = P.K > 0 ? S_rec_doubles(P.S, P.K) : 0;
#ifdef FX_NO_LSTAGE   // (probe builds: A/B of the staging)
    const bool rec_staged = false;
#else
    const bool rec_staged = LSTAGE && rec_n > 0 && (P.mode & FX_MODE_INT_REC_LDS) != 0;
#endif
    double *__restrict__ rec_lds = reinterpret_cast<double *>(lds_tail + sizeof(double) * (((size_t)P.M + 1) & ~(size_t)1));
    unsigned long long *__restrict__ pm_lds = reinterpret_cast<unsigned long long *>(rec_lds + rec_n);   // [S] | [S]
    // the knots themselves are only touched once per (pair, step) item: read them through L1/L2
    const FX_GLOBAL double *__restrict__ kn = as_global(P.ref);
    if constexpr (G >= 4) {
        // Planner-sized decompositions: the step is a chain of round trips, so EVERY table's first batch of loads is requested
        // before the first LDS write (a copy loop per table is a round trip per table -- and per iteration: 1.9 us of phase 1 grew
        // to 3.4 us when the record table joined as its own loop, tools/probe_timeline.py); what a batch does not cover loops on.
        const FX_GLOBAL double *__restrict__ tsrc = as_global(P.tpow);
        const bool masks = OBST && P.K > 0;
        const int it = min(tid, S - 1);
        const double t1 = tsrc[it], t2 = tsrc[S + it], t3 = tsrc[2 * S + it], t4 = tsrc[3 * S + it], t5 = tsrc[4 * S + it];
        unsigned long long m0 = 0ULL, m1 = 0ULL;
        if (masks) { m0 = as_global(P.obs_pmask)[it]; m1 = as_global(P.obs_hmask)[it]; }
        double rp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) rp[u] = kn[(int64_t)min(tid + u * BLK, M - 1) * FX_REF_FIELDS];
        const FX_GLOBAL fx_d2 *__restrict__ rsrc = reinterpret_cast<const FX_GLOBAL fx_d2 *>(as_global(P.obs_rec));   // (256-byte aligned slot)
        fx_d2 *__restrict__ rdst = reinterpret_cast<fx_d2 *>(rec_lds);
        const int n2 = rec_n / 2;   // (12 doubles per record: even)
        fx_d2 rv[8];
        if (rec_staged) {
#pragma unroll
            for (int u = 0; u < 8; u++) rv[u] = rsrc[min(tid + u * BLK, n2 - 1)];
        }
        int32_t cid = 0;
        double cw = 0.0;
        if (tid < P.n_cost) { cid = Pg.cost_id[tid]; cw = Pg.cost_w[tid]; }
        // ---- LDS writes ----
        if (tid < S) {
            fill_time_row(tpw + tid * FX_TP, t1, t2, t3, t4, t5);
            if (masks) {
                if (rec_staged) { pm_lds[tid] = m0; pm_lds[S + tid] = m1; }
                else if (G == 1 || WSPLIT) {   // (four lanes per candidate on four waves: the masks ride in the time table)
                    unsigned long long *m = reinterpret_cast<unsigned long long *>(tpw + tid * FX_TP + 12);
                    m[0] = m0; m[1] = m1;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (tid + u * BLK < M) rpos[tid + u * BLK] = rp[u];
        if (rec_staged) {
#pragma unroll
            for (int u = 0; u < 8; u++) if (tid + u * BLK < n2) rdst[tid + u * BLK] = rv[u];
        }
        if (tid < P.n_cost) { sh_cost_id[tid] = cid; sh_cost_w[tid] = cw; }
        // ---- what the first batch does not cover ----
        for (int i = tid + BLK; i < S; i += BLK) {
            fill_time_row(tpw + i * FX_TP, tsrc[i], tsrc[S + i], tsrc[2 * S + i], tsrc[3 * S + i], tsrc[4 * S + i]);
            if (rec_staged) { pm_lds[i] = as_global(P.obs_pmask)[i]; pm_lds[S + i] = as_global(P.obs_hmask)[i]; }
        }
        for (int i0 = tid + 4 * BLK; i0 < M; i0 += 4 * BLK) {
#pragma unroll
            for (int u = 0; u < 4; u++) rp[u] = kn[(int64_t)min(i0 + u * BLK, M - 1) * FX_REF_FIELDS];
#pragma unroll
            for (int u = 0; u < 4; u++) if (i0 + u * BLK < M) rpos[i0 + u * BLK] = rp[u];
        }
        if (rec_staged) {
            for (int e0 = tid + 8 * BLK; e0 < n2; e0 += 8 * BLK) {
#pragma unroll
                for (int u = 0; u < 8; u++) rv[u] = rsrc[min(e0 + u * BLK, n2 - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) if (e0 + u * BLK < n2) rdst[e0 + u * BLK] = rv[u];
            }
        }
        if (WPE >= 3 && OBST && tid < FX_ATAN_K) sh_atan_k[tid] = fxm::fx_ktab[FX_ATAN_K0 + tid];
    } else {
        const FX_GLOBAL double *__restrict__ tsrc = as_global(P.tpow);
        for (int i = tid; i < S; i += BLK) {
            fill_time_row(tpw + i * FX_TP, tsrc[i], tsrc[S + i], tsrc[2 * S + i], tsrc[3 * S + i], tsrc[4 * S + i]);
            if (OBST && (G == 1 || WSPLIT) && P.K > 0) {   // staged obstacle stage (K <= 64: one mask word per step)
                unsigned long long *m = reinterpret_cast<unsigned long long *>(tpw + i * FX_TP + 12);
                m[0] = as_global(P.obs_pmask)[i];
                m[1] = as_global(P.obs_hmask)[i];
            }
        }
        for (int i = tid; i < M; i += BLK) rpos[i] = kn[(int64_t)i * FX_REF_FIELDS];
        if (rec_staged) {   // (G = 2 lane split: the plain loops -- these kernels' register allocation is left alone)
            const FX_GLOBAL double *__restrict__ rsrc = as_global(P.obs_rec);
            for (int e = tid; e < rec_n; e += BLK) rec_lds[e] = rsrc[e];
            for (int i = tid; i < S; i += BLK) { pm_lds[i] = as_global(P.obs_pmask)[i]; pm_lds[S + i] = as_global(P.obs_hmask)[i]; }
        }
        if (tid < P.n_cost) { sh_cost_id[tid] = Pg.cost_id[tid]; sh_cost_w[tid] = Pg.cost_w[tid]; }
        if (WPE >= 3 && OBST && tid < FX_ATAN_K) sh_atan_k[tid] = fxm::fx_ktab[FX_ATAN_K0 + tid];
    }
    if (tid < 2 + FX_NUM_REASONS) red_cnt[tid] = 0;
    __syncthreads();
    const double *__restrict__ tp = tpw;
    FX_STAMP(1);

    const double dt = P.dt;
    const bool low_vel = P.low_vel_mode != 0;
    const bool D = (P.mode & FX_MODE_DRAW_TRAJ_SET) != 0;
    const bool dbg = D || (P.mode & FX_MODE_KINEMATIC_DEBUG) != 0;
    const bool do_collision = OBST && (P.mode & FX_MODE_COLLISION) != 0;
    const bool bundle = BUNDLE && (P.mode & FX_MODE_WRITE_BUNDLE) != 0;
    const double a_max = P.veh.a_max;
    const int64_t ld = P.ld;
    const double s0 = P.x0_lon[0], ss0 = P.x0_lon[1], sss0 = P.x0_lon[2];

    // ---- prologue: longitudinal table of the pairs this workgroup touches ----
    // rows carry cos / sin of the reference heading when some stage needs the ego footprint
    const bool want_trig = OBST && ((do_collision && P.K > 0) || ((P.mode & FX_MODE_ROAD_BOUNDARY) && P.n_bound > 0));
    const double rp_first = rpos[0], rp_last = rpos[M - 1];
    const double guess_scale = fdiv((double)(M - 1), rp_last - rp_first);
    for (int item = tid; item < n_pairs * S; item += BLK) {
        const int pl = item / S, i = item - pl * S;
        double T, v1;
        if (item == tid) { T = T_item0; v1 = v_item0; }  // fetched with phase 1
        else {
            const int64_t pair = pair0 + pl;
            const int it = (int)(pair / nV), iv = (int)(pair - (int64_t)it * nV);
            T = as_global(P.t_samp)[it]; v1 = as_global(P.v_samp)[iv];
        }
        double cl3, cl4, cl5;
        lon_coeffs(P.lon_mode, s0, ss0, sss0, T, v1, 0.0, cl3, cl4, cl5);
        int traj_len = (int)ceil((T + dt) / dt);
        traj_len = traj_len > S ? S : (traj_len < 1 ? 1 : traj_len);
        rows[item] = make_lon_row(
            i, S, M, dt, a_max, s0, ss0, .5 * sss0, cl3, cl4, cl5, traj_len, tp, rp_first, rp_last, guess_scale, want_trig,
            [&](int k) {
                const FX_GLOBAL double *q = kn + (int64_t)k * FX_REF_FIELDS;
                Knot kt;
                kt.pos = q[0]; kt.theta = q[1]; kt.curv = q[2]; kt.curv_d = q[3]; kt.x = q[4]; kt.y = q[5]; kt.nx = q[6]; kt.ny = q[7];
                return kt;
            },
            [&](int k) { return rpos[k]; }, (P.mode & FX_MODE_PROJ_PSEUDO_NORMAL) != 0);
    }
    __syncthreads();
    FX_STAMP(2);

    // ---- candidate: lateral quintic (reactive_planner.py:158-171) ----
    const double d0 = P.x0_lat[0], dd0 = P.x0_lat[1], ddd0 = P.x0_lat[2];
    const LonRow *__restrict__ my = rows + (int)(pair - pair0) * S;
    double cl3, cl4, cl5;
    lon_coeffs(P.lon_mode, s0, ss0, sss0, T, v1, 0.0, cl3, cl4, cl5);
    double tau = T;
    if (low_vel) {
        const double cl0 = s0, cl1 = ss0, cl2 = .5 * sss0;
        double t2 = T * T, t3 = t2 * T, t4 = t2 * t2, t5 = t3 * t2;  // evaluate_state_at_tau, polynomial_trajectory.py:213-216
        double s_lon_goal = (cl0 + cl1 * T + cl2 * t2 + cl3 * t3 + cl4 * t4 + cl5 * t5) - s0;
        if (s_lon_goal <= 0) s_lon_goal = T;
        tau = s_lon_goal;
    }
    LatPoly L;
    {
        double T2 = tau * tau, T3 = T2 * tau, T4 = T3 * tau, T5 = T4 * tau;
        double b0 = d1 - d0 - dd0 * tau - .5 * ddd0 * T2;
        double b1 = 0.0 - dd0 - ddd0 * tau;
        double b2 = 0.0 - ddd0;
        L.set(d0, dd0, .5 * ddd0, fdiv(10.0 * b0 - 4.0 * b1 * tau + .5 * b2 * T2, T3),
              fdiv(-15.0 * b0 + 7.0 * b1 * tau - b2 * T2, T4), fdiv(6.0 * b0 - 3.0 * b1 * tau + .5 * b2 * T2, T5));
    }
    int traj_len = (int)ceil((T + dt) / dt);
    traj_len = traj_len > S ? S : (traj_len < 1 ? 1 : traj_len);

    if (bundle && active && part == 0) {
        FX_GLOBAL double *__restrict__ co = as_global(P.coeffs) + g;
        // (write-through where the step's last workgroup gathers the winner package itself, fx_tail.h)
        const bool wt = MEGA || (FX_TAIL_IN_KERNEL(G, false) && fuse.host_result != nullptr && (fuse.tail() & FX_TAIL_PACKAGE));
        const double cv[FX_COEFF_ROWS] = {s0, ss0, .5 * sss0, cl3, cl4, cl5, L.c0, L.c1, L.c2, L.c3, L.c4, L.c5,
                                          tau};  // tau: PolynomialTrajectory.delta_tau of the lateral polynomial (reactive_planner.py:161-171)
#pragma unroll
        for (int q = 0; q < FX_COEFF_ROWS; q++) st_out(co + q * ld, cv[q], wt);
        st_out(as_global(P.traj_len) + g, (int32_t)traj_len, wt);
    }

    auto lat_eval = [&](int i, double u_lowvel, double &d, double &dv, double &da) {
        LatU U;
        if (low_vel) U.from_parameter(u_lowvel);
        else U.from_table(tp + i * FX_TP);
        L.eval(U, d, dv, da);
    };
    // lateral value the extension holds: d[traj_len-1] (reactive_planner.py:344)
    double d_ext, dv_u, da_u;
    lat_eval(traj_len - 1, my[traj_len - 1].u1, d_ext, dv_u, da_u);

    // ---- this lane's chunk of the horizon ----
    const int CH = G == 1 ? S : (S + G - 1) / G;
    const int i_begin = part * CH;
    const int i_end = min(S, i_begin + CH);
    // one step per lane (32 lanes per candidate, horizons up to 32 samples): no carry-in step -- the left lane walks step i - 1 in
    // the same call and hands its heading, curvature and box over (walk_step, `neigh`)
    const bool neigh = G == 32 && !WSPLIT && CH == 1;
    const int i_first = (G > 1 && part > 0 && !neigh) ? i_begin - 1 : i_begin;

    // The walk's wave-uniform doubles live in VGPRs (KV): the loop keeps ~50 lane masks and pointers in scalar registers and
    // the allocator otherwise parks these constants in VGPR lanes and reads them back (v_readlane) every step.
    auto KV = [](double x) { return WPE <= 3 ? fxk::uniform_to_vgpr(x) : x; };  // 256 VGPRs to spend only at two waves per SIMD
    StepConst K;
    K.dt = KV(dt); K.r_dt = KV(1.0 / dt); K.kappa_max = KV(P.veh.kappa_max); K.a_max = KV(a_max); K.v_switch = KV(P.veh.v_switch);
    K.av_switch = KV(a_max * P.veh.v_switch); K.v_des = KV(P.v_des); K.wb = KV(P.veh.wb_rear_axle); K.half_len = KV(P.veh.length / 2);
    K.half_wid = KV(P.veh.width / 2); K.S = S; K.half = S / 2; K.K = P.K; K.low_vel = low_vel; K.dbg = dbg;
    K.do_collision = do_collision; K.store_wt = (P.mode & FX_MODE_INT_STORE_WT) != 0;
    K.n_bound = (OBST && (P.mode & FX_MODE_ROAD_BOUNDARY)) ? P.n_bound : 0; K.bound_d_reach = P.bound_d_reach;
    K.ox = KV(P.hot_origin[0]); K.oy = KV(P.hot_origin[1]); K.gap_margin = KV(P.hot_gap_margin);
    K.cull_r0 = (float)(1.41423 * sqrt(P.veh.length * P.veh.length + P.veh.width * P.veh.width) * 0.5);
    K.atan_k = (fxm::lds_cptr)sh_atan_k;
    const BoundView Bv{as_global(P.bound_piece), as_global(P.bound_bin), as_global(P.bound_item)};
    const FX_GLOBAL double *__restrict__ obs_rec = as_global(P.obs_rec);
    const FX_GLOBAL unsigned long long *__restrict__ obs_pmask = as_global(P.obs_pmask);
    const FX_GLOBAL unsigned long long *__restrict__ obs_hmask = as_global(P.obs_hmask);

    StepCarry Cy;
    Cy.th_prev = P.x0_orientation; Cy.kap_prev = 0.0; Cy.bx_prev = Cy.by_prev = Cy.ux_prev = Cy.uy_prev = 0.0;
    // one step per lane: the 32 lanes of a candidate hold its steps in order, so ONE ballot of "my step moves" tells every lane
    // the last moving step in front of its own (a lane-by-lane scan back through the LDS rows cost the workgroups whose pair ends
    // in the extension at the slowest sampled velocity -- every step behind T stands still there -- 2 us: they were the last to take
    // their ticket, and the step's tail waits for the last)
    // (two steps per lane -- 16 lanes per candidate, horizons up to 32 samples -- the same from two ballots, interleaved)
    const bool pairs = G == 16 && !WSPLIT && CH == 2;
    unsigned long long moving_lanes = 0ULL, moving_lanes1 = 0ULL;
    if (neigh || pairs) {
        moving_lanes = __ballot(!low_vel && i_begin < S && (my[min(i_begin, S - 1)].flags & LON_MOVING) != 0);
        if (pairs) moving_lanes1 = __ballot(!low_vel && i_begin + 1 < S && (my[min(i_begin + 1, S - 1)].flags & LON_MOVING) != 0);
    }
    if (G > 1 && !low_vel && i_first > 0 && i_first < S && !(my[i_first].flags & LON_MOVING)) {
        // the carry-in step keeps the previous heading: the last moving step in front of it (:447)
        int j;
        if (neigh) {
            const unsigned int grp = (unsigned int)(moving_lanes >> ((tid & 63) & 32));        // this candidate's 32 steps
            const unsigned int below = grp & ((1u << part) - 1u);                              // (part = step, 1 <= part <= 31 here)
            j = below ? 31 - __clz((int)below) : -1;
        } else if (pairs) {
            auto spread = [](unsigned int x) {   // bit k of a 16-bit word -> bit 2k
                x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu;
                x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
                return x;
            };
            const int sh = (tid & 63) & 48;                                                    // this candidate's 16 lanes
            const unsigned int steps = spread((unsigned int)(moving_lanes >> sh) & 0xffffu) |
                                       (spread((unsigned int)(moving_lanes1 >> sh) & 0xffffu) << 1);   // bit i = step i moves
            const unsigned int below = steps & ((1u << i_first) - 1u);                         // (1 <= i_first <= 29: the carry-in step)
            j = below ? 31 - __clz((int)below) : -1;
        } else {
            j = i_first - 1;
            while (j >= 0 && !(my[j].flags & LON_MOVING)) j--;
        }
        if (j >= 0) {
            double d_j, dv_j = 0.0, da_j;
            if (j < traj_len) lat_eval(j, 0.0, d_j, dv_j, da_j);
            Cy.th_prev = heading_of_moving_step(my[j], dv_j);
        }
    }
    StepAcc A;
    A.neg = A.acc_viol = A.collided = false;
    A.step_reasons = 0; A.first_key = 0xffffffffu; A.fail_step = 0x7fffffff; A.bound_step = 0x7fffffff;
    A.sum_abs_d = A.sum_voff = A.pred = A.d_end = A.v_end = 0.0;
    StepOut O;
    FX_GLOBAL double *__restrict__ planes = as_global(P.planes);
    const int64_t ps = (int64_t)S * ld;

    // hot obstacle table: staged per wave when the step index is wave-uniform (one lane per candidate or wave split);
    // the blocks sit behind the exchange block in dynamic LDS
    constexpr bool HOT = OBST && (G == 1 || WSPLIT);
    ObsHot Hs;
    Hs.lds = nullptr; Hs.tab = as_global(P.obs_hot); Hs.n_el = P.K * FX_HOT_STRIDE; Hs.lane = tid & 63;
    if (HOT && P.K > 0) {
        char *hot_base = lds_tail;   // over the knots' arc lengths, which only the prologue reads
        const size_t hot_block = (sizeof(double) * FX_HOT_STRIDE * (size_t)fuse.kmax() + 15) & ~(size_t)15;
        Hs.lds = reinterpret_cast<double *>(hot_base + (size_t)(tid >> 6) * hot_block);
        if (i_first < i_end) Hs.prefetch(i_first);
    }

#ifdef FX_CULL_STATS
    double cs_base = 0.0, cs_wd = 0.0, cs_wv = 0.0, cs_wp = 0.0;
    {
        const double tt = dt, tt2 = tt * tt, tt3 = tt2 * tt, tt4 = tt3 * tt, tt5 = tt4 * tt;
        for (int n = 0; n < P.n_cost; n++) {
            const int id = P.cost_id[n];
            const double w = P.cost_w[n];
            if (id == FX_COST_DISTANCE_TO_REFERENCE_PATH) cs_wd = w / S;
            if (id == FX_COST_VELOCITY_OFFSET) cs_wv = w;
            if (id == FX_COST_PREDICTION) cs_wp = w;
            if (id == FX_COST_LATERAL_JERK)
                cs_base += w * (36 * L.c3 * L.c3 * tt + 144 * L.c3 * L.c4 * tt2 + 240 * L.c3 * L.c5 * tt3 + 192 * L.c4 * L.c4 * tt3 +
                                720 * L.c4 * L.c5 * tt4 + 720 * L.c5 * L.c5 * tt5);
            if (id == FX_COST_LONGITUDINAL_JERK)
                cs_base += w * (36 * cl3 * cl3 * tt + 144 * cl3 * cl4 * tt2 + 240 * cl3 * cl5 * tt3 + 192 * cl4 * cl4 * tt3 +
                                720 * cl4 * cl5 * tt4 + 720 * cl5 * cl5 * tt5);
        }
    }
#endif
    // wave-uniform step index: scalar row base + the lane's 32-bit byte offset for the plane stores (the host refuses bundles
    // whose rows exceed 4 GiB)
    constexpr bool USTEP32 = (G == 1 || WSPLIT);
    FX_STAMP(3);
#pragma unroll 1
    for (int i = i_first; i < i_end; i++) {
        const bool emit = i >= i_begin;
        const LonRow r = my[i];
        if (LSTAGE && rec_staged)   // (workgroup-uniform: the obstacle records and masks of the step come from LDS)
            walk_step<OBST, false, false, false>(K, r, L, tp, i, traj_len, d_ext, emit, bundle && active && emit, planes + (int64_t)i * ld + g, 0u,
                                                 ps, Cy, A, O, (const double *)rec_lds, (const unsigned long long *)pm_lds,
                                                 (const unsigned long long *)(pm_lds + S), Bv, &Hs, -1, neigh, part > 0);
        else
        walk_step<OBST, (G == 1 || WSPLIT), HOT, (WPE >= 3 && OBST)>(K, r, L, tp, i, traj_len, d_ext, emit, bundle && active && emit,
                                                 USTEP32 ? planes + (int64_t)i * ld : planes + (int64_t)i * ld + g,
                                                 USTEP32 ? (uint32_t)g * 8u : 0u, ps, Cy, A, O, obs_rec, obs_pmask, obs_hmask, Bv,
                                                 &Hs, i + 1 < i_end ? i + 1 : -1, neigh, part > 0);
#ifdef FX_CULL_STATS
        {   // how many (wave, step) pairs lie behind the point where every lane of the wave is decided (production flag set), and
            // behind the point where every lane is decided or its cost so far already exceeds a bound (the winner's cost)
            const bool dead = !active || A.first_key != 0xffffffffu || A.neg || A.acc_viol;
            const double lb = cs_base + cs_wd * A.sum_abs_d + cs_wv * A.sum_voff + cs_wp * A.pred;
            const bool out = dead || lb > fx_probe_bound * 1.000001;
            const unsigned long long md = __ballot(dead), mo = __ballot(out);
            FX_CSTAT(6, 1); FX_CSTAT(5, md == ~0ULL); FX_CSTAT(7, __popcll(md)); FX_CSTAT(8, mo == ~0ULL); FX_CSTAT(9, __popcll(mo));
        }
#endif
    }

    FX_STAMP(4);
    WalkResult W;
    W.neg = A.neg; W.acc_viol = A.acc_viol; W.collided = A.collided;
    W.step_reasons = A.step_reasons; W.first_key = A.first_key; W.fail_step = A.fail_step; W.bound_step = A.bound_step;
    W.sum_abs_d = A.sum_abs_d; W.sum_voff = A.sum_voff; W.pred = A.pred; W.dto = 0.0; W.lane_off = 0.0; W.d_end = A.d_end; W.v_end = A.v_end;
    W.cl3 = cl3; W.cl4 = cl4; W.cl5 = cl5; W.ct3 = L.c3; W.ct4 = L.c4; W.ct5 = L.c5;
    // wave split: the exchange block sits behind the rows in dynamic LDS
    double *xch = reinterpret_cast<double *>(rows + (size_t)n_pairs_max * S);
    finish_candidate<G, BUNDLE, OBST, false, WSPLIT, MEGA>(P, Pg, W, g, active, part, i_begin, i_end, bundle, do_collision, dbg, D,
                                                           red_cost, red_idx, red_cnt, fuse, xch, CPB, cand_local);
}

template <int G, bool BUNDLE, bool OBST, int WPE, bool WSPLIT>
__global__ __launch_bounds__(FX_BLOCK, WPE) void fx_eval_grid_kernel(const DevProblem *__restrict__ probs,
                                                                         const FuseArgs fuse) {
    fx_eval_grid_body<G, BUNDLE, OBST, WPE, WSPLIT, false>(probs, fuse);
}
