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

namespace fxk {

struct alignas(16) LonRow {  // longitudinal quantities of one (pair, step); 128 B
    double s, sv, sa;        // s, clamped s_dot, s_ddot
    double th_ref, k_r, k_r_d;
    double px, py, nhx, nhy;  // foot point and unit normal (0 outside the projection domain)
    double r_sv, sv2, r_sv2;  // 1/s_dot, s_dot^2, 1/s_dot^2 (only meaningful when moving)
    double u1;                // s - s[0] (LOW_VEL_MODE lateral parameter)
    uint32_t flags;           // LON_* bits
    uint32_t pad0;
    double pad1;
};
static_assert(sizeof(LonRow) == 128, "LonRow must be 128 bytes");

enum : uint32_t { LON_NEG = 1u, LON_ACC = 2u, LON_MOVING = 4u, LON_INDOMAIN = 8u };

}  // namespace fxk

// Workgroup size is a launch parameter (64, 128 or 256 lanes): small grids run wave-sized workgroups so that the
// dispatcher balances the chip at wave granularity and nothing is staged that a single wave does not need; the
// host guarantees that the rows of a workgroup ((CPB + nD - 2) / nD + 1 pairs) fit its dynamic LDS.
template <int G, bool BUNDLE, bool OBST, int WPE>
__global__ __launch_bounds__(FX_BLOCK, WPE) void fx_eval_grid_kernel(const DevProblem *__restrict__ probs) {
    using namespace fxk;
    const int BLK = blockDim.x;
    const int CPB = BLK / G;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];  // [5][S] time powers | rows[n_pairs][S]
    __shared__ double red_cost[FX_BLOCK / 64];
    __shared__ long long red_idx[FX_BLOCK / 64];
    __shared__ unsigned int red_cnt[2 + FX_NUM_REASONS];

    const DevProblem &P = probs[blockIdx.y];
    const int tid = threadIdx.x;
    const int64_t C = P.C;
    if ((int64_t)blockIdx.x * CPB >= C) return;
    const int part = G == 1 ? 0 : (tid & (G - 1));
    const int64_t c0 = (int64_t)blockIdx.x * CPB;  // first local candidate of this workgroup
    const int64_t g_raw = c0 + (G == 1 ? tid : tid / G);
    const bool active = g_raw < C;
    const int64_t g = active ? g_raw : C - 1;

    const int M = P.M, S = P.S;
    const int nD = P.nD, nV = P.nV;
    {
        const FX_GLOBAL double *__restrict__ tsrc = as_global(P.tpow);
        for (int i = tid; i < 5 * S; i += BLK) lds_dyn[i] = tsrc[i];
    }
    if (tid < 2 + FX_NUM_REASONS) red_cnt[tid] = 0;
    __syncthreads();
    // the knots are only touched by the prologue (one lookup per (pair, step) item): read them through L1/L2
    const FX_GLOBAL double *__restrict__ kn = as_global(P.ref);
    auto knot_at = [&](int k) {
        const FX_GLOBAL double *q = kn + (int64_t)k * FX_REF_FIELDS;
        Knot r;
        r.pos = q[0]; r.theta = q[1]; r.curv = q[2]; r.curv_d = q[3]; r.x = q[4]; r.y = q[5]; r.nx = q[6]; r.ny = q[7];
        return r;
    };
    const double *__restrict__ tp = lds_dyn;
    LonRow *__restrict__ rows = reinterpret_cast<LonRow *>(lds_dyn + ((5 * S + 1) & ~1));  // 16-byte aligned

    const double dt = P.dt;
    const bool low_vel = P.low_vel_mode != 0;
    const bool D = (P.mode & FX_MODE_DRAW_TRAJ_SET) != 0;
    const bool dbg = D || (P.mode & FX_MODE_KINEMATIC_DEBUG) != 0;
    const bool do_collision = OBST && (P.mode & FX_MODE_COLLISION) != 0;
    const bool bundle = BUNDLE && (P.mode & FX_MODE_WRITE_BUNDLE) != 0;
    const double a_max = P.veh.a_max, kappa_max = P.veh.kappa_max, v_switch = P.veh.v_switch, v_des = P.v_des;
    const double av_switch = a_max * v_switch;
    const double wb = P.veh.wb_rear_axle, half_len = P.veh.length / 2, half_wid = P.veh.width / 2;
    const int64_t ld = P.ld;
    const double r_dt = 1.0 / dt;
    const double s0 = P.x0_lon[0], ss0 = P.x0_lon[1], sss0 = P.x0_lon[2];

    // ---- prologue: longitudinal table of the pairs this workgroup touches ----
    const int64_t gbase = P.g_base;
    const int64_t pair0 = (c0 + gbase) / nD;
    const int64_t c_last = min(c0 + CPB, C) - 1;
    const int n_pairs = (int)((c_last + gbase) / nD - pair0) + 1;
    const double rp_first = kn[0], rp_last = kn[(int64_t)(M - 1) * FX_REF_FIELDS];
    for (int item = tid; item < n_pairs * S; item += BLK) {
        const int pl = item / S, i = item - pl * S;
        const int64_t pair = pair0 + pl;
        const int it = (int)(pair / nV), iv = (int)(pair - (int64_t)it * nV);
        const double T = P.t_samp[it], v1 = as_global(P.v_samp)[iv];
        // longitudinal quartic (polynomial_trajectory.py:452-488)
        const double b1 = v1 - ss0 - sss0 * T, b2 = 0.0 - sss0, T2 = T * T;
        const double cl0 = s0, cl1 = ss0, cl2 = .5 * sss0;
        const double cl3 = (3.0 * b1 - T * b2) / (3.0 * T2);
        const double cl4 = (T * b2 - 2.0 * b1) / (4.0 * T2 * T);
        int traj_len = (int)ceil((T + dt) / dt);
        traj_len = traj_len > S ? S : (traj_len < 1 ? 1 : traj_len);
        const int ie = i < traj_len ? i : traj_len - 1;  // sample that is evaluated (last one feeds the extension)
        const double t1 = tp[ie], t2 = tp[S + ie], t3 = tp[2 * S + ie], t4 = tp[3 * S + ie];
        double s_i = cl0 + cl1 * t1 + cl2 * t2 + cl3 * t3 + cl4 * t4;
        double sv_i = cl1 + 2. * cl2 * t1 + 3. * cl3 * t2 + 4. * cl4 * t3;
        double sa_i = 2 * cl2 + 6 * cl3 * t1 + 12 * cl4 * t2;
        if (i >= traj_len) {  // s[i] = s[i-1] + dt * s_dot_end, one rounding per step as in the reference loop (:319-322)
            for (int j = traj_len; j <= i; j++) s_i = s_i + dt * sv_i;
            sa_i = 0.0;
        }
        LonRow r;
        r.flags = (sv_i < -FX_EPS ? LON_NEG : 0u) | (fabs(sa_i) > a_max ? LON_ACC : 0u);
        if (fabs(sv_i) < FX_EPS) sv_i = 0.0;
        const bool moving = sv_i > 0.001;
        if (moving) r.flags |= LON_MOVING;
        r.s = s_i; r.sv = sv_i; r.sa = sa_i;
        r.u1 = s_i - cl0;
        r.r_sv = 1.0 / sv_i;
        r.sv2 = sv_i * sv_i;
        r.r_sv2 = 1.0 / r.sv2;
        // reference segment: np.argmax(ref_pos > s) - 1 with Python's negative-index wrap (:415-420)
        int lo = 0, hi = M;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (kn[(int64_t)mid * FX_REF_FIELDS] > s_i) hi = mid; else lo = mid + 1;
        }
        const int ub = lo;
        const int i1 = ub == M ? 0 : ub;
        const int i0 = i1 == 0 ? M - 1 : i1 - 1;
        const Knot k0 = knot_at(i0), k1 = knot_at(i1);
        const double seg = k1.pos - k0.pos, r_seg = 1.0 / seg;
        const double s_lambda = div_rcp(s_i - k0.pos, seg, r_seg);
        r.th_ref = wrap_pm_2pi(div_rcp((k1.theta - k0.theta) * (s_i - k0.pos), seg, r_seg) + k0.theta);
        r.k_r = (k1.curv - k0.curv) * s_lambda + k0.curv;
        r.k_r_d = (k1.curv_d - k0.curv_d) * s_lambda + k0.curv_d;
        r.px = r.py = r.nhx = r.nhy = 0.0;
        if (s_i >= rp_first && s_i <= rp_last) {
            r.flags |= LON_INDOMAIN;
            int kk = ub - 1;
            kk = kk < 0 ? 0 : (kk > M - 2 ? M - 2 : kk);
            const Knot q0 = kk == i0 ? k0 : knot_at(kk), q1 = kk == i0 ? k1 : knot_at(kk + 1);
            const double lam = kk == i0 ? s_lambda : (s_i - q0.pos) / (q1.pos - q0.pos);
            r.px = q0.x + lam * (q1.x - q0.x);
            r.py = q0.y + lam * (q1.y - q0.y);
            const double nx = q0.nx + lam * (q1.nx - q0.nx), ny = q0.ny + lam * (q1.ny - q0.ny);
            const double nn = sqrt(nx * nx + ny * ny), r_nn = 1.0 / nn;
            r.nhx = div_rcp(nx, nn, r_nn);
            r.nhy = div_rcp(ny, nn, r_nn);
        }
        r.pad0 = 0; r.pad1 = 0.0;
        rows[item] = r;
    }
    __syncthreads();

    // ---- candidate: lateral quintic (reactive_planner.py:158-171) ----
    const int64_t gg = g + gbase;
    const int64_t pair = gg / nD;
    const int id = (int)(gg - pair * nD);
    const int it = (int)(pair / nV), iv = (int)(pair - (int64_t)it * nV);
    const double T = P.t_samp[it], v1 = P.v_samp[iv], d1 = as_global(P.d_samp)[id];
    const double d0 = P.x0_lat[0], dd0 = P.x0_lat[1], ddd0 = P.x0_lat[2];
    const LonRow *__restrict__ my = rows + (int)(pair - pair0) * S;
    double cl3, cl4;
    {
        const double b1 = v1 - ss0 - sss0 * T, b2 = 0.0 - sss0, T2 = T * T;
        cl3 = (3.0 * b1 - T * b2) / (3.0 * T2);
        cl4 = (T * b2 - 2.0 * b1) / (4.0 * T2 * T);
    }
    double tau = T;
    if (low_vel) {
        const double cl0 = s0, cl1 = ss0, cl2 = .5 * sss0;
        double t2 = T * T, t3 = t2 * T, t4 = t2 * t2;
        double s_lon_goal = (cl0 + cl1 * T + cl2 * t2 + cl3 * t3 + cl4 * t4) - s0;
        if (s_lon_goal <= 0) s_lon_goal = T;
        tau = s_lon_goal;
    }
    double ct0, ct1, ct2, ct3, ct4, ct5;
    {
        double T2 = tau * tau, T3 = T2 * tau, T4 = T3 * tau, T5 = T4 * tau;
        double b0 = d1 - d0 - dd0 * tau - .5 * ddd0 * T2;
        double b1 = 0.0 - dd0 - ddd0 * tau;
        double b2 = 0.0 - ddd0;
        ct0 = d0;
        ct1 = dd0;
        ct2 = .5 * ddd0;
        ct3 = (10.0 * b0 - 4.0 * b1 * tau + .5 * b2 * T2) / T3;
        ct4 = (-15.0 * b0 + 7.0 * b1 * tau - b2 * T2) / T4;
        ct5 = (6.0 * b0 - 3.0 * b1 * tau + .5 * b2 * T2) / T5;
    }
    int traj_len = (int)ceil((T + dt) / dt);
    traj_len = traj_len > S ? S : (traj_len < 1 ? 1 : traj_len);

    if (bundle && active && part == 0) {
        FX_GLOBAL double *__restrict__ co = as_global(P.coeffs) + g;
        co[0 * ld] = s0; co[1 * ld] = ss0; co[2 * ld] = .5 * sss0; co[3 * ld] = cl3; co[4 * ld] = cl4; co[5 * ld] = 0.0;
        co[6 * ld] = ct0; co[7 * ld] = ct1; co[8 * ld] = ct2; co[9 * ld] = ct3; co[10 * ld] = ct4; co[11 * ld] = ct5;
        as_global(P.traj_len)[g] = traj_len;
    }

    auto lat_at = [&](int i, double u_lowvel, double &d, double &dv, double &da) {
        double u1 = tp[i], u2 = tp[S + i], u3 = tp[2 * S + i], u4 = tp[3 * S + i], u5 = tp[4 * S + i];
        if (low_vel) { u1 = u_lowvel; u2 = u1 * u1; u3 = u2 * u1; u4 = u2 * u2; u5 = u4 * u1; }
        d = ct0 + ct1 * u1 + ct2 * u2 + ct3 * u3 + ct4 * u4 + ct5 * u5;
        dv = ct1 + 2. * ct2 * u1 + 3. * ct3 * u2 + 4. * ct4 * u3 + 5. * ct5 * u4;
        da = 2 * ct2 + 6 * ct3 * u1 + 12 * ct4 * u2 + 20 * ct5 * u3;
    };
    // lateral value the extension holds: d[traj_len-1] (reactive_planner.py:344)
    double d_ext, dv_u, da_u;
    lat_at(traj_len - 1, my[traj_len - 1].u1, d_ext, dv_u, da_u);

    // ---- this lane's chunk of the horizon ----
    const int CH = G == 1 ? S : (S + G - 1) / G;
    const int i_begin = part * CH;
    const int i_end = min(S, i_begin + CH);
    const int i_first = (G > 1 && part > 0) ? i_begin - 1 : i_begin;

    double th_prev = P.x0_orientation, kap_prev = 0.0;
    if (G > 1 && !low_vel && i_first > 0 && i_first < S && !(my[i_first].flags & LON_MOVING)) {
        // the carry-in step keeps the previous heading: scan back to the last moving step (:447)
        int j = i_first - 1;
        while (j >= 0 && !(my[j].flags & LON_MOVING)) j--;
        if (j >= 0) {
            double d_j, dv_j, da_j;
            if (j < traj_len) lat_at(j, 0.0, d_j, dv_j, da_j); else { dv_j = 0.0; }
            th_prev = fxm::atan(div_rcp(dv_j, my[j].sv, my[j].r_sv)) + my[j].th_ref;
        }
    }

    bool neg = false, acc_viol = false;
    uint32_t step_reasons = 0, first_key = 0xffffffffu;
    int fail_step = 0x7fffffff;
    double sum_abs_d = 0.0, sum_voff = 0.0, pred = 0.0, d_end = 0.0, v_end = 0.0;
    const int half = S / 2;
    bool collided = false;
    double bx_prev = 0.0, by_prev = 0.0, ux_prev = 0.0, uy_prev = 0.0;
    const int K = P.K, Pn = P.P;
    const FX_GLOBAL double *__restrict__ obs_pos = as_global(P.obs_pos);
    const FX_GLOBAL double *__restrict__ obs_cov_inv = as_global(P.obs_cov_inv);
    const FX_GLOBAL double *__restrict__ obs_hull = as_global(P.obs_hull);
    const FX_GLOBAL int32_t *__restrict__ obs_npred = as_global(P.obs_npred);
    const FX_GLOBAL int32_t *__restrict__ obs_nhull = as_global(P.obs_nhull);
    int max_nhull = 0;
    if (do_collision) for (int k = 0; k < K; k++) max_nhull = max(max_nhull, obs_nhull[k]);
    FX_GLOBAL double *__restrict__ planes = as_global(P.planes);

#pragma unroll 1
    for (int i = i_first; i < i_end; i++) {
        const bool emit = i >= i_begin;
        const LonRow r = my[i];
        const double s_i = r.s, sv_i = r.sv, sa_i = r.sa;
        double d_i, dv_i, da_i;
        if (i < traj_len) lat_at(i, r.u1, d_i, dv_i, da_i);
        else { d_i = d_ext; dv_i = 0.0; da_i = 0.0; }
        if (emit) {
            neg |= (r.flags & LON_NEG) != 0;
            acc_viol |= (r.flags & LON_ACC) != 0;
        }
        // -- d', d'' (reactive_planner.py:392-412) --
        const bool moving = (r.flags & LON_MOVING) != 0;
        double dp, dpp;
        if (!low_vel) {
            dp = moving ? div_rcp(dv_i, sv_i, r.r_sv) : 0.;
            const double ddot = da_i - dp * sa_i;
            dpp = moving ? div_rcp(ddot, r.sv2, r.r_sv2) : 0.;
        } else {
            dp = dv_i;
            dpp = da_i;
        }
        const double th_ref = r.th_ref, k_r = r.k_r, k_r_d = r.k_r_d;
        double th_cl, th_gl, cosTheta, tanTheta, secTheta;
        if (moving || low_vel) {
            th_cl = fxm::atan(dp);
            th_gl = th_cl + th_ref;
            secTheta = sqrt(1.0 + dp * dp);
            cosTheta = 1.0 / secTheta;
            tanTheta = dp;
        } else {  // standstill at high-speed mode keeps the previous global heading (:447-454)
            th_gl = th_prev;
            th_cl = th_gl - th_ref;
            double sinTheta;
            fxm::sincos(th_cl, &sinTheta, &cosTheta);
            secTheta = 1.0 / cosTheta;
            tanTheta = sinTheta * secTheta;
        }
        // -- global curvature, velocity, acceleration (:463-478) --
        const double oneKrD = (1 - k_r * d_i);
        const double cok = cosTheta / oneKrD;
        const double okc = oneKrD * secTheta;
        const double kap = (dpp + (k_r * dp + k_r_d * d_i) * tanTheta) * cosTheta * (cok * cok) + cok * k_r;
        const double v_i = sv_i * okc;
        const double a_i = sa_i * okc +
                           (r.sv2 * secTheta) * (oneKrD * tanTheta * (kap * okc - k_r) - (k_r_d * d_i + k_r * dp));
        // -- constraints (:480-533) --
        if (emit) {
            uint32_t hit = 0;
            if (v_i < -FX_EPS) hit |= 1u << 4;
            if (fabs(kap) > kappa_max) hit |= 1u << 5;
            const double yaw_rate = i > 0 ? div_rcp(th_gl - th_prev, dt, r_dt) : 0.;
            if (fabs(np_round5(yaw_rate)) > kappa_max * v_i) hit |= 1u << 6;
            const double kap_rate = i > 0 ? div_rcp(kap - kap_prev, dt, r_dt) : 0.;
            if (fabs(kap_rate) > 0.4) hit |= 1u << 7;
            const double a_hi = v_i > v_switch ? av_switch / v_i : a_max;
            if (!(-a_max <= a_i && a_i <= a_hi)) hit |= 1u << 8;
            if (dbg) step_reasons |= hit;
            else if (hit && first_key == 0xffffffffu) first_key = ((uint32_t)i << 4) | (uint32_t)(__ffs((int)hit) - 1);
        }
        const double kap_dot = i > 0 ? kap - kap_prev : 0.0;

        // -- (s, d) -> (x, y): foot point + d * unit normal, 0 from the first step outside the domain on (:537-547) --
        double x_i = 0.0, y_i = 0.0;
        if (!(r.flags & LON_INDOMAIN)) {
            if (emit && fail_step == 0x7fffffff) fail_step = i;
        } else if (fail_step == 0x7fffffff) {
            x_i = r.px + d_i * r.nhx;
            y_i = r.py + d_i * r.nhy;
        }

        if (bundle && active && emit) {
            FX_GLOBAL double *__restrict__ row = planes + (int64_t)i * ld + g;
            const int64_t ps = (int64_t)S * ld;
            row[FX_PL_X * ps] = x_i;
            row[FX_PL_Y * ps] = y_i;
            row[FX_PL_THETA * ps] = th_gl;
            row[FX_PL_V * ps] = v_i;
            row[FX_PL_A * ps] = a_i;
            row[FX_PL_KAPPA * ps] = kap;
            row[FX_PL_KAPPA_DOT * ps] = kap_dot;
            row[FX_PL_S * ps] = s_i;
            row[FX_PL_D * ps] = d_i;
            row[FX_PL_THETA_CL * ps] = th_cl;
            row[FX_PL_S_DOT * ps] = sv_i;
            row[FX_PL_S_DDOT * ps] = sa_i;
            row[FX_PL_D_DOT * ps] = dv_i;
            row[FX_PL_D_DDOT * ps] = da_i;
        }

        if (emit) {
            sum_abs_d += fabs(d_i);
            if (i >= half && i < S - 1) sum_voff += fabs(v_i - v_des);
            if (i == S - 1) { d_end = d_i; v_end = v_i; }
        }
        if (OBST) {
            if (emit && i >= 1) {
                for (int k = 0; k < K; k++) {
                    if (i < obs_npred[k]) {
                        const FX_GLOBAL double *__restrict__ mu = obs_pos + ((int64_t)k * Pn + (i - 1)) * 2;
                        const FX_GLOBAL double *__restrict__ iv = obs_cov_inv + ((int64_t)k * Pn + (i - 1)) * 4;
                        const double e0 = x_i - mu[0], e1 = y_i - mu[1];
                        const double r0 = e0 * iv[0] + e1 * iv[2], r1 = e0 * iv[1] + e1 * iv[3];
                        const double m = r0 * e0 + r1 * e1;
                        pred += 1.0 / (m * m);
                    }
                }
            }
            if (do_collision) {
                const bool need = (i >= 2 ? i - 2 : 0) < max_nhull;
                if (need && i >= 1) {
                    double su, cu;
                    fxm::sincos(th_gl, &su, &cu);
                    const double bx = x_i + wb * cu, by = y_i + wb * su;
                    if (emit && i >= 2) {
                        const Obb hull = obb_hull(bx_prev, by_prev, ux_prev, uy_prev, bx, by, cu, su, half_len, half_wid);
                        for (int k = 0; k < K; k++) {
                            if (i - 2 < obs_nhull[k]) {
                                const FX_GLOBAL double *__restrict__ oh = obs_hull + ((int64_t)k * (Pn - 1) + (i - 2)) * 6;
                                collided |= obb_overlap(hull, oh);
                            }
                        }
                    }
                    bx_prev = bx; by_prev = by; ux_prev = cu; uy_prev = su;
                }
            }
        }
        th_prev = th_gl;
        kap_prev = kap;
    }

    WalkResult W;
    W.neg = neg; W.acc_viol = acc_viol; W.collided = collided;
    W.step_reasons = step_reasons; W.first_key = first_key; W.fail_step = fail_step;
    W.sum_abs_d = sum_abs_d; W.sum_voff = sum_voff; W.pred = pred; W.dto = 0.0; W.d_end = d_end; W.v_end = v_end;
    W.cl3 = cl3; W.cl4 = cl4; W.ct3 = ct3; W.ct4 = ct4; W.ct5 = ct5;
    finish_candidate<G, BUNDLE, OBST, false>(P, W, g, active, part, i_begin, i_end, bundle, do_collision, dbg, D, red_cost,
                                             red_idx, red_cnt);
}
