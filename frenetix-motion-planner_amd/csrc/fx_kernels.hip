// fx_kernels.hip -- the fused Frenet sampling-and-evaluation pipeline for gfx950 (MI355X).
//
// One lane evaluates one candidate trajectory end to end, streaming over the horizon:
//   sampling index -> (T, v1, d1)                       reactive_planner.py:149-158 / sampling_matrix.py:85-121
//   quartic / quintic coefficients (closed form)        polynomial_trajectory.py:293-343, :452-488
//   s, s', s'', d, d', d'' on the reference time grid   reactive_planner.py:295-346
//   Frenet -> Cartesian kinematics                      reactive_planner.py:389-478
//   five kinematic constraints -> reason bits           reactive_planner.py:480-533
//   (s, d) -> (x, y) along the reference polyline       utils_coordinate_system.py:263-270 (CCosy)
//   partial costs, weighted sum                         partial_cost_functions.py, cost_function.py:78-91
//   OBB-sum hull + SAT vs predicted obstacle hulls      planner.py:342-357, collision_check.py:110-200
//   (cost, index) arg-min over the workgroup            trajectories.py:560 + planner.py:336-390
//
// Why lane-per-candidate: every quantity of step i depends on step i-1 of the same candidate (theta/kappa
// finite differences, horizon extension, standstill carry), and nothing depends on another candidate.  A lane
// that walks its own horizon needs no cross-lane traffic, all 64 lanes of a wave are busy for any horizon
// length, and every SoA plane store is a 512-byte contiguous row segment (plane[p][step][candidate]).
// Reference knots (64 B per knot, AoS) are staged once per workgroup in LDS -- the only divergent reads.
// Everything indexed by (step) or (obstacle, step) is wave-uniform and comes in through scalar loads.
//
// FP64 throughout, compiled with -ffp-contract=off: the expression trees mirror the NumPy expressions of the
// reference term by term (the CPU oracle does the same), so GPU and oracle differ only in libm (OCML vs glibc)
// and in the order of the long cost sums (the device accumulates in step order, NumPy pairwise).
#include "fx_device.h"

#define FX_EPS 1e-5
#define FX_TWO_PI 6.283185307179586

namespace {

__device__ __forceinline__ double wrap_pm_2pi(double a) {
    // commonroad make_valid_orientation (restated): bring into [-2pi, 2pi]
    while (a > FX_TWO_PI) a -= FX_TWO_PI;
    while (a < -FX_TWO_PI) a += FX_TWO_PI;
    return a;
}

__device__ __forceinline__ double np_round5(double x) { return rint(x * 1e5) / 1e5; }

struct Knot {  // one reference knot as staged in LDS
    double pos, theta, curv, curv_d, x, y, nx, ny;
};

// streaming composite Simpson (scipy.integrate.simpson, equal spacing) over y_0..y_{n-1}
struct Simpson {
    double acc, ym1, ym2, ym3;
    int n;
    __device__ __forceinline__ void init() { acc = 0.0; ym1 = ym2 = ym3 = 0.0; n = 0; }
    __device__ __forceinline__ void push(double y, int n_total) {
        // the basic 1-4-1 rule runs over the first n_total (odd) or n_total-1 (even) samples
        const int n_basic = (n_total & 1) ? n_total : n_total - 1;
        if (n >= 2 && !(n & 1) && n < n_basic) acc += (ym2 + 4.0 * ym1) + y;
        ym3 = ym2;
        ym2 = ym1;
        ym1 = y;
        n++;
    }
    // corr = (alpha, beta, eta): scipy's last-interval correction for an even sample count
    __device__ __forceinline__ double finish(int n_total, double dx, const double *__restrict__ corr) const {
        if (n_total & 1) return (0.0 + acc) * (dx / 3.0);
        if (n_total == 2) return 0.0 + 0.5 * dx * (ym1 + ym2);
        double r = (0.0 + acc) * (dx / 3.0);
        r += (corr[0] * ym1 + corr[1] * ym2) - corr[2] * ym3;
        return r + 0.0;
    }
};

// OBB hull of two boxes (centre, unit heading, half extents hl x hw): heading = normalised sum of headings,
// extents = tight range of both boxes on the hull axes (DESIGN.md "OBB-sum hull").
struct Obb {
    double cx, cy, ex, ey, h1, h2;
};

__device__ __forceinline__ Obb obb_hull(double c0x, double c0y, double u0x, double u0y, double c1x, double c1y,
                                        double u1x, double u1y, double hl, double hw) {
    double mx = u0x + u1x, my = u0y + u1y;
    double mn = sqrt(mx * mx + my * my);
    double ex, ey;
    if (mn < 1e-12) { ex = u0x; ey = u0y; } else { ex = mx / mn; ey = my / mn; }
    double fx = -ey, fy = ex;
    double p1 = c0x * ex + c0y * ey, p2 = c0x * fx + c0y * fy;
    double r1 = hl * fabs(u0x * ex + u0y * ey) + hw * fabs(-u0y * ex + u0x * ey);
    double r2 = hl * fabs(u0x * fx + u0y * fy) + hw * fabs(-u0y * fx + u0x * fy);
    double lo1 = p1 - r1, hi1 = p1 + r1, lo2 = p2 - r2, hi2 = p2 + r2;
    p1 = c1x * ex + c1y * ey;
    p2 = c1x * fx + c1y * fy;
    r1 = hl * fabs(u1x * ex + u1y * ey) + hw * fabs(-u1y * ex + u1x * ey);
    r2 = hl * fabs(u1x * fx + u1y * fy) + hw * fabs(-u1y * fx + u1x * fy);
    lo1 = fmin(lo1, p1 - r1); hi1 = fmax(hi1, p1 + r1);
    lo2 = fmin(lo2, p2 - r2); hi2 = fmax(hi2, p2 + r2);
    double m1 = 0.5 * (lo1 + hi1), m2 = 0.5 * (lo2 + hi2);
    Obb o;
    o.cx = m1 * ex + m2 * fx;
    o.cy = m1 * ey + m2 * fy;
    o.ex = ex;
    o.ey = ey;
    o.h1 = 0.5 * (hi1 - lo1);
    o.h2 = 0.5 * (hi2 - lo2);
    return o;
}

// separating-axis test; b = (cx, cy, ex, ey, h1, h2) read through wave-uniform (scalar) loads
__device__ __forceinline__ bool obb_overlap(const Obb &a, const double *__restrict__ b) {
    double tx = b[0] - a.cx, ty = b[1] - a.cy;
    double c = a.ex * b[2] + a.ey * b[3];
    double s = a.ex * b[3] - a.ey * b[2];
    double ac = fabs(c), as = fabs(s);
    bool sep = fabs(tx * a.ex + ty * a.ey) > a.h1 + (b[4] * ac + b[5] * as);
    sep |= fabs(-tx * a.ey + ty * a.ex) > a.h2 + (b[4] * as + b[5] * ac);
    sep |= fabs(tx * b[2] + ty * b[3]) > b[4] + (a.h1 * ac + a.h2 * as);
    sep |= fabs(-tx * b[3] + ty * b[2]) > b[5] + (a.h1 * as + a.h2 * ac);
    return !sep;
}

__device__ __forceinline__ int wave_count(bool p) { return __popcll(__ballot(p)); }

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Main kernel.  grid = (ceil(maxC / 256), n_agents), block = 256, dynamic LDS = M * 64 B.
//   BUNDLE : write the 14-plane SoA TrajectoryBundle + coefficients (FX_MODE_WRITE_BUNDLE)
//   OBST   : obstacles present (prediction cost and/or collision stage)
//   EXTRA  : any of the Simpson / distance_to_obstacles costs active (more live state per lane)
// ---------------------------------------------------------------------------------------------------
template <bool BUNDLE, bool OBST, bool EXTRA>
__global__ __launch_bounds__(FX_BLOCK) void fx_eval_kernel(const DevProblem *__restrict__ probs) {
    extern __shared__ double lds_ref[];  // [M][8]
    __shared__ double red_cost[FX_BLOCK / 64];
    __shared__ long long red_idx[FX_BLOCK / 64];
    __shared__ unsigned int red_cnt[2 + FX_NUM_REASONS];

    const DevProblem &P = probs[blockIdx.y];
    const int tid = threadIdx.x;
    const int64_t C = P.C;
    const int64_t g_raw = (int64_t)blockIdx.x * FX_BLOCK + tid;
    if ((int64_t)blockIdx.x * FX_BLOCK >= C) return;  // whole workgroup beyond this agent's grid
    const bool active = g_raw < C;
    const int64_t g = active ? g_raw : C - 1;

    const int M = P.M;
    {
        const double *__restrict__ src = P.ref;
        for (int i = tid; i < M * FX_REF_FIELDS; i += FX_BLOCK) lds_ref[i] = src[i];
    }
    if (tid < 2 + FX_NUM_REASONS) red_cnt[tid] = 0;
    __syncthreads();
    const Knot *__restrict__ knots = reinterpret_cast<const Knot *>(lds_ref);

    const int S = P.S;
    const double dt = P.dt;
    const bool low_vel = P.low_vel_mode != 0;
    const bool D = (P.mode & FX_MODE_DRAW_TRAJ_SET) != 0;
    const bool KD = (P.mode & FX_MODE_KINEMATIC_DEBUG) != 0;
    const bool dbg = D || KD;
    const bool do_collision = OBST && (P.mode & FX_MODE_COLLISION) != 0;
    // a batch launch is specialised for the union of its agents' modes; each agent still honours its own
    const bool bundle = BUNDLE && (P.mode & FX_MODE_WRITE_BUNDLE) != 0;
    const double a_max = P.veh.a_max, kappa_max = P.veh.kappa_max, v_switch = P.veh.v_switch;
    const double v_des = P.v_des;
    const int64_t ld = P.ld;

    // ---- candidate parameters (reactive_planner.py:149-171 / sampling matrix row) ----
    double T, s0, ss0, sss0, v1, a1, d0, dd0, ddd0, d1, dd1, ddd1;
    if (P.has_matrix) {
        const double *__restrict__ r = P.matrix + 13 * (g + P.g_base);
        T = r[1] - r[0];
        s0 = r[2]; ss0 = r[3]; sss0 = r[4]; v1 = r[5]; a1 = r[6];
        d0 = r[7]; dd0 = r[8]; ddd0 = r[9]; d1 = r[10]; dd1 = r[11]; ddd1 = r[12];
    } else {
        const int nD = P.nD, nV = P.nV;
        const int64_t gg = g + P.g_base;
        const int64_t q = gg / nD;
        const int id = (int)(gg - q * nD);
        const int it = (int)(q / nV);
        const int iv = (int)(q - (int64_t)it * nV);
        T = P.t_samp[it];
        v1 = P.v_samp[iv];
        d1 = P.d_samp[id];
        s0 = P.x0_lon[0]; ss0 = P.x0_lon[1]; sss0 = P.x0_lon[2];
        d0 = P.x0_lat[0]; dd0 = P.x0_lat[1]; ddd0 = P.x0_lat[2];
        a1 = 0.0; dd1 = 0.0; ddd1 = 0.0;
    }

    // ---- longitudinal quartic (polynomial_trajectory.py:452-488, closed form of the 2x2 solve) ----
    double cl0, cl1, cl2, cl3, cl4;
    {
        double b1 = v1 - ss0 - sss0 * T;
        double b2 = a1 - sss0;
        double T2 = T * T;
        cl0 = s0;
        cl1 = ss0;
        cl2 = .5 * sss0;
        cl3 = (3.0 * b1 - T * b2) / (3.0 * T2);
        cl4 = (T * b2 - 2.0 * b1) / (4.0 * T2 * T);
    }
    // ---- lateral quintic over time (high speed) or arclength (LOW_VEL_MODE), reactive_planner.py:161-171 ----
    double tau = T;
    if (low_vel) {
        double t2 = T * T, t3 = t2 * T, t4 = t2 * t2;
        double s_lon_goal = (cl0 + cl1 * T + cl2 * t2 + cl3 * t3 + cl4 * t4) - s0;
        if (s_lon_goal <= 0) s_lon_goal = T;
        tau = s_lon_goal;
    }
    double ct0, ct1, ct2, ct3, ct4, ct5;
    {
        double T2 = tau * tau, T3 = T2 * tau, T4 = T3 * tau, T5 = T4 * tau;
        double b0 = d1 - d0 - dd0 * tau - .5 * ddd0 * T2;
        double b1 = dd1 - dd0 - ddd0 * tau;
        double b2 = ddd1 - ddd0;
        ct0 = d0;
        ct1 = dd0;
        ct2 = .5 * ddd0;
        ct3 = (10.0 * b0 - 4.0 * b1 * tau + .5 * b2 * T2) / T3;
        ct4 = (-15.0 * b0 + 7.0 * b1 * tau - b2 * T2) / T4;
        ct5 = (6.0 * b0 - 3.0 * b1 * tau + .5 * b2 * T2) / T5;
    }
    // len(np.arange(0, T+dt, dt)) (reactive_planner.py:296,303), clamped to the horizon
    int traj_len = (int)ceil((T + dt) / dt);
    traj_len = traj_len > S ? S : (traj_len < 1 ? 1 : traj_len);

    if (bundle && active) {
        double *__restrict__ co = P.coeffs + g;
        co[0 * ld] = cl0; co[1 * ld] = cl1; co[2 * ld] = cl2; co[3 * ld] = cl3; co[4 * ld] = cl4; co[5 * ld] = 0.0;
        co[6 * ld] = ct0; co[7 * ld] = ct1; co[8 * ld] = ct2; co[9 * ld] = ct3; co[10 * ld] = ct4; co[11 * ld] = ct5;
        P.traj_len[g] = traj_len;
    }

    // ---- streaming state ----
    const double *__restrict__ tp = P.tpow;
    const double rp_first = knots[0].pos, rp_last = knots[M - 1].pos;
    double s_first = 0.0, s_prev = 0.0, sv_last = 0.0, d_last = 0.0;
    double th_prev = 0.0, kap_prev = 0.0;
    bool neg = false, acc_viol = false, proj_ok = true;
    uint32_t step_reasons = 0;
    int ub = 0;  // first knot index with pos > s (upper bound), carried between steps
    // cost accumulators
    double sum_abs_d = 0.0, sum_voff = 0.0, pred = 0.0, dto = 0.0, d_end = 0.0, v_end = 0.0;
    const int half = S / 2;
    Simpson sim_acc, sim_jerk, sim_orient, sim_path;
    double a_prev = 0.0, thcl_prev = 0.0;
    if (EXTRA) { sim_acc.init(); sim_jerk.init(); sim_orient.init(); sim_path.init(); }
    // collision state
    bool collided = false;
    double bx_prev = 0.0, by_prev = 0.0, ux_prev = 0.0, uy_prev = 0.0;
    const double wb = P.veh.wb_rear_axle, hl = P.veh.length / 2, hw = P.veh.width / 2;
    const int K = P.K, Pn = P.P, n_dto = EXTRA ? P.n_dto : 0;
    const double *__restrict__ obs_pos = P.obs_pos;
    const double *__restrict__ obs_cov_inv = P.obs_cov_inv;
    const double *__restrict__ obs_hull = P.obs_hull;
    const int32_t *__restrict__ obs_npred = P.obs_npred;
    const int32_t *__restrict__ obs_nhull = P.obs_nhull;
    const double *__restrict__ dto_pos = P.dto_pos;
    int max_nhull = 0;
    if (do_collision) for (int k = 0; k < K; k++) max_nhull = max(max_nhull, obs_nhull[k]);

    double *__restrict__ planes = P.planes;

    for (int i = 0; i < S; i++) {
        // -- polynomials on the rounded time grid, horizon extension (reactive_planner.py:313-346) --
        const double t1 = tp[i], t2 = tp[S + i], t3 = tp[2 * S + i], t4 = tp[3 * S + i], t5 = tp[4 * S + i];
        double s_i, sv_i, sa_i, d_i, dv_i, da_i;
        if (i < traj_len) {
            s_i = cl0 + cl1 * t1 + cl2 * t2 + cl3 * t3 + cl4 * t4;
            sv_i = cl1 + 2. * cl2 * t1 + 3. * cl3 * t2 + 4. * cl4 * t3;
            sa_i = 2 * cl2 + 6 * cl3 * t1 + 12 * cl4 * t2;
            sv_last = sv_i;
            if (i == 0) s_first = s_i;
            double u1 = t1, u2 = t2, u3 = t3, u4 = t4, u5 = t5;
            if (low_vel) {
                u1 = s_i - s_first; u2 = u1 * u1; u3 = u2 * u1; u4 = u2 * u2; u5 = u4 * u1;
            }
            d_i = ct0 + ct1 * u1 + ct2 * u2 + ct3 * u3 + ct4 * u4 + ct5 * u5;
            dv_i = ct1 + 2. * ct2 * u1 + 3. * ct3 * u2 + 4. * ct4 * u3 + 5. * ct5 * u4;
            da_i = 2 * ct2 + 6 * ct3 * u1 + 12 * ct4 * u2 + 20 * ct5 * u3;
            d_last = d_i;
        } else {
            s_i = s_prev + dt * sv_last;
            sv_i = sv_last;
            sa_i = 0.0;
            d_i = d_last;
            dv_i = 0.0;
            da_i = 0.0;
        }
        s_prev = s_i;
        // -- validity, clamp, pre-filter (reactive_planner.py:350-355, :375) --
        neg |= sv_i < -FX_EPS;
        if (fabs(sv_i) < FX_EPS) sv_i = 0.0;
        acc_viol |= fabs(sa_i) > a_max;

        // -- d', d'' (reactive_planner.py:392-412) --
        double dp, dpp;
        const bool moving = sv_i > 0.001;
        if (!low_vel) {
            dp = moving ? dv_i / sv_i : 0.;
            double ddot = da_i - dp * sa_i;
            dpp = moving ? ddot / (sv_i * sv_i) : 0.;
        } else {
            dp = dv_i;
            dpp = da_i;
        }
        // -- reference segment: np.argmax(ref_pos > s) - 1 with Python's negative-index wrap (:415-420) --
        if (i == 0) {
            int lo = 0, hi = M;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (knots[mid].pos > s_i) hi = mid; else lo = mid + 1;
            }
            ub = lo;
        } else {
            while (ub < M && knots[ub].pos <= s_i) ub++;
            while (ub > 0 && knots[ub - 1].pos > s_i) ub--;
        }
        const int i1 = ub == M ? 0 : ub;
        const int i0 = i1 == 0 ? M - 1 : i1 - 1;
        const Knot k0 = knots[i0], k1 = knots[i1];
        const double s_lambda = (s_i - k0.pos) / (k1.pos - k0.pos);
        // interpolate_angle (utils_coordinate_system.py:137-155)
        const double th_ref = wrap_pm_2pi((k1.theta - k0.theta) * (s_i - k0.pos) / (k1.pos - k0.pos) + k0.theta);
        double th_cl, th_gl;
        if (moving || low_vel) {
            th_cl = atan2(dp, 1.0);
            th_gl = th_cl + th_ref;
        } else {  // standstill at high-speed mode keeps the previous global heading (:447-454)
            th_gl = i == 0 ? P.x0_orientation : th_prev;
            th_cl = th_gl - th_ref;
        }
        const double k_r = (k1.curv - k0.curv) * s_lambda + k0.curv;
        const double k_r_d = (k1.curv_d - k0.curv_d) * s_lambda + k0.curv_d;
        // -- global curvature, velocity, acceleration (:463-478) --
        const double oneKrD = (1 - k_r * d_i);
        const double cosTheta = cos(th_cl);
        const double tanTheta = tan(th_cl);
        const double cok = cosTheta / oneKrD;
        const double kap = (dpp + (k_r * dp + k_r_d * d_i) * tanTheta) * cosTheta * (cok * cok) + cok * k_r;
        const double v_i = sv_i * (oneKrD / cosTheta);
        const double a_i = sa_i * (oneKrD / cosTheta) +
                           ((sv_i * sv_i) / cosTheta) *
                               (oneKrD * tanTheta * (kap * (oneKrD / cosTheta) - k_r) - (k_r_d * d_i + k_r * dp));
        // -- constraints (:480-533): bit r = reason r --
        uint32_t hit = 0;
        if (v_i < -FX_EPS) hit |= 1u << 4;
        if (fabs(kap) > kappa_max) hit |= 1u << 5;
        const double yaw_rate = i > 0 ? (th_gl - th_prev) / dt : 0.;
        if (fabs(np_round5(yaw_rate)) > kappa_max * v_i) hit |= 1u << 6;
        const double kap_rate = i > 0 ? (kap - kap_prev) / dt : 0.;
        if (fabs(kap_rate) > 0.4) hit |= 1u << 7;
        const double a_hi = v_i > v_switch ? a_max * v_switch / v_i : a_max;
        if (!(-a_max <= a_i && a_i <= a_hi)) hit |= 1u << 8;
        if (dbg) step_reasons |= hit;
        else if (step_reasons == 0 && hit) step_reasons = hit & (~hit + 1u);  // first violated check only (break)
        const double kap_dot = i > 0 ? kap - kap_prev : 0.0;  // np.append([0], np.diff(kappa_gl)) (:552)

        // -- (s, d) -> (x, y) (:537-547; normative projection) --
        double x_i = 0.0, y_i = 0.0;
        if (proj_ok) {
            if (!(s_i >= rp_first && s_i <= rp_last)) {
                proj_ok = false;
            } else {
                int kk = ub - 1;
                kk = kk < 0 ? 0 : (kk > M - 2 ? M - 2 : kk);
                const Knot q0 = knots[kk], q1 = knots[kk + 1];
                const double lam = (s_i - q0.pos) / (q1.pos - q0.pos);
                const double px = q0.x + lam * (q1.x - q0.x), py = q0.y + lam * (q1.y - q0.y);
                const double nx = q0.nx + lam * (q1.nx - q0.nx), ny = q0.ny + lam * (q1.ny - q0.ny);
                const double nn = sqrt(nx * nx + ny * ny);
                x_i = px + d_i * (nx / nn);
                y_i = py + d_i * (ny / nn);
            }
        }

        // -- SoA bundle (trajectories.py:56-334) --
        if (bundle && active) {
            double *__restrict__ row = planes + (int64_t)i * ld + g;
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

        // -- partial costs, streamed --
        sum_abs_d += fabs(d_i);                                         // partial_cost_functions.py:166-167
        if (i >= half && i < S - 1) sum_voff += fabs(v_i - v_des);      // :125-127
        if (i == S - 1) { d_end = d_i; v_end = v_i; }
        if (EXTRA) {
            sim_acc.push(a_i * a_i, S);                                 // :29-31
            sim_path.push(v_i, S);                                      // :194-195
            if (i > 0) {
                const double j = (a_i - a_prev) / dt;                   // :41-44
                const double w = (th_cl - thcl_prev) / dt;              // :146-149
                sim_jerk.push(j * j, S - 1);
                sim_orient.push(w * w, S - 1);
            }
            a_prev = a_i;
            thcl_prev = th_cl;
            for (int o = 0; o < n_dto; o++) {                           // :177-184
                const double ex = x_i - dto_pos[2 * o], ey = y_i - dto_pos[2 * o + 1];
                const double dist = sqrt(ex * ex + ey * ey);
                dto += 1.0 / (dist * dist);
            }
        }
        if (OBST) {
            if (i >= 1) {  // ego step i pairs with prediction i-1 (collision_probability.py:283-292)
                for (int k = 0; k < K; k++) {
                    if (i < obs_npred[k]) {
                        const double *__restrict__ mu = obs_pos + ((int64_t)k * Pn + (i - 1)) * 2;
                        const double *__restrict__ iv = obs_cov_inv + ((int64_t)k * Pn + (i - 1)) * 4;
                        const double e0 = x_i - mu[0], e1 = y_i - mu[1];
                        const double r0 = e0 * iv[0] + e1 * iv[2], r1 = e0 * iv[1] + e1 * iv[3];
                        const double m = r0 * e0 + r1 * e1;
                        pred += 1.0 / (m * m);
                    }
                }
            }
            if (do_collision) {
                // ego box: centre = rear axle + wb_rear_axle along heading (state.py:30-39), heading theta_gl
                // wave-uniform: some obstacle hull exists at time index i-1 (this step's pair) or later
                const bool need = (i >= 2 ? i - 2 : 0) < max_nhull;
                if (need && i >= 1) {
                    double su, cu;
                    sincos(th_gl, &su, &cu);
                    const double bx = x_i + wb * cu, by = y_i + wb * su;
                    if (i >= 2) {
                        // OBB-sum hull of ego boxes (i-1, i) lives at time index i-1 and meets obstacle hull i-2
                        const Obb hull = obb_hull(bx_prev, by_prev, ux_prev, uy_prev, bx, by, cu, su, hl, hw);
                        for (int k = 0; k < K; k++) {
                            if (i - 2 < obs_nhull[k]) {
                                const double *__restrict__ oh = obs_hull + ((int64_t)k * (Pn - 1) + (i - 2)) * 6;
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

    // ---- flags: return-list membership and reasons exactly as check_feasibility assembles them ----
    uint32_t flags = FX_FLAG_VALID | FX_FLAG_FEASIBLE;
    uint32_t reasons = 0;
    bool done = false;
    if (neg) {
        flags &= ~FX_FLAG_VALID;
        reasons |= 1u << 10;
        if (!dbg) done = true;  // dropped: `continue` at :353-354
    }
    if (!done && !D) {
        if (acc_viol) { flags &= ~FX_FLAG_FEASIBLE; reasons |= 1u << 1; flags |= FX_FLAG_RETURNED; done = true; }
        else if (neg) { flags &= ~FX_FLAG_FEASIBLE; reasons |= 1u << 2; flags |= FX_FLAG_RETURNED; done = true; }
    }
    if (!done) {
        reasons |= step_reasons;
        if (step_reasons) flags &= ~FX_FLAG_FEASIBLE;
        if ((flags & FX_FLAG_FEASIBLE) || D) {
            if (!proj_ok) { flags &= ~FX_FLAG_VALID; reasons |= 1u << 9; }
            flags |= FX_FLAG_RETURNED;
        }
    }
    bool costed, selectable;
    if (D) {
        costed = (flags & FX_FLAG_RETURNED) != 0;
        selectable = costed && (flags & FX_FLAG_FEASIBLE);
    } else {
        costed = (flags & FX_FLAG_RETURNED) && (flags & FX_FLAG_VALID) && (flags & FX_FLAG_FEASIBLE);
        selectable = costed;
    }
    if (costed) flags |= FX_FLAG_COSTED;
    if (selectable) flags |= FX_FLAG_SELECTABLE;
    if (selectable && do_collision && collided) flags |= FX_FLAG_COLLISION;
    flags |= reasons << FX_REASON_SHIFT;

    // ---- weighted cost sum in name-sorted order (cost_function.py:78-91) ----
    double total = 0.0;
    {
        const double tt = dt, tt2 = tt * tt, tt3 = tt2 * tt, tt4 = tt3 * tt, tt5 = tt4 * tt;
        const int n_cost = P.n_cost;
        double sum = -0.0;
        for (int n = 0; n < n_cost; n++) {
            double c = 0.0;
            switch (P.cost_id[n]) {
            case FX_COST_DISTANCE_TO_REFERENCE_PATH: c = ((0.0 + sum_abs_d) + fabs(d_end) * 5) / S; break;
            case FX_COST_LATERAL_JERK:  // squared_jerk_integral(dt) (polynomial_trajectory.py:172-191, cost :54)
                c = (36 * ct3 * ct3 * tt + 144 * ct3 * ct4 * tt2 + 240 * ct3 * ct5 * tt3 + 192 * ct4 * ct4 * tt3 +
                     720 * ct4 * ct5 * tt4 + 720 * ct5 * ct5 * tt5);
                break;
            case FX_COST_LONGITUDINAL_JERK:
                c = (36 * cl3 * cl3 * tt + 144 * cl3 * cl4 * tt2 + 240 * cl3 * 0.0 * tt3 + 192 * cl4 * cl4 * tt3 +
                     720 * cl4 * 0.0 * tt4 + 720 * 0.0 * 0.0 * tt5);
                break;
            case FX_COST_VELOCITY_OFFSET: {
                const double e = v_end - v_des;
                c = (0.0 + sum_voff) + fabs(e * e);
                break;
            }
            case FX_COST_PREDICTION: c = OBST ? pred : 0.0; break;
            case FX_COST_ACCELERATION: c = EXTRA ? sim_acc.finish(S, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_PATH_LENGTH: c = EXTRA ? sim_path.finish(S, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_JERK: c = EXTRA ? sim_jerk.finish(S - 1, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_ORIENTATION_OFFSET: c = EXTRA ? sim_orient.finish(S - 1, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_DISTANCE_TO_OBSTACLES: c = EXTRA ? dto : 0.0; break;
            default: break;
            }
            if ((P.mode & FX_MODE_WRITE_COSTMAP) && active) P.costmap[(int64_t)n * ld + g] = costed ? c : 0.0;
            sum += P.cost_w[n] * c;
        }
        total = 0.0 + sum;
    }
    if (active) {
        P.cost[g] = costed ? total : 0.0;
        P.flags[g] = flags;
    }

    // ---- workgroup reductions: counters and the (cost, index) arg-min partial ----
    const int lane = tid & 63, wave = tid >> 6;
    {
        const bool ret = active && (flags & FX_FLAG_RETURNED);
        const int n_ret = wave_count(ret);
        const int n_feas = wave_count(ret && (flags & FX_FLAG_VALID) && (flags & FX_FLAG_FEASIBLE));
        if (lane == 0) {
            if (n_ret) atomicAdd(&red_cnt[0], (unsigned)n_ret);
            if (n_feas) atomicAdd(&red_cnt[1], (unsigned)n_feas);
        }
        for (int r = 0; r < FX_NUM_REASONS; r++) {
            const int n = wave_count(active && ((reasons >> r) & 1u));
            if (lane == 0 && n) atomicAdd(&red_cnt[2 + r], (unsigned)n);
        }
    }
    const bool eligible = active && selectable && !(flags & FX_FLAG_COLLISION) && total == total;
    double bc = eligible ? total : INFINITY;
    long long bi = eligible ? (long long)(g + P.g_base) : 0x7fffffffffffffffLL;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oc = __shfl_xor(bc, off);
        const long long oi = __shfl_xor(bi, off);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (lane == 0) { red_cost[wave] = bc; red_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < FX_BLOCK / 64; w++)
            if (red_cost[w] < bc || (red_cost[w] == bc && red_idx[w] < bi)) { bc = red_cost[w]; bi = red_idx[w]; }
        P.part_cost[blockIdx.x] = bc;
        P.part_idx[blockIdx.x] = bi;
    }
    if (tid < 2 + FX_NUM_REASONS && red_cnt[tid]) atomicAdd(&P.counters[tid], (unsigned long long)red_cnt[tid]);
}

// ---------------------------------------------------------------------------------------------------
// Selection kernel: one workgroup per agent.  Reduces the per-workgroup partials to the winner and
// counts the colliding candidates that the reference's cost-ordered walk would have visited before it
// (planner.py:336-357 `_collision_counter`).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fx_select_kernel(const DevProblem *__restrict__ probs) {
    __shared__ double sc[16];
    __shared__ long long si[16];
    __shared__ unsigned int scnt;
    const DevProblem &P = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bc = INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    for (int b = tid; b < P.n_blocks; b += 1024) {
        const double c = P.part_cost[b];
        const long long ix = P.part_idx[b];
        if (c < bc || (c == bc && ix < bi)) { bc = c; bi = ix; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oc = __shfl_xor(bc, off);
        const long long oi = __shfl_xor(bi, off);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (lane == 0) { sc[wave] = bc; si[wave] = bi; }
    if (tid == 0) scnt = 0;
    __syncthreads();
    bc = sc[0]; bi = si[0];
    for (int w = 1; w < 16; w++)
        if (sc[w] < bc || (sc[w] == bc && si[w] < bi)) { bc = sc[w]; bi = si[w]; }
    const bool none = bi == 0x7fffffffffffffffLL;
    // colliding selectable candidates ordered before the winner (all of them when nothing is collision-free)
    unsigned int cnt = 0;
    if (P.mode & FX_MODE_COLLISION) {
        for (int64_t g = tid; g < P.C; g += 1024) {
            const uint32_t f = P.flags[g];
            if ((f & FX_FLAG_SELECTABLE) && (f & FX_FLAG_COLLISION)) {
                const double c = P.cost[g];
                if (none || c < bc || (c == bc && g + P.g_base < bi)) cnt++;
            }
        }
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane == 0 && cnt) atomicAdd(&scnt, cnt);
    }
    __syncthreads();
    if (tid == 0) {
        P.counters[FX_CNT_BEST_IDX] = none ? ~0ULL : (unsigned long long)bi;
        P.counters[FX_CNT_BEST_COST] = none ? 0ULL : (unsigned long long)__double_as_longlong(bc);
        P.counters[FX_CNT_COLLISIONS] = scnt;
    }
}

// ---------------------------------------------------------------------------------------------------
// Top-k: the k best selectable collision-free candidates in (cost, index) order, one workgroup per agent.
// k successive arg-min passes with a strict lower bound; k <= 64, C up to millions is fine for the
// occasional host-side walk, the per-step fast path only needs fx_select_kernel.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fx_topk_kernel(const DevProblem *__restrict__ probs, int k, double *out_cost,
                                                       long long *out_idx) {
    __shared__ double sc[16];
    __shared__ long long si[16];
    const DevProblem &P = probs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double lb_c = -INFINITY;
    long long lb_i = -1;
    for (int r = 0; r < k; r++) {
        double bc = INFINITY;
        long long bi = 0x7fffffffffffffffLL;
        for (int64_t g = tid; g < P.C; g += 1024) {
            const uint32_t f = P.flags[g];
            if ((f & FX_FLAG_SELECTABLE) && !(f & FX_FLAG_COLLISION)) {
                const double c = P.cost[g];
                const long long gg = (long long)(g + P.g_base);
                const bool after = c > lb_c || (c == lb_c && gg > lb_i);
                if (after && (c < bc || (c == bc && gg < bi))) { bc = c; bi = gg; }
            }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            const double oc = __shfl_xor(bc, off);
            const long long oi = __shfl_xor(bi, off);
            if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { sc[wave] = bc; si[wave] = bi; }
        __syncthreads();
        bc = sc[0]; bi = si[0];
        for (int w = 1; w < 16; w++)
            if (sc[w] < bc || (sc[w] == bc && si[w] < bi)) { bc = sc[w]; bi = si[w]; }
        if (tid == 0) {
            out_cost[(int64_t)blockIdx.x * k + r] = bi == 0x7fffffffffffffffLL ? INFINITY : bc;
            out_idx[(int64_t)blockIdx.x * k + r] = bi == 0x7fffffffffffffffLL ? -1 : bi;
        }
        if (bi == 0x7fffffffffffffffLL) { lb_c = INFINITY; lb_i = 0x7fffffffffffffffLL; } else { lb_c = bc; lb_i = bi; }
    }
}

// ---- launchers (called from fx_api.hip) ----
extern "C" hipError_t fx_launch_eval(const DevProblem *d_probs, int n_agents, int max_blocks, int M_max, bool bundle,
                                     bool obst, bool extra, hipStream_t stream) {
    dim3 grid(max_blocks, n_agents), block(FX_BLOCK);
    size_t lds = (size_t)M_max * FX_REF_FIELDS * sizeof(double);
#define FX_LAUNCH(B, O, E)                                                                                     \
    do {                                                                                                       \
        if (lds > 48 * 1024) {                                                                                 \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&fx_eval_kernel<B, O, E>),      \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            if (e_ != hipSuccess) return e_;                                                                   \
        }                                                                                                      \
        hipLaunchKernelGGL((fx_eval_kernel<B, O, E>), grid, block, lds, stream, d_probs);                       \
    } while (0)
    const int sel = (bundle ? 4 : 0) | (obst ? 2 : 0) | (extra ? 1 : 0);
    switch (sel) {
    case 0: FX_LAUNCH(false, false, false); break;
    case 1: FX_LAUNCH(false, false, true); break;
    case 2: FX_LAUNCH(false, true, false); break;
    case 3: FX_LAUNCH(false, true, true); break;
    case 4: FX_LAUNCH(true, false, false); break;
    case 5: FX_LAUNCH(true, false, true); break;
    case 6: FX_LAUNCH(true, true, false); break;
    default: FX_LAUNCH(true, true, true); break;
    }
#undef FX_LAUNCH
    return hipGetLastError();
}

extern "C" hipError_t fx_launch_select(const DevProblem *d_probs, int n_agents, hipStream_t stream) {
    hipLaunchKernelGGL(fx_select_kernel, dim3(n_agents), dim3(1024), 0, stream, d_probs);
    return hipGetLastError();
}

extern "C" hipError_t fx_launch_topk(const DevProblem *d_probs, int n_agents, int k, double *out_cost, long long *out_idx,
                                     hipStream_t stream) {
    hipLaunchKernelGGL(fx_topk_kernel, dim3(n_agents), dim3(1024), 0, stream, d_probs, k, out_cost, out_idx);
    return hipGetLastError();
}
