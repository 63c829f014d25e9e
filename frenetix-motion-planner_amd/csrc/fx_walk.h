// fx_walk.h -- the per-candidate, per-step arithmetic shared by both evaluation kernels.
//
// A step consumes the longitudinal quantities of (candidate, step) -- a `LonRow`, either read from the
// workgroup's shared table (grid kernel) or derived by the lane itself (generic kernel) -- and does the
// lateral polynomial, the Frenet->Cartesian kinematics (reactive_planner.py:389-478), the five constraints
// (:480-533), the projection, the plane stores, the streamed partial costs and the obstacle stage.  Both kernels
// call the same inlined function, so they agree bit for bit by construction.
//
// Instruction diet (the walk is FP64-issue-bound):
//   * lateral polynomial with explicit FMAs and pre-multiplied derivative coefficients (15 instead of 37 ops),
//   * 1/sqrt(1+d'^2) from v_rsq_f64 + two coupled Newton steps gives cos and sec without sqrt + division,
//   * divisions as v_rcp_f64 + two Newton steps + one residual correction (8 ops, no div_scale/fmas/fixup),
//   * atan: wave-uniform fast path when every lane has |d'| < 7/16 (no range reduction, no division),
//   * the standstill branch (heading carried, real sin/cos) only runs when some lane of the wave needs it.
#pragma once

#include "fx_device.h"
#include "fx_math.h"

// Plane stores of the bundle: plain write-back stores, or agent-scope (write-through) stores that leave nothing dirty in
// the XCD's L2 for the end-of-kernel write-back (StepConst::store_wt, chosen by the host from the bundle size)
#define FX_ST_WT(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FX_ST_WB(ptr, val) (*(ptr) = (val))
// plane p of this step: the base of the step's row (wave-uniform when the step index is: scalar base + the lane's 32-bit byte
// offset, one address register for all 14 stores instead of a 64-bit add per plane)
#define FX_PLANE_AT(p) reinterpret_cast<FX_GLOBAL double *>(reinterpret_cast<FX_GLOBAL char *>(planes_i + (p) * ps) + lane_off)

// Developer statistics of the collision broad phase (-DFX_CULL_STATS, tools/cull_stats.py): compiles to nothing in the product
#ifdef FX_CULL_STATS
extern __device__ unsigned long long fx_cull_stats[16];
extern __device__ double fx_probe_bound;
#define FX_CSTAT(slot, val) do { if ((threadIdx.x & 63) == 0) atomicAdd(&fx_cull_stats[slot], (unsigned long long)(val)); } while (0)
#else
#define FX_CSTAT(slot, val) do { } while (0)
#endif

namespace fxk {

// a wave-uniform double pinned into a vector register pair (see fx_eval_grid_kernel.h, StepConst)
__device__ __forceinline__ double uniform_to_vgpr(double x) {
    double r;
    asm volatile("v_mov_b64 %0, %1" : "=v"(r) : "s"(x));
    return r;
}

struct alignas(16) LonRow {  // longitudinal quantities of one (pair, step); 128 B
    double s, sv, sa;        // s, clamped s_dot, s_ddot
    double th_ref, k_r, k_r_d;
    double px, py, nhx, nhy;  // foot point and unit normal (0 outside the projection domain)
    double r_sv;              // 1/s_dot (only meaningful when moving)
    double u1;                // s - s[0] (LOW_VEL_MODE lateral parameter)
    double c_ref, s_ref;      // cos / sin of th_ref (filled when a footprint is needed: collision stage, road boundary)
    uint32_t flags;           // LON_* bits
    uint32_t pad0;
    double pad1;
};
static_assert(sizeof(LonRow) == 128, "LonRow must be 128 bytes");

enum : uint32_t { LON_NEG = 1u, LON_ACC = 2u, LON_MOVING = 4u, LON_INDOMAIN = 8u };

// 1/d for normal, finite d: hardware estimate + two Newton steps (full double precision)
__device__ __forceinline__ double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}

// 1/d of a prediction-cost term: hardware estimate (~28 bits) + ONE Newton step, <= 11 ulp (tools/micro/rcpacc.hip).
// The term is one of K*S summands of a cost that is compared at 1e-9; every kernel variant uses this same function.
__device__ __forceinline__ double rcp_pred(double d) {
    double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    return fma(r, e, r);
}

// min of two doubles as one v_min_f64 (no canonicalisation of the operands; NaNs do not occur where it is used)
__device__ __forceinline__ double fx_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// 1 if the predicate holds in any lane of the wave, as a value in a scalar register (a wave-uniform bool is a lane mask to
// the compiler and gets widened to 0 / 1 per lane with vector instructions; this keeps bit bookkeeping on the scalar unit)
__device__ __forceinline__ unsigned wave_any_bit(bool p) {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(p);
    unsigned r;
    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(b) : "scc");
    return r;
}

// first active lane's double as a wave-uniform value (two v_readfirstlane)
__device__ __forceinline__ double uniform_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(b & 0xffffffffu)), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// max over the wave of an unsigned value, wave-uniform result: four row shifts and two row broadcasts in the data-parallel
// primitives of the vector unit (no LDS crossbar), then lane 63
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define FX_DPP_MAX(ctrl, rows) v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false))
    FX_DPP_MAX(0x111, 0xf);  // row_shr:1
    FX_DPP_MAX(0x112, 0xf);  // row_shr:2
    FX_DPP_MAX(0x114, 0xf);  // row_shr:4
    FX_DPP_MAX(0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds the row's maximum
    FX_DPP_MAX(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    FX_DPP_MAX(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef FX_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// n / d with one residual correction on the quotient (correctly rounded for the operands of this kernel)
__device__ __forceinline__ double fdiv(double n, double d) {
    const double r = rcp_nr(d);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// sqrt(x) and 1/sqrt(x) together: v_rsq_f64 + two coupled (Goldschmidt) Newton steps
__device__ __forceinline__ void sqrt_rsqrt(double x, double &sq, double &rsq) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    // final residual on sqrt
    g = fma(fma(-g, g, x), h, g);
    sq = g;
    rsq = h + h;
}

// Longitudinal polynomial c0..c5 with c0 = s0, c1 = ss0, c2 = sss0 / 2 (closed forms of the reference's LAPACK solves):
//   FX_LON_VELOCITY_KEEPING: quartic to (x1 = end velocity, end acceleration a1)   polynomial_trajectory.py:452-488
//   FX_LON_STOP_POINT:       quintic to (x1 = end position, 0, 0)                  :293-343, reactive_planner.py:641-643
__device__ __forceinline__ void lon_coeffs(int lon_mode, double s0, double ss0, double sss0, double T, double x1, double a1,
                                           double &c3, double &c4, double &c5) {
    const double T2 = T * T;
    if (lon_mode == FX_LON_STOP_POINT) {
        const double T3 = T2 * T, T4 = T3 * T, T5 = T4 * T;
        const double b0 = x1 - s0 - ss0 * T - .5 * sss0 * T2;
        const double b1 = 0.0 - ss0 - sss0 * T;
        const double b2 = 0.0 - sss0;
        c3 = fdiv(10.0 * b0 - 4.0 * b1 * T + .5 * b2 * T2, T3);
        c4 = fdiv(-15.0 * b0 + 7.0 * b1 * T - b2 * T2, T4);
        c5 = fdiv(6.0 * b0 - 3.0 * b1 * T + .5 * b2 * T2, T5);
    } else {
        const double b1 = x1 - ss0 - sss0 * T, b2 = a1 - sss0;
        c3 = fdiv(3.0 * b1 - T * b2, 3.0 * T2);
        c4 = fdiv(T * b2 - 2.0 * b1, 4.0 * T2 * T);
        c5 = 0.0;
    }
}

// Longitudinal quantities of (T, v1) at step i: quartic sample or horizon extension (reactive_planner.py:313-322),
// validity / pre-filter predicates (:350-355,375), reference-segment lookup with Python's negative-index wrap
// (:415-420), interpolate_angle (utils_coordinate_system.py:137-155), reference curvature (:457-460) and the
// projection foot point + unit normal.  `knot(k)` returns knot k, `kpos(k)` its arclength.
template <typename KnotFn, typename PosFn>
__device__ __forceinline__ LonRow make_lon_row(int i, int S, int M, double dt, double a_max, double cl0, double cl1, double cl2,
                                               double cl3, double cl4, double cl5, int traj_len, const double *tp, double rp_first,
                                               double rp_last, double guess_scale, bool want_trig, KnotFn knot, PosFn kpos,
                                               bool pseudo_normal = false) {
    const int ie = i < traj_len ? i : traj_len - 1;  // sample that is evaluated (the last one feeds the extension)
    const double *te = tp + ie * FX_TP;
    const double t1 = te[0], t2 = te[1], t3 = te[2], t4 = te[3], t5 = te[4];
    // calc_position / calc_velocity / calc_acceleration (polynomial_trajectory.py:241-273); c5 = 0 for the quartic
    double s_i = cl0 + cl1 * t1 + cl2 * t2 + cl3 * t3 + cl4 * t4 + cl5 * t5;
    double sv_i = cl1 + 2. * cl2 * t1 + 3. * cl3 * t2 + 4. * cl4 * t3 + 5. * cl5 * t4;
    double sa_i = 2 * cl2 + 6 * cl3 * t1 + 12 * cl4 * t2 + 20 * cl5 * t3;
    if (i >= traj_len) {  // s[i] = s[i-1] + dt * s_dot_end, one rounding per step as in the reference loop
        for (int j = traj_len; j <= i; j++) s_i = s_i + dt * sv_i;
        sa_i = 0.0;
    }
    LonRow r;
    r.flags = (sv_i < -FX_EPS ? LON_NEG : 0u) | (fabs(sa_i) > a_max ? LON_ACC : 0u);
    if (fabs(sv_i) < FX_EPS) sv_i = 0.0;
    if (sv_i > 0.001) r.flags |= LON_MOVING;
    r.s = s_i; r.sv = sv_i; r.sa = sa_i;
    r.u1 = s_i - cl0;
    // reciprocals are only consumed on moving steps (walk_step selects them away otherwise)
    const bool mv = (r.flags & LON_MOVING) != 0;
    r.r_sv = mv ? rcp_nr(sv_i) : 0.0;
    // upper_bound(ref_pos, s): first try the segment a uniformly spaced reference would put s in (resampled
    // reference paths are close to uniform), then bisect whatever interval that probe leaves
    int lo = 0, hi = M;
    if (M >= 2 && s_i >= rp_first && s_i < rp_last) {
        int kg = (int)((s_i - rp_first) * guess_scale);
        kg = kg < 0 ? 0 : (kg > M - 2 ? M - 2 : kg);
        const double p0 = kpos(kg), p1 = kpos(kg + 1);
        if (p0 > s_i) hi = kg;
        else if (p1 > s_i) { lo = kg + 1; hi = kg + 1; }
        else lo = kg + 2 > M ? M : kg + 2;
    }
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (kpos(mid) > s_i) hi = mid; else lo = mid + 1;
    }
    const int ub = lo;
    const int i1 = ub == M ? 0 : ub;
    const int i0 = i1 == 0 ? M - 1 : i1 - 1;
    const Knot k0 = knot(i0), k1 = knot(i1);
    const double seg = k1.pos - k0.pos, r_seg = rcp_nr(seg);
    const double s_lambda = div_rcp(s_i - k0.pos, seg, r_seg);
    r.th_ref = wrap_pm_2pi(div_rcp((k1.theta - k0.theta) * (s_i - k0.pos), seg, r_seg) + k0.theta);
    r.k_r = (k1.curv - k0.curv) * s_lambda + k0.curv;
    r.k_r_d = (k1.curv_d - k0.curv_d) * s_lambda + k0.curv_d;
    r.px = r.py = r.nhx = r.nhy = 0.0;
    r.pad0 = 0;
    if (s_i >= rp_first && s_i <= rp_last) {
        r.flags |= LON_INDOMAIN;
        int kk = ub - 1;
        kk = kk < 0 ? 0 : (kk > M - 2 ? M - 2 : kk);
        r.pad0 = (uint32_t)kk;  // reference segment of the foot point: selects the road-boundary bin
        // kk == i0 unless the lookup wrapped (s outside the reference): same segment, same lambda
        const Knot q0 = kk == i0 ? k0 : knot(kk), q1 = kk == i0 ? k1 : knot(kk + 1);
        const double lam = kk == i0 ? s_lambda : fdiv(s_i - q0.pos, q1.pos - q0.pos);
        r.px = q0.x + lam * (q1.x - q0.x);
        r.py = q0.y + lam * (q1.y - q0.y);
        const double nx = q0.nx + lam * (q1.nx - q0.nx), ny = q0.ny + lam * (q1.ny - q0.ny);
        double nn, r_nn;
        sqrt_rsqrt(nx * nx + ny * ny, nn, r_nn);
        // FX_MODE_PROJ_PSEUDO_NORMAL: d runs along the interpolated normal itself (a pseudo-distance), not along its unit vector
        r.nhx = pseudo_normal ? nx : div_rcp(nx, nn, r_nn);
        r.nhy = pseudo_normal ? ny : div_rcp(ny, nn, r_nn);
    }
    r.pad1 = 0.0;
    r.c_ref = 1.0; r.s_ref = 0.0;
    if (want_trig) fxm::sincos(r.th_ref, &r.s_ref, &r.c_ref);
    return r;
}

// heading of a *moving* step j from its row: atan(d'/1) + theta_ref, exactly as walk_step computes it
__device__ __forceinline__ double heading_of_moving_step(const LonRow &r, double dv_j) {
    const double q = dv_j * r.r_sv;
    const double dp = fma(fma(-r.sv, q, dv_j), r.r_sv, q);
    return fxm::atan(dp) + r.th_ref;
}

// Time table row of step i in LDS (FX_TP doubles): the powers t .. t^5 and the derivative factors 2t, 3t^2, 4t^3, 5t^4,
// 6t, 12t^2, 20t^3.  The factors of the reference's velocity / acceleration polynomials (2. * c2 * t, 3. * c3 * t2, ...,
// polynomial_trajectory.py:251-257) ride on the wave-uniform powers instead of on eight more per-lane coefficients.
struct LatU {
    double u1, u2, u3, u4, u5, w2, w3, w4, w5, z3, z4, z5;
    __device__ __forceinline__ void from_table(const double *row) {
        u1 = row[0]; u2 = row[1]; u3 = row[2]; u4 = row[3]; u5 = row[4];
        w2 = row[5]; w3 = row[6]; w4 = row[7]; w5 = row[8]; z3 = row[9]; z4 = row[10]; z5 = row[11];
    }
    __device__ __forceinline__ void from_parameter(double u) {  // LOW_VEL_MODE: the lateral parameter is the arc length s - s0
        u1 = u; u2 = u1 * u1; u3 = u2 * u1; u4 = u2 * u2; u5 = u4 * u1;
        w2 = 2. * u1; w3 = 3. * u2; w4 = 4. * u3; w5 = 5. * u4; z3 = 6 * u1; z4 = 12 * u2; z5 = 20 * u3;
    }
};
__device__ __forceinline__ void fill_time_row(double *row, double t1, double t2, double t3, double t4, double t5) {
    row[0] = t1; row[1] = t2; row[2] = t3; row[3] = t4; row[4] = t5;
    row[5] = 2. * t1; row[6] = 3. * t2; row[7] = 4. * t3; row[8] = 5. * t4; row[9] = 6 * t1; row[10] = 12 * t2; row[11] = 20 * t3;
    row[12] = 0.0; row[13] = 0.0;   // obstacle masks of the step (filled by the kernel that stages them)
}

// lateral quintic: position, velocity and acceleration by fused sums left to right
struct LatPoly {
    double c0, c1, c2, c3, c4, c5;
    double a2;                  // 2 c2
    __device__ __forceinline__ void set(double k0, double k1, double k2, double k3, double k4, double k5) {
        c0 = k0; c1 = k1; c2 = k2; c3 = k3; c4 = k4; c5 = k5;
        a2 = 2 * k2;
    }
    __device__ __forceinline__ void eval(const LatU &U, double &d, double &dv, double &da) const {
        d = fma(c5, U.u5, fma(c4, U.u4, fma(c3, U.u3, fma(c2, U.u2, fma(c1, U.u1, c0)))));
        dv = fma(c5, U.w5, fma(c4, U.w4, fma(c3, U.w3, fma(c2, U.w2, c1))));
        da = fma(c5, U.z5, fma(c4, U.z4, fma(c3, U.z3, a2)));
    }
};

// a 64-bit value known to be wave-uniform, moved to scalar registers
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffu)) |
           ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32);
}

struct StepConst {  // wave-uniform constants of the walk
    double dt, r_dt, kappa_max, a_max, v_switch, av_switch, v_des, wb, half_len, half_wid;
    int S, half, K;
    bool low_vel, dbg, do_collision;
    bool store_wt;          // write-through plane stores
    int n_bound;            // road-boundary pieces (0: stage off)
    double bound_d_reach;   // |d| beyond this counts as off the road
    double ox, oy;          // origin of the hot obstacle table's coordinates (a reference-path point near the ego)
    double gap_margin;      // broad phase: centre-gap values up to this go to the exact test (rounding of the expanded form)
    fxm::lds_cptr atan_k;   // KTAB: LDS copy of atan's polynomial coefficients (fx_math.h, atan_small_tab)
    float cull_r0;          // wave-level cull: sqrt(2) * |(L/2, W/2)|, rounded up (constant part of a hull's bounding radius)
};

struct StepCarry {  // per-lane state carried from step to step
    double th_prev, kap_prev;
    double bx_prev, by_prev, ux_prev, uy_prev;
};

struct StepAcc {  // per-lane accumulators over the emitted steps
    bool neg, acc_viol, collided;
    uint32_t step_reasons, first_key;
    int fail_step;
    int bound_step;  // first emitted step whose footprint meets the road boundary, INT_MAX if none
    double sum_abs_d, sum_voff, pred, d_end, v_end;
};

struct StepOut {  // per-step values the windowed (EXTRA) costs of the generic kernel need
    double a, v, th_cl, x, y;
};

// One step of one candidate.  `planes_i` + `lane_off` bytes = address of plane 0 at (step i, this candidate); ps = plane stride.
// USTEP: the step index is wave-uniform (one lane per candidate, or parts on different waves), so the obstacle
// records of the step come in through scalar loads; otherwise every lane reads its own step's records.
// Road boundary of the agent as the walk sees it (address-space-1 pointers; n_bound lives in StepConst).
struct BoundView {
    const FX_GLOBAL double *piece;
    const FX_GLOBAL int32_t *bin, *item;
};

// Per-wave staging of the hot obstacle table (grid kernel, wave-uniform step index).  The walk visits every
// (step, obstacle) pair with wave-uniform operands; fetching them with dependent scalar loads inside the obstacle loop
// costs one scalar-cache / L2 round trip per visit.  Instead the wave copies the K x 80 B block of the step with ONE
// coalesced vector load, issued a whole step ahead (`pre`), parks it in its own LDS block and the obstacle loops read
// entries back with broadcast ds_read_b128, the next two entries requested before the current two are consumed.
//
// A hot entry holds the operands in the form that costs the fewest operations per visit (host: fx_api.hip):
//   prediction cost (collision_probability.py:283-292): the inverse covariance A (symmetric part) is factored
//   A = L^T L, L = [[l11, l12], [0, l22]], so that the Mahalanobis form of the ego point p = (x, y) - O is
//       m = (l11 px + l12 py - cu)^2 + (l22 py - cw)^2 ,   cu = l11 mx + l12 my,  cw = l22 my   (mu relative to O)
//   -- five operations instead of eight.  O is a point of the reference path near the ego, so the products stay small
//   against their difference.
//   broad phase of the collision test: |h - c|^2 - (r_o + r_e)^2 expanded in the ego hull's (c, r_e):
//       g = (ck + w) + hx2 cx + hy2 cy + hr2 r_e ,  hx2 = -2 hx, hy2 = -2 hy, hr2 = -2 r_o, ck = |h|^2 - r_o^2, w = |c|^2 - r_e^2
//   (all relative to O) -- four operations and a minimum instead of seven and a compare.
#define FX_HOT_L11 0
#define FX_HOT_L12 1
#define FX_HOT_CU 2
#define FX_HOT_L22 3
#define FX_HOT_CW 4
#define FX_HOT_HX2 5
#define FX_HOT_HY2 6
#define FX_HOT_HR2 7
#define FX_HOT_CK 8
typedef double fx_d2 __attribute__((ext_vector_type(2)));  // one 16-byte load
#ifndef FX_STAGE_FENCE
#define FX_STAGE_FENCE 1
#endif
struct ObsHot {
    double *lds;                    // this wave's block [K][FX_HOT_STRIDE] (nullptr: staging off)
    const FX_GLOBAL double *tab;    // hot[S][K][FX_HOT_STRIDE]
    int n_el;                       // K * FX_HOT_STRIDE
    int lane;
    double pre[FX_HOT_PRE];         // elements lane, lane + 64, ... of the NEXT staged step

    __device__ __forceinline__ void prefetch(int step) {
        const FX_GLOBAL double *src = tab + (int64_t)step * n_el;
#pragma unroll
        for (int j = 0; j < FX_HOT_PRE; j++) {
            const int e = lane + 64 * j;
            pre[j] = e < n_el ? src[e] : 0.0;
        }
    }
    // park the prefetched block of `step` in LDS (elements beyond the prefetch window come straight from memory),
    // then start the load of `next` (< 0: none)
    __device__ __forceinline__ void stage(int step, int next) {
        // the previous step's reads are behind the new writes and the writes in front of this step's reads: LDS
        // operations of one wave execute in order, the fences only keep the compiler from reordering across lanes' views
#if FX_STAGE_FENCE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#endif
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < FX_HOT_PRE; j++) {
            const int e = lane + 64 * j;
            if (e < n_el) lds[e] = pre[j];
        }
        if (n_el > 64 * FX_HOT_PRE) {
            const FX_GLOBAL double *src = tab + (int64_t)step * n_el;
            for (int e = lane + 64 * FX_HOT_PRE; e < n_el; e += 64) lds[e] = src[e];
        }
#if FX_STAGE_FENCE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#endif
        __builtin_amdgcn_wave_barrier();
#if FX_STAGE_FENCE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
        if (next >= 0) prefetch(next);
    }
};

// cos / sin of the global heading from the pieces the step already has: theta_gl = theta_cl + theta_ref with
// cos(theta_cl), sin(theta_cl) = tan * cos from the kinematics and cos / sin(theta_ref) from the row -- four operations
// instead of a sincos evaluation (the carried heading of a standstill step satisfies the same identity to an ulp)
__device__ __forceinline__ void heading_trig(const LonRow &r, double cosTheta, double tanTheta, double &cu, double &su) {
    const double sinTheta = tanTheta * cosTheta;
    cu = fma(-sinTheta, r.s_ref, cosTheta * r.c_ref);
    su = fma(sinTheta, r.c_ref, cosTheta * r.s_ref);
}

// KTAB: the constants of the rare paths (general atan reduction, standstill sincos) come from constant memory instead of
// loop-long registers (fx_math.h)
template <bool OBST, bool USTEP, bool HOT = false, bool KTAB = false, typename PlanePtr, typename ObsD, typename ObsM>
__device__ __forceinline__ void walk_step(const StepConst &K, const LonRow &r, const LatPoly &L, const double *tp, int i,
                                          int traj_len, double d_ext, bool emit, bool store, PlanePtr planes_i, uint32_t lane_off, int64_t ps,
                                          StepCarry &C, StepAcc &A, StepOut &O, ObsD obs_rec, ObsM obs_pmask, ObsM obs_hmask,
                                          const BoundView &B, ObsHot *H = nullptr, int i_next = -1, const bool neigh = false,
                                          const bool has_prev = false) {
    // neigh (wave-uniform; lane split with ONE step per lane): the lane to the left walks step i - 1 of the same candidate in this
    // very call, so what this step needs of it -- heading, curvature, ego box -- is taken from that lane (has_prev: it exists, i.e.
    // this is not the candidate's first part) instead of being recomputed in a carry-in step: the same arithmetic on the same step
    // by another lane, bit for bit, and one walk_step per lane instead of two.
    const int S = K.S;
#if defined(FX_PROBE) && defined(FX_PROBE_STEP)   // in-step timeline of one step (tools/probe_step.py): core-clock stamps
#define FX_STEP_STAMP(k) do { if (OBST && HOT && i == FX_PROBE_STEP + ((k) == 5)) FX_OSTAMP((k)); } while (0)
#else
#define FX_STEP_STAMP(k) do { } while (0)
#endif
    FX_STEP_STAMP(5);   // (the step after the probed one: its entry closes the probed step)
    FX_STEP_STAMP(1);
    if (OBST && HOT && USTEP && K.K > 0) H->stage(__builtin_amdgcn_readfirstlane(i), i_next);
    const double s_i = r.s, sv_i = r.sv, sa_i = r.sa;
    // -- lateral polynomial (reactive_planner.py:326-346) --
    double d_i, dv_i, da_i;
    if (i < traj_len) {
        LatU U;
        if (K.low_vel) U.from_parameter(r.u1);
        else U.from_table(tp + i * FX_TP);
        L.eval(U, d_i, dv_i, da_i);
    } else {
        d_i = d_ext; dv_i = 0.0; da_i = 0.0;
    }
    if (emit) {
        A.neg |= (r.flags & LON_NEG) != 0;
        A.acc_viol |= (r.flags & LON_ACC) != 0;
    }
    // -- d', d'' (:392-412) --
    const bool moving = (r.flags & LON_MOVING) != 0;
    const double sv2 = sv_i * sv_i;
    double dp, dpp;
    if (!K.low_vel) {
        const double q = dv_i * r.r_sv;
        dp = moving ? fma(fma(-sv_i, q, dv_i), r.r_sv, q) : 0.;
        const double ddot = da_i - dp * sa_i;
        const double r_sv2 = r.r_sv * r.r_sv;  // 1 / s_dot^2 to an ulp; the residual step below settles the quotient
        const double q2 = ddot * r_sv2;
        dpp = moving ? fma(fma(-sv2, q2, ddot), r_sv2, q2) : 0.;
    } else {
        dp = dv_i;
        dpp = da_i;
    }
    const double th_ref = r.th_ref, k_r = r.k_r, k_r_d = r.k_r_d;
    // -- theta_cl = arctan2(d', 1); cos = 1/sqrt(1+d'^2), tan = d' (:423-433) --
    double secTheta, cosTheta;
    sqrt_rsqrt(fma(dp, dp, 1.0), secTheta, cosTheta);
    double tanTheta = dp;
    double th_cl = __all(fabs(dp) < 0.4375) ? (KTAB ? fxm::atan_small_tab(dp, K.atan_k) : fxm::atan_small(dp)) : fxm::atan<KTAB>(dp);
    double th_gl = th_cl + th_ref;
    const bool still = !(moving || K.low_vel);
    if (__any(still)) {  // standstill at high-speed mode keeps the previous global heading (:447-454)
        const double th_gl_s = C.th_prev;  // x_0.orientation at i == 0
        const double th_cl_s = th_gl_s - th_ref;
        double sn, cs;
        fxm::sincos<KTAB>(th_cl_s, &sn, &cs);
        const double sec_s = rcp_nr(cs);
        if (still) { th_gl = th_gl_s; th_cl = th_cl_s; cosTheta = cs; secTheta = sec_s; tanTheta = sn * sec_s; }
    }
    // -- global curvature, velocity, acceleration (:463-478) --
    const double oneKrD = fma(-k_r, d_i, 1.0);
    const double cok = cosTheta * rcp_nr(oneKrD);   // cos / (1 - k_r d)
    const double okc = oneKrD * secTheta;           // (1 - k_r d) / cos
    const double kap = fma(fma(fma(k_r, dp, k_r_d * d_i), tanTheta, dpp) * cosTheta, cok * cok, cok * k_r);
    const double v_i = sv_i * okc;
    const double a_i = fma(sa_i, okc, (sv2 * secTheta) * (oneKrD * tanTheta * fma(kap, okc, -k_r) - fma(k_r, dp, k_r_d * d_i)));
    if (neigh) {
        const double th_left = __shfl_up(th_gl, 1), kap_left = __shfl_up(kap, 1);
        if (has_prev) { C.th_prev = th_left; C.kap_prev = kap_left; }
    }
    // -- constraints (:480-533): bit r = reason r --
    if (emit) {
        uint32_t hit = 0;
        if (v_i < -FX_EPS) hit |= 1u << 4;
        if (!(fabs(kap) <= K.kappa_max)) hit |= 1u << 5;  // also catches a NaN curvature
        const double yaw_rate = i > 0 ? div_rcp(th_gl - C.th_prev, K.dt, K.r_dt) : 0.;
        if (fabs(np_round5(yaw_rate)) > K.kappa_max * v_i) hit |= 1u << 6;
        const double kap_rate = i > 0 ? div_rcp(kap - C.kap_prev, K.dt, K.r_dt) : 0.;
        if (fabs(kap_rate) > 0.4) hit |= 1u << 7;
        // a <= a_max * v_switch / v  <=>  a * v <= a_max * v_switch for v > v_switch > 0: no division
        const bool over = v_i > K.v_switch ? !(a_i * v_i <= K.av_switch) : !(a_i <= K.a_max);
        if (!(-K.a_max <= a_i) || over) hit |= 1u << 8;
        if (K.dbg) A.step_reasons |= hit;
        else if (hit && A.first_key == 0xffffffffu) A.first_key = ((uint32_t)i << 4) | (uint32_t)(__ffs((int)hit) - 1);
    }
    const double kap_dot = i > 0 ? kap - C.kap_prev : 0.0;  // np.append([0], np.diff(kappa_gl)) (:552)

    // -- (s, d) -> (x, y): foot point + d * unit normal, 0 from the first step outside the domain on (:537-547) --
    double x_i = 0.0, y_i = 0.0;
    if (!(r.flags & LON_INDOMAIN)) {
        if (emit && A.fail_step == 0x7fffffff) A.fail_step = i;
    } else if (A.fail_step == 0x7fffffff) {
        x_i = fma(d_i, r.nhx, r.px);
        y_i = fma(d_i, r.nhy, r.py);
    }

    // -- SoA bundle (trajectories.py:56-334) --
    if (store) {
        // the offset's widening has to happen next to the stores for them to take the scalar-base + 32-bit-offset form
        asm volatile("" : "+v"(lane_off));
        // ONE wave-uniform branch for the store mode, then fourteen straight stores
        if (K.store_wt) {
            FX_ST_WT(FX_PLANE_AT(FX_PL_X), x_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_Y), y_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_THETA), th_gl);
            FX_ST_WT(FX_PLANE_AT(FX_PL_V), v_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_A), a_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_KAPPA), kap);
            FX_ST_WT(FX_PLANE_AT(FX_PL_KAPPA_DOT), kap_dot);
            FX_ST_WT(FX_PLANE_AT(FX_PL_S), s_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_D), d_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_THETA_CL), th_cl);
            FX_ST_WT(FX_PLANE_AT(FX_PL_S_DOT), sv_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_S_DDOT), sa_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_D_DOT), dv_i);
            FX_ST_WT(FX_PLANE_AT(FX_PL_D_DDOT), da_i);
        } else {
            FX_ST_WB(FX_PLANE_AT(FX_PL_X), x_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_Y), y_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_THETA), th_gl);
            FX_ST_WB(FX_PLANE_AT(FX_PL_V), v_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_A), a_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_KAPPA), kap);
            FX_ST_WB(FX_PLANE_AT(FX_PL_KAPPA_DOT), kap_dot);
            FX_ST_WB(FX_PLANE_AT(FX_PL_S), s_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_D), d_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_THETA_CL), th_cl);
            FX_ST_WB(FX_PLANE_AT(FX_PL_S_DOT), sv_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_S_DDOT), sa_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_D_DOT), dv_i);
            FX_ST_WB(FX_PLANE_AT(FX_PL_D_DDOT), da_i);
        }
    }

    // -- partial costs, streamed --
    if (emit) {
        A.sum_abs_d += fabs(d_i);                                               // partial_cost_functions.py:166-167
        if (i >= K.half && i < S - 1) A.sum_voff += fabs(v_i - K.v_des);        // :125-127
        if (i == S - 1) { A.d_end = d_i; A.v_end = v_i; }
    }
    FX_STEP_STAMP(2);
    if (OBST && HOT && USTEP && K.K > 0) {
        // ---- obstacle stage on the staged table (wave-uniform step index, every branch wave-uniform) ----
        const int iu = __builtin_amdgcn_readfirstlane(i);
        const int nK = K.K;
        // the step's masks ride in the time table's row (LDS, filled in the kernel's first phase): no global load per step
        const unsigned long long *mrow = reinterpret_cast<const unsigned long long *>(tp + iu * FX_TP + 12);
        unsigned long long pm = mrow[0], hm_now = mrow[1];
        const unsigned long long hm_next = iu + 1 < S ? mrow[FX_TP + 1] : 0ULL;
        pm = uniform_u64(pm); hm_now = uniform_u64(hm_now);
        if (!emit) pm = 0ULL;
        const unsigned long long full = nK >= 64 ? ~0ULL : ((1ULL << nK) - 1ULL);
        const double *hot = H->lds;
        const auto rec_i = obs_rec + (int64_t)iu * nK * 12;
        // -- prediction cost (collision_probability.py:283-292): sum over the obstacles of 1 / m^2 --
        if (pm) {
            const double xr = x_i - K.ox, yr = y_i - K.oy;
            auto msq = [&](fx_d2 l, fx_d2 c, double cw) {  // m^2 of one obstacle; l = (l11, l12), c = (cu, l22)
                const double u = fma(l.x, xr, fma(l.y, yr, -c.x));
                const double w = fma(c.y, yr, -cw);
                const double m = fma(u, u, w * w);
                return m * m;
            };
            auto term = [&](fx_d2 l, fx_d2 c, double cw) { return rcp_pred(msq(l, c, cw)); };
            // 1/a + 1/b + 1/c + 1/d over ONE reciprocal: ((a + b) cd + (c + d) ab) / (ab cd) -- 11 operations for four terms
            // instead of 16, one quarter-rate reciprocal instead of four.  The products stay far inside the exponent range
            // (a term is a squared Mahalanobis form); a zero or non-finite term surfaces in the sum and takes the exact path.
            auto four = [&](double a, double b, double c, double d) {
                const double ab = a * b, cd = c * d;
                const double num = fma(a + b, cd, (c + d) * ab);
                return num * rcp_pred(ab * cd);
            };
            auto ld = [&](int k, fx_d2 &l, fx_d2 &c, double &cw) {
                const double *q = hot + (size_t)k * FX_HOT_STRIDE;
                l = *reinterpret_cast<const fx_d2 *>(q);
                c = *reinterpret_cast<const fx_d2 *>(q + 2);
                cw = q[FX_HOT_CW];
            };
            double s0 = 0.0, s1 = 0.0;
            if (pm == full) {
                // four obstacles per iteration in two alternating register sets: the loads of one pair are in flight
                // while the other pair is consumed, and no value is copied between the sets.  Every complete group of four
                // shares one reciprocal (the look-ahead loads clamp their index instead of ending the loop a group early:
                // K = 20 is five groups, not four groups and four single reciprocals)
                fx_d2 la, ca, lb, cb, lc, cc, ld_, cd;
                double wa, wb, wc, wd;
                int k = 0;
                const int kz = nK - 1;
                ld(0, la, ca, wa);
                ld(min(1, kz), lb, cb, wb);
                for (; k + 3 < nK; k += 4) {
                    ld(k + 2, lc, cc, wc); ld(k + 3, ld_, cd, wd);
                    const double qa = msq(la, ca, wa), qb = msq(lb, cb, wb);
                    ld(min(k + 4, kz), la, ca, wa); ld(min(k + 5, kz), lb, cb, wb);
                    s0 += four(qa, qb, msq(lc, cc, wc), msq(ld_, cd, wd));
                }
                // entries k (a) and k + 1 (b) are loaded where they exist; up to three remain
                if (k < nK) s0 += term(la, ca, wa);
                if (k + 1 < nK) s1 += term(lb, cb, wb);
                if (k + 2 < nK) { ld(k + 2, lc, cc, wc); s0 += term(lc, cc, wc); }
            } else {
                unsigned long long m = pm;
                while (m) {
                    const int k = __builtin_ctzll(m);
                    m &= m - 1;
                    fx_d2 l, c; double cw;
                    ld(k, l, c, cw);
                    s0 += term(l, c, cw);
                }
            }
            double ssum = s0 + s1;
            // anything not finite (an obstacle centre hit to the last bit, a covariance without a Cholesky factor: the
            // host leaves a zero entry) sends the step to the reference form on the raw records -- rare
            if (__any(!(ssum < 1e300))) {
                double acc = 0.0;
                unsigned long long m = pm;
                while (m) {
                    const int k = __builtin_ctzll(m);
                    m &= m - 1;
                    const auto q = rec_i + k * 12;
                    const double e0 = x_i - q[0], e1 = y_i - q[1];
                    const double r0 = fma(e1, q[4], e0 * q[2]), r1 = fma(e1, q[5], e0 * q[3]);
                    const double mq = fma(r1, e1, r0 * e0);
                    const double mm = mq * mq;
                    acc += mm > 0.0 ? rcp_pred(mm) : 1.0 / mm;
                }
                ssum = !(ssum < 1e300) ? acc : ssum;
            }
            A.pred += ssum;
        }
        FX_STEP_STAMP(3);
        // -- collision: OBB-sum hull of ego boxes (i-1, i) against the obstacle hulls of this step --
        if (K.do_collision && (hm_now | hm_next) != 0ULL && i >= 1) {
            double su, cu;
            heading_trig(r, cosTheta, tanTheta, cu, su);
            const double bx = fma(K.wb, cu, x_i), by = fma(K.wb, su, y_i);
            const unsigned long long hm = emit ? hm_now : 0ULL;
            if (hm) {
                // Wave-level cull BEFORE the hull is built, all obstacles at once.  Both ego boxes lie inside the discs of radius
                // hd = |(L/2, W/2)| around their centres c0, c1, hence inside the disc around the midpoint m of the centres with
                // radius rho = |c1 - c0| / 2 + hd; the OBB-sum hull is the bounding rectangle of the two boxes in some orthonormal
                // frame, so it lies inside that disc's bounding square in the same frame, i.e. inside the disc (m, sqrt(2) rho).
                // The hulls of the wave's 64 candidates at this step therefore lie inside the disc around the first lane's
                // midpoint m0 with radius R = max over the lanes of (|m - m0| + sqrt(2) rho), and lane k tests obstacle k's circle
                // (h_k, r_o) against that one: a pair of hulls that overlap has |h_k - m| <= r_o + sqrt(2) rho, hence
                // |h_k - m0| <= r_o + R -- nothing the exact test would report is dropped (the slack factors cover the
                // single-precision square roots and conversions; a box that is not finite keeps every obstacle).  Candidates of a
                // wave are neighbours in the sampling grid, R is a few metres, and for most steps no obstacle survives: the hull
                // (~70 FP64 operations per candidate and step) is only built where something is near.
                unsigned long long cand;
                {
                    const double mxr = fma(0.5, C.bx_prev + bx, -K.ox), myr = fma(0.5, C.by_prev + by, -K.oy);
                    const double tx = bx - C.bx_prev, ty = by - C.by_prev;
                    const double qd = fma(tx, tx, ty * ty);
                    const double m0x = uniform_f64(mxr), m0y = uniform_f64(myr);
                    const double dx = mxr - m0x, dy = myr - m0y;
                    const double qe = fma(dx, dx, dy * dy);
                    const bool bad = !(qd + qe < 1e300);
                    // |m - m0| + sqrt(2) (|c1 - c0| / 2 + hd), each term rounded up
                    const float reach = fmaf(__builtin_amdgcn_sqrtf((float)qe), 1.00001f,
                                             fmaf(__builtin_amdgcn_sqrtf((float)qd), 0.707115f, K.cull_r0));
                    const unsigned rb = wave_max_u32(bad ? 0u : __float_as_uint(reach));  // reach >= 0: the bit patterns order like the values
                    const double R = (double)__uint_as_float(rb);
                    const int kl = min(H->lane, nK - 1);
                    const double *qo = hot + (size_t)kl * FX_HOT_STRIDE;
                    const double ex = fma(-0.5, qo[FX_HOT_HX2], -m0x), ey = fma(-0.5, qo[FX_HOT_HY2], -m0y);  // h - m0
                    const double rr = fma(-0.5, qo[FX_HOT_HR2], R) * 1.00001;                                 // r_o + R
                    cand = __builtin_amdgcn_ballot_w64(!(fma(ex, ex, ey * ey) > rr * rr)) & hm;
                    if (wave_any_bit(bad)) cand = hm;
                }
                FX_CSTAT(0, 1); FX_CSTAT(1, cand != 0); FX_CSTAT(2, __popcll(cand));
                if (cand) {
                    const Obb hull = obb_hull(C.bx_prev, C.by_prev, C.ux_prev, C.uy_prev, bx, by, cu, su, K.half_len, K.half_wid);
                    // per-lane broad phase on circles for the survivors: a box lies inside the circle around its centre with
                    // radius h1 + h2, so hulls whose centres are farther apart than the sum of those radii (1e-6 relative slack on
                    // both) are separated and the axis test would say so.  g = |h - c|^2 - (r_o + r_e)^2 in the expanded form of
                    // the table; the exact test runs for the obstacles where some lane has g <= margin (the margin covers the
                    // rounding of the expanded form).
                    const double re = (hull.h1 + hull.h2) * 1.000001;
                    const double cxr = hull.cx - K.ox, cyr = hull.cy - K.oy;
                    const double wq = fma(cxr, cxr, fma(cyr, cyr, -re * re));
                    unsigned long long nm = 0ULL;
                    do {
                        const int k = __builtin_ctzll(cand);
                        cand &= cand - 1;
                        const double *q = hot + (size_t)k * FX_HOT_STRIDE;
                        const double g = fma(q[FX_HOT_HX2], cxr, fma(q[FX_HOT_HY2], cyr, fma(q[FX_HOT_HR2], re, q[FX_HOT_CK] + wq)));
                        nm |= (unsigned long long)wave_any_bit(!(g > K.gap_margin)) << k;
                    } while (cand);
                    FX_CSTAT(3, nm != 0); FX_CSTAT(4, __popcll(nm));
                    while (nm) {  // exact axis test for whatever is near (rare; the obstacle hull comes in through scalar loads)
                        const int k = __builtin_ctzll(nm);
                        nm &= nm - 1;
                        A.collided |= obb_overlap(hull, rec_i + k * 12 + 6);
                    }
                }
            }
            C.bx_prev = bx; C.by_prev = by; C.ux_prev = cu; C.uy_prev = su;
        }
        FX_STEP_STAMP(4);
    } else if (OBST && K.K > 0) {  // a batched launch may mix agents with and without obstacles
        const int nK = K.K;
        // per-step masks: one 64-bit word per 64 obstacles, word-major ([word][step]); more than one word only on this path
        // (the host sends agents with more than 64 obstacles to the generic kernel)
        const int nW = (nK + 63) >> 6;
        const int iu = USTEP ? __builtin_amdgcn_readfirstlane(i) : i;
        const auto rec_i = obs_rec + (int64_t)iu * nK * 12;
        // -- prediction cost: ego step i pairs with prediction i-1 (collision_probability.py:283-292) --
        for (int w = 0; w < nW; w++) {
            unsigned long long pm = emit ? obs_pmask[w * S + iu] : 0ULL;
            if (USTEP) pm = uniform_u64(pm);   // (the builtin returns int: without the helper's casts bit 31 sign-extends into 32 ... 63)
            if (USTEP && !emit) pm = 0ULL;  // emit is wave-uniform whenever the step index is
            while (pm) {
                const int k = __builtin_ctzll(pm) + 64 * w;
                pm &= pm - 1;
                const auto q = rec_i + k * 12;
                const double e0 = x_i - q[0], e1 = y_i - q[1];
                const double r0 = fma(e1, q[4], e0 * q[2]), r1 = fma(e1, q[5], e0 * q[3]);
                const double m = fma(r1, e1, r0 * e0);
                const double mm = m * m;
                A.pred += mm > 0.0 ? rcp_pred(mm) : 1.0 / mm;
            }
        }
        if (K.do_collision) {
            // ego box: centre = rear axle + wb_rear_axle along heading (state.py:30-39), heading theta_gl; it is needed
            // when this step closes a hull that meets an obstacle hull, or opens the next step's
            unsigned long long any_now = 0ULL, any_next = 0ULL;
            for (int w = 0; w < nW; w++) {
                any_now |= obs_hmask[w * S + iu];
                any_next |= iu + 1 < S ? obs_hmask[w * S + iu + 1] : 0ULL;
            }
            const bool boxed = (any_now | any_next) != 0ULL && i >= 1;
            double su = 0.0, cu = 0.0, bx = 0.0, by = 0.0;
            if (boxed) {
                heading_trig(r, cosTheta, tanTheta, cu, su);
                bx = fma(K.wb, cu, x_i); by = fma(K.wb, su, y_i);
            }
            if (neigh) {   // the left lane's box of step i - 1 (it built one whenever this step meets a hull: its `any_next` is this step's `any_now`)
                const double bx_l = __shfl_up(bx, 1), by_l = __shfl_up(by, 1), cu_l = __shfl_up(cu, 1), su_l = __shfl_up(su, 1);
                if (has_prev) { C.bx_prev = bx_l; C.by_prev = by_l; C.ux_prev = cu_l; C.uy_prev = su_l; }
            }
            if (boxed) {
                if (!emit) any_now = 0ULL;
                if (USTEP) any_now = uniform_u64(any_now);
                if (any_now) {
                    // OBB-sum hull of ego boxes (i-1, i) lives at time index i-1 and meets obstacle hull i-2
                    const Obb hull = obb_hull(C.bx_prev, C.by_prev, C.ux_prev, C.uy_prev, bx, by, cu, su, K.half_len, K.half_wid);
                    // broad phase: a box lies inside the circle around its centre with radius h1 + h2 (>= its half
                    // diagonal), so two boxes whose centres are farther apart than the sum of those radii are
                    // separated and the axis test below would say so.  The 1e-6 slack keeps every pair that is
                    // anywhere near touching (and every NaN) on the exact path, so decisions are unchanged.
                    const double re = (hull.h1 + hull.h2) * 1.000001;
                    for (int w = 0; w < nW; w++) {
                        unsigned long long hm = obs_hmask[w * S + iu];
                        if (USTEP) hm = uniform_u64(hm);
                        while (hm) {
                            const int k = __builtin_ctzll(hm) + 64 * w;
                            hm &= hm - 1;
                            const auto q = rec_i + k * 12 + 6;
                            const double tx = q[0] - hull.cx, ty = q[1] - hull.cy;
                            const double rr = fma(q[4] + q[5], 1.000001, re);
                            const bool near = !(fma(tx, tx, ty * ty) > rr * rr);
                            if (USTEP ? __any(near) : near) A.collided |= obb_overlap(hull, q);
                        }
                    }
                }
                C.bx_prev = bx; C.by_prev = by; C.ux_prev = cu; C.uy_prev = su;
            }
        }
    }
    // -- road boundary (planner.py:362-381): footprint rectangle vs the pieces binned at this step's
    //    reference segment; separating axes = the box axes and the piece's normal, touching intersects --
    if (OBST && K.n_bound > 0) {
        bool test = emit && (r.flags & LON_INDOMAIN) && A.bound_step == 0x7fffffff;
        bool hit = test && fabs(d_i) > K.bound_d_reach;
        test = test && !hit;
        if (__any(test)) {
            double su, cu;
            heading_trig(r, cosTheta, tanTheta, cu, su);
            const double cx = fma(K.wb, cu, x_i), cy = fma(K.wb, su, y_i);
            int j = test ? B.bin[r.pad0] : 0;
            const int j_end = test ? B.bin[r.pad0 + 1] : 0;
            while (__any(j < j_end)) {
                if (j < j_end) {
                    const FX_GLOBAL double *q = B.piece + 4 * (int64_t)B.item[j];
                    const double ex0 = q[0] - cx, ey0 = q[1] - cy;
                    const double ex = ex0 * cu + ey0 * su, ey = ey0 * cu - ex0 * su;
                    const double hx = q[2] * cu + q[3] * su, hy = q[3] * cu - q[2] * su;
                    const bool sep = fabs(ex) - (K.half_len + fabs(hx)) > 0 || fabs(ey) - (K.half_wid + fabs(hy)) > 0 ||
                                     fabs(ex * hy - ey * hx) - (K.half_len * fabs(hy) + K.half_wid * fabs(hx)) > 0;
                    if (!sep) { hit = true; j = j_end; }  // first hit of this step is enough
                    else j++;
                }
            }
        }
        if (hit) A.bound_step = i;
    }
    C.th_prev = th_gl;
    C.kap_prev = kap;
    O.a = a_i; O.v = v_i; O.th_cl = th_cl; O.x = x_i; O.y = y_i;
}

}  // namespace fxk
