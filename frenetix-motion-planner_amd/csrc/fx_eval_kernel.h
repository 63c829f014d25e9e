// fx_eval_kernel.h -- the fused evaluation kernel (included by fx_kernels.hip).
//
// Work decomposition.  A candidate's horizon of S samples is cut into G contiguous chunks ("parts") that are
// walked by G adjacent lanes; G = 1 ... 32 is chosen on the host from the candidate count:
//   * G = 1 -- one lane per candidate.  Zero redundancy; right once there are enough candidates to fill the
//     chip with waves (256 CUs x 4 SIMDs want >= 2-4 waves each, i.e. >= 130k-260k candidates).
//   * G > 1 -- the 50k-candidate plan step of BASELINE config 2 is only 788 waves at G = 1 (< 1 per SIMD),
//     which leaves every dependent FP64 chain, LDS lookup and store exposed.  Splitting the horizon gives
//     G x the waves for (CH+1)/CH x the arithmetic (each part re-evaluates the one step before its chunk to
//     obtain theta/kappa/box of step i-1 for the finite differences and the OBB-sum hull).
// Every quantity of step i needs only step i-1 of the same candidate, with two exceptions that are handled
// explicitly: the horizon-extension recurrence s[i] = s[i-1] + dt*s_dot_end (re-run from traj_len-1 by the
// part that starts inside the extension) and the standstill heading carry (reactive_planner.py:447), for
// which a part scans back to the last moving step.  Per-candidate results (masks, sums, reasons) are combined
// across the G lanes with wave shuffles; the part-0 lane owns the candidate's outputs.
//
// Memory.  plane[p][step][candidate]: in one store instruction the wave writes G row segments of 64/G
// consecutive candidates (>= 128 B for G <= 4).  Reference knots (AoS, 64 B) and the time-power table live in
// LDS; obstacle arrays are read from global memory (wave-uniform => scalar loads when G = 1).
#pragma once

#include "fx_device.h"
#include "fx_math.h"
#include "fx_tail.h"

namespace fxk {

#define FX_EPS 1e-5
#define FX_TWO_PI 6.283185307179586

struct Knot {  // one reference knot as staged in LDS
    double pos, theta, curv, curv_d, x, y, nx, ny;
};

__device__ __forceinline__ double wrap_pm_2pi(double a) {
    // commonroad make_valid_orientation (restated): bring into [-2pi, 2pi]
    while (a > FX_TWO_PI) a -= FX_TWO_PI;
    while (a < -FX_TWO_PI) a += FX_TWO_PI;
    return a;
}

// a / b given r = RN(1/b): one Newton correction on the quotient (Markstein).  Correctly rounded for the
// finite, normal operands of this kernel; 3 FP64 ops instead of the ~12-instruction IEEE division sequence.
__device__ __forceinline__ double div_rcp(double a, double b, double r) {
    const double q = a * r;
    const double e = fma(-b, q, a);
    return fma(e, r, q);
}

// np.round(x, 5) = rint(x * 1e5) / 1e5
__device__ __forceinline__ double np_round5(double x) { return div_rcp(rint(x * 1e5), 1e5, 1e-5); }

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

// OBB-sum hull of two boxes with UNIT headings u0, u1, equal half extents (hl, hw) and centres c0, c1: axis
// e = normalize(u0 + u1), extents = tight range of both boxes on (e, f).  Because e bisects the two headings, both boxes have the
// same extent along either hull axis: |u_j . e| = |u0 + u1| / 2 and |u_j x e| = |u0 x u1| / |u0 + u1| for j = 0 and 1.  The tight
// range on e is therefore [min_j c_j . e - r1, max_j c_j . e + r1], i.e. the hull's centre is the midpoint of the two centres
// and its half extent |(c1 - c0) . e| / 2 + r1 (likewise on f) -- ~35 operations where forming both boxes' ranges and merging
// them took ~70 (as many as the twenty prediction visits of the step).  The unit axis comes from v_rsq_f64 + two coupled Newton
// steps (sqrt_rsqrt in fx_walk.h).  Differs from the oracle's box-by-box form by an ulp or two; decisions are unaffected
// (the oracle reports every collision decision closer than 1e-9 to its threshold as fragile, tests/admissible.py).
__device__ __forceinline__ void sqrt_rsqrt(double x, double &sq, double &rsq);
__device__ __forceinline__ Obb obb_hull(double c0x, double c0y, double u0x, double u0y, double c1x, double c1y,
                                        double u1x, double u1y, double hl, double hw) {
    const double mx = u0x + u1x, my = u0y + u1y;
    double mn, r_mn;
    sqrt_rsqrt(fma(mx, mx, my * my), mn, r_mn);
    const bool flat = !(mn >= 1e-12);   // opposite headings (or a non-finite one): the first box's axis
    Obb o;
    o.ex = flat ? u0x : mx * r_mn;
    o.ey = flat ? u0y : my * r_mn;
    const double a = flat ? 1.0 : 0.5 * mn;                                    // |u_j . e|
    const double b = flat ? 0.0 : fabs(fma(u0x, u1y, -(u0y * u1x))) * r_mn;    // |u_j x e|
    const double tx = c1x - c0x, ty = c1y - c0y;
    o.cx = fma(0.5, tx, c0x);
    o.cy = fma(0.5, ty, c0y);
    o.h1 = fma(0.5, fabs(fma(tx, o.ex, ty * o.ey)), fma(hl, a, hw * b));
    o.h2 = fma(0.5, fabs(fma(ty, o.ex, -(tx * o.ey))), fma(hl, b, hw * a));
    return o;
}

// separating-axis test; b = (cx, cy, ex, ey, h1, h2)
template <typename PtrT>
__device__ __forceinline__ bool obb_overlap(const Obb &a, PtrT b) {
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

// m[lane K] = n (n wave-uniform): v_writelane_b32, one instruction
template <int K>
__device__ __forceinline__ void write_lane(unsigned int &m, int n) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(m) : "s"(n), "n"(K));
}

// reductions over the G adjacent lanes that share a candidate
template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off);
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_or(uint32_t v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) v |= (uint32_t)__shfl_xor((int)v, off);
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_min(uint32_t v) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1) { uint32_t o = (uint32_t)__shfl_xor((int)v, off); v = o < v ? o : v; }
    return v;
}


}  // namespace fxk

#include "fx_walk.h"

namespace fxk {

// Per-lane results of the horizon walk, handed to the shared epilogue.
// lane_center_offset, one point (partial_cost_functions.py:106-115; fxplan.h FxProblem.n_lane is the normative definition): the
// first lanelet whose outline contains the point -- bounding box, then ray casting over the closed outline -- and the distance
// to the closest point of its centre polyline; 5 when no lanelet contains it.  Every lane walks the lanelets on its own (the
// generic kernel's windowed-cost path: one candidate per lane).
template <typename PR>
__device__ __forceinline__ double lane_center_distance(const PR &P, double x, double y) {
    const FX_GLOBAL double *__restrict__ bbox = as_global(P.lane_bbox);
    const FX_GLOBAL int32_t *__restrict__ poff = as_global(P.lane_poly_off);
    const FX_GLOBAL double *__restrict__ poly = as_global(P.lane_poly);
    const FX_GLOBAL int32_t *__restrict__ coff = as_global(P.lane_ctr_off);
    const FX_GLOBAL double *__restrict__ ctr = as_global(P.lane_ctr);
    for (int l = 0; l < P.n_lane; l++) {
        if (x < bbox[4 * l] || x > bbox[4 * l + 1] || y < bbox[4 * l + 2] || y > bbox[4 * l + 3]) continue;
        const int v0 = poff[l], v1 = poff[l + 1];
        bool odd = false;
        double xj = poly[2 * (v1 - 1)], yj = poly[2 * (v1 - 1) + 1];   // edge k runs from vertex k-1 to vertex k
        for (int k = v0; k < v1; k++) {
            const double xi = poly[2 * k], yi = poly[2 * k + 1];
            if ((yi > y) != (yj > y)) {
                const double t = (xj - xi) * (y - yi) / (yj - yi) + xi;
                if (x < t) odd = !odd;
            }
            xj = xi; yj = yi;
        }
        if (!odd) continue;
        const int c0 = coff[l], c1 = coff[l + 1];
        double best = INFINITY;
        for (int k = c0; k + 1 < c1; k++) {
            const double ax = ctr[2 * k], ay = ctr[2 * k + 1];
            const double bx = ctr[2 * k + 2] - ax, by = ctr[2 * k + 3] - ay;
            const double len2 = bx * bx + by * by;
            double t = len2 > 0.0 ? ((x - ax) * bx + (y - ay) * by) / len2 : 0.0;
            t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
            const double ex = x - (ax + t * bx), ey = y - (ay + t * by);
            const double d2 = ex * ex + ey * ey;
            if (d2 < best) best = d2;
        }
        if (c1 - c0 == 1) {
            const double ex = x - ctr[2 * c0], ey = y - ctr[2 * c0 + 1];
            best = ex * ex + ey * ey;
        }
        return best < INFINITY ? sqrt(best) : 5.0;
    }
    return 5.0;
}

struct WalkResult {
    bool neg, acc_viol, collided;
    uint32_t step_reasons;  // dbg: OR of all violated checks
    uint32_t first_key;     // !dbg: (step << 4 | reason) of the first violated check, 0xffffffff if none
    int fail_step;          // first step of this lane's chunk outside the projection domain, INT_MAX if none
    int bound_step;         // first step of this lane's chunk whose footprint meets the road boundary, INT_MAX if none
    double sum_abs_d, sum_voff, pred, dto, lane_off, d_end, v_end;
    double cl3, cl4, cl5, ct3, ct4, ct5;
    Simpson sim_acc, sim_jerk, sim_orient, sim_path;
};

// Combine the G lanes of a candidate, assemble the flag word exactly as check_feasibility does, form the weighted
// cost, write the per-candidate outputs and contribute to the workgroup's counters and (cost, index) arg-min.
//
// The G parts of a candidate are either G adjacent lanes (WSPLIT = false: combined with wave shuffles) or the same
// lane of G different lane-groups of the workgroup (WSPLIT = true: part = tid / CPB, combined through the LDS block
// `xch` of G * CPB * 56 bytes).  The second form keeps every store of the walk a contiguous row segment per wave.
template <int G, bool BUNDLE, bool OBST, bool EXTRA, bool WSPLIT = false, bool MEGA = false>
__device__ __forceinline__ void finish_candidate(const ProblemRegs &P, const DevProblem &Pg, WalkResult &W, int64_t g, bool active, int part,
                                                 int i_begin, int i_end, bool bundle, bool do_collision, bool dbg, bool D,
                                                 double *red_cost, long long *red_idx, unsigned int *red_cnt,
                                                 const FuseArgs &fuse, double *xch = nullptr, int CPB = 0,
                                                 int cand_local = 0) {
    const int tid = threadIdx.x;
    const int S = P.S, K = P.K, Pn = P.P;
    const int64_t ld = P.ld;
    const double dt = P.dt;
    // the agent's last workgroup ends the step itself (fx_tail.h): what it reads of other workgroups' outputs is stored write-through
    // Compiled into the planner-sized decompositions only (four or more lanes per candidate: below 200 waves; and the windowed-cost
    // kernel): the kernels of the large grids stay exactly what they were -- their register allocation is tuned to the last VGPR.
    constexpr bool TAIL = FX_TAIL_IN_KERNEL(G, EXTRA);
    // MEGA (fx_step_kernel.h: the whole step in one launch): later phases of the SAME launch read what this one stores -- write-through
    const bool tail_mode = TAIL && fuse.host_result != nullptr && fuse.tail() != 0u;
    const bool tail_pkg = MEGA || (tail_mode && (fuse.tail() & FX_TAIL_PACKAGE));
    const bool out_wt = MEGA || tail_mode;
    FX_GLOBAL double *__restrict__ planes = as_global(P.planes);
    const FX_GLOBAL double *__restrict__ obs_pos = as_global(P.obs_pos);
    const FX_GLOBAL double *__restrict__ obs_cov_inv = as_global(P.obs_cov_inv);
    const FX_GLOBAL int32_t *__restrict__ obs_npred = as_global(P.obs_npred);
    bool neg = W.neg, acc_viol = W.acc_viol, collided = W.collided;
    uint32_t step_reasons = W.step_reasons, first_key = W.first_key;
    int fail_step = W.fail_step;
    uint32_t bound_step = (uint32_t)W.bound_step;
    double sum_abs_d = W.sum_abs_d, sum_voff = W.sum_voff, pred = W.pred, dto = W.dto, lane_off = W.lane_off, d_end = W.d_end, v_end = W.v_end;
    const double cl3 = W.cl3, cl4 = W.cl4, cl5 = W.cl5, ct3 = W.ct3, ct4 = W.ct4, ct5 = W.ct5;
    Simpson &sim_acc = W.sim_acc, &sim_jerk = W.sim_jerk, &sim_orient = W.sim_orient, &sim_path = W.sim_path;

    // ---- combine the G parts of a candidate ----
    uint32_t bits = (neg ? 1u : 0u) | (acc_viol ? 2u : 0u) | (collided ? 4u : 0u);
    if (G > 1) {
        uint32_t fail_all;
        uint32_t *xu = reinterpret_cast<uint32_t *>(xch + 5 * G * CPB);  // [4][G*CPB] after the five double planes
        const int slot = part * CPB + cand_local, n_slot = G * CPB;
        if (!WSPLIT) {
            bits = group_or<G>(bits);
            step_reasons = group_or<G>(step_reasons);
            first_key = group_min<G>(first_key);
            fail_all = group_min<G>((uint32_t)fail_step);
            if (OBST) bound_step = group_min<G>(bound_step);
        } else {
            xu[slot] = (uint32_t)fail_step;
            __syncthreads();
            fail_all = 0x7fffffffu;
#pragma unroll
            for (int q = 0; q < G; q++) { const uint32_t f = xu[q * CPB + cand_local]; fail_all = f < fail_all ? f : fail_all; }
        }
        if (fail_all != 0x7fffffffu && (int)fail_all < i_begin) {
            // an earlier part left the projection domain: every later (x, y) is 0 (the reference's loop breaks, :547)
            if (bundle && active) {
                const int64_t ps = (int64_t)S * ld;
                for (int i = i_begin; i < i_end; i++) {
                    FX_GLOBAL double *__restrict__ row = planes + (int64_t)i * ld + g;
                    st_out(row + FX_PL_X * ps, 0.0, (P.mode & FX_MODE_INT_STORE_WT) != 0);   // (store mode of the walk's plane stores)
                    st_out(row + FX_PL_Y * ps, 0.0, (P.mode & FX_MODE_INT_STORE_WT) != 0);
                }
            }
            if (OBST) {
                pred = 0.0;
                for (int i = max(i_begin, 1); i < i_end; i++)
                    for (int k = 0; k < K; k++)
                        if (i < obs_npred[k] && i - 1 < Pn) {   // npred is the real length of the prediction, Pn what is stored
                            const FX_GLOBAL double *__restrict__ mu = obs_pos + ((int64_t)k * Pn + (i - 1)) * 2;
                            const FX_GLOBAL double *__restrict__ iv = obs_cov_inv + ((int64_t)k * Pn + (i - 1)) * 4;
                            const double e0 = 0.0 - mu[0], e1 = 0.0 - mu[1];
                            const double r0 = fma(e1, iv[2], e0 * iv[0]), r1 = fma(e1, iv[3], e0 * iv[1]);
                            const double m = fma(r1, e1, r0 * e0);
                            const double mm = m * m;
                            pred += mm > 0.0 ? rcp_pred(mm) : 1.0 / mm;
                        }
            }
        }
        fail_step = (int)fail_all;
        if (!WSPLIT) {
            sum_abs_d = group_sum<G>(sum_abs_d);
            sum_voff = group_sum<G>(sum_voff);
            if (OBST) pred = group_sum<G>(pred);
            d_end = group_sum<G>(d_end);  // only the part that owns step S-1 holds a non-zero value
            v_end = group_sum<G>(v_end);
        } else {
            xch[0 * n_slot + slot] = sum_abs_d;
            xch[1 * n_slot + slot] = sum_voff;
            xch[2 * n_slot + slot] = pred;
            xch[3 * n_slot + slot] = d_end;
            xch[4 * n_slot + slot] = v_end;
            xu[1 * n_slot + slot] = bits;
            xu[2 * n_slot + slot] = step_reasons;
            xu[3 * n_slot + slot] = first_key;
            if (OBST) xu[4 * n_slot + slot] = bound_step;
            __syncthreads();
            if (part == 0) {
                // same association as the xor-shuffle tree: ((p0 + p1) + (p2 + p3)) + ...
                auto tree = [&](int plane) {
                    double v[G];
#pragma unroll
                    for (int q = 0; q < G; q++) v[q] = xch[plane * n_slot + q * CPB + cand_local];
#pragma unroll
                    for (int w = 1; w < G; w <<= 1)
#pragma unroll
                        for (int q = 0; q < G; q += 2 * w) v[q] += v[q + w];
                    return v[0];
                };
                sum_abs_d = tree(0);
                sum_voff = tree(1);
                if (OBST) pred = tree(2);
                d_end = tree(3);
                v_end = tree(4);
#pragma unroll
                for (int q = 1; q < G; q++) {
                    bits |= xu[1 * n_slot + q * CPB + cand_local];
                    step_reasons |= xu[2 * n_slot + q * CPB + cand_local];
                    const uint32_t fk = xu[3 * n_slot + q * CPB + cand_local];
                    first_key = fk < first_key ? fk : first_key;
                    if (OBST) {
                        const uint32_t bs = xu[4 * n_slot + q * CPB + cand_local];
                        bound_step = bs < bound_step ? bs : bound_step;
                    }
                }
            }
        }
    }
    neg = bits & 1u; acc_viol = bits & 2u; collided = bits & 4u;
    // a boundary hit after the projection left its domain does not count (the reference's loop has stopped, :537-547)
    const bool off_road = OBST && bound_step != 0x7fffffffu && (int)bound_step < fail_step;
    if (!dbg) step_reasons = first_key == 0xffffffffu ? 0u : (1u << (first_key & 15u));
    const bool proj_ok = fail_step == 0x7fffffff;
    const bool leader = part == 0;

    FX_STAMP(5);
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
    if (selectable && off_road) flags |= FX_FLAG_BOUNDARY;
    flags |= reasons << FX_REASON_SHIFT;

    FX_STAMP(6);
    // Deferred obstacle stage: the costed candidates of this wave go on the agent's list -- the obstacle kernel visits the list,
    // not the grid (a third of a production step's candidates is infeasible and has neither a prediction cost nor a collision
    // check, reactive_planner.py:480-533 / cost_function.py:78-91).  One returning atomic per wave, requested here and consumed
    // behind the cost sum; the order of the list varies from run to run, nothing that is computed from it does.
    unsigned long long live_mask = 0ULL;
    unsigned long long list_base = 0ULL;
    if (!OBST && (P.mode & FX_MODE_INT_DEFER_OBST)) {
        live_mask = __ballot(active && leader && costed);
        if (live_mask) {
            unsigned long long b = 0ULL;
            if ((tid & 63) == __ffsll((long long)live_mask) - 1)
                b = __hip_atomic_fetch_add(as_global(P.counters) + FX_DCNT_LIVE, (unsigned long long)__popcll(live_mask), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            list_base = b;   // (read back from the issuing lane below)
        }
    }
    // ---- weighted cost sum in name-sorted order (cost_function.py:78-91) ----
    // Deferred obstacle stage (FX_MODE_INT_DEFER_OBST, only in kernels built without the stage): the prediction term is not
    // known yet -- cost[] receives the running sum in front of it, cost_tail[] the weighted terms behind it (by the ids'
    // order that is velocity_offset alone), and fx_obstacle_kernel continues the same sequence of additions.
    const bool defer = !OBST && (P.mode & FX_MODE_INT_DEFER_OBST) != 0;
    bool have_pre = false;
    double pre = 0.0, tail = 0.0;
    double total = 0.0;
    {
        const double tt = dt, tt2 = tt * tt, tt3 = tt2 * tt, tt4 = tt3 * tt, tt5 = tt4 * tt;
        const int n_cost = P.n_cost;
        double sum = -0.0;
        // ids / weights come from LDS: fetch entry n+1 while term n is computed
        int id_next = n_cost > 0 ? P.cost_id[0] : 0;
        double w_next = n_cost > 0 ? P.cost_w[0] : 0.0;
        for (int n = 0; n < n_cost; n++) {
            const int id = id_next;
            const double w = w_next;
            if (n + 1 < n_cost) { id_next = P.cost_id[n + 1]; w_next = P.cost_w[n + 1]; }
            double c = 0.0;
            switch (id) {
            case FX_COST_DISTANCE_TO_REFERENCE_PATH: c = ((0.0 + sum_abs_d) + fabs(d_end) * 5) / S; break;
            case FX_COST_LATERAL_JERK:  // squared_jerk_integral(dt) (polynomial_trajectory.py:172-191, cost :54)
                c = (36 * ct3 * ct3 * tt + 144 * ct3 * ct4 * tt2 + 240 * ct3 * ct5 * tt3 + 192 * ct4 * ct4 * tt3 +
                     720 * ct4 * ct5 * tt4 + 720 * ct5 * ct5 * tt5);
                break;
            case FX_COST_LONGITUDINAL_JERK:
                c = (36 * cl3 * cl3 * tt + 144 * cl3 * cl4 * tt2 + 240 * cl3 * cl5 * tt3 + 192 * cl4 * cl4 * tt3 +
                     720 * cl4 * cl5 * tt4 + 720 * cl5 * cl5 * tt5);
                break;
            case FX_COST_VELOCITY_OFFSET: {
                const double e = v_end - P.v_des;
                c = (0.0 + sum_voff) + fabs(e * e);
                break;
            }
            case FX_COST_PREDICTION: c = OBST ? pred : 0.0; break;
            case FX_COST_ACCELERATION: c = EXTRA ? sim_acc.finish(S, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_PATH_LENGTH: c = EXTRA ? sim_path.finish(S, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_JERK: c = EXTRA ? sim_jerk.finish(S - 1, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_ORIENTATION_OFFSET: c = EXTRA ? sim_orient.finish(S - 1, dt, P.simpson_corr) : 0.0; break;
            case FX_COST_DISTANCE_TO_OBSTACLES: c = EXTRA ? dto : 0.0; break;
            case FX_COST_LANE_CENTER_OFFSET: c = EXTRA ? lane_off / S : 0.0; break;   // partial_cost_functions.py:117
            default: break;
            }
            if (defer && id == FX_COST_PREDICTION) {
                pre = sum; have_pre = true;
                // (the obstacle kernel fills the cost-map entry of the candidates it visits -- the costed ones)
                if ((P.mode & FX_MODE_WRITE_COSTMAP) && active && leader && !costed) st_out(as_global(P.costmap) + (int64_t)n * ld + g, 0.0, tail_pkg);
                continue;
            }
            if ((P.mode & FX_MODE_WRITE_COSTMAP) && active && leader) st_out(as_global(P.costmap) + (int64_t)n * ld + g, costed ? c : 0.0, tail_pkg);
            if (have_pre) tail += w * c;
            sum += w * c;
        }
        total = 0.0 + sum;
    }
    if (live_mask) {
        const int src = __ffsll((long long)live_mask) - 1;
        const unsigned lo = (unsigned)__shfl((int)(unsigned)(list_base & 0xffffffffULL), src);   // (lists stay below 2^31 entries)
        if (active && leader && costed)
            st_out(as_global(P.obs_list) + (lo + __popcll(live_mask & ((1ULL << (tid & 63)) - 1ULL))), (int32_t)g, MEGA);
    }
    if (active && leader) {
        if (defer) st_out(as_global(P.cost_tail) + g, tail, MEGA);
        st_out(as_global(P.cost) + g, costed ? (have_pre ? pre : total) : 0.0, out_wt);
        st_out(as_global(P.flags) + g, flags, out_wt);
        if (OBST && (P.mode & FX_MODE_ROAD_BOUNDARY)) as_global(P.bound_step)[g] = (selectable && off_road) ? (int)bound_step : -1;
    }

    FX_STAMP(7);
    // ---- workgroup reductions (candidate leaders only): counters and the (cost, index) arg-min partial ----
    const int lane = tid & 63, wave = tid >> 6;
    const bool own = active && leader;
    {
        // lane k of the wave collects counter k (scalar popcounts dropped into a lane each), one LDS atomic per wave
        unsigned int mine = 0;
        const bool ret = own && (flags & FX_FLAG_RETURNED);
        write_lane<0>(mine, wave_count(ret));
        write_lane<1>(mine, wave_count(ret && (flags & FX_FLAG_VALID) && (flags & FX_FLAG_FEASIBLE)));
        static_assert(FX_NUM_REASONS == 11, "one write_lane per reason below");
#define FX_REASON_LANE(r) write_lane<2 + r>(mine, wave_count(own && ((reasons >> r) & 1u)))
        FX_REASON_LANE(0); FX_REASON_LANE(1); FX_REASON_LANE(2); FX_REASON_LANE(3); FX_REASON_LANE(4); FX_REASON_LANE(5);
        FX_REASON_LANE(6); FX_REASON_LANE(7); FX_REASON_LANE(8); FX_REASON_LANE(9); FX_REASON_LANE(10);
#undef FX_REASON_LANE
        if (lane < 2 + FX_NUM_REASONS && mine) atomicAdd(&red_cnt[lane], mine);
    }
    FX_STAMP(8);
    const bool eligible = own && selectable && !(flags & (FX_FLAG_COLLISION | FX_FLAG_BOUNDARY)) && total == total;
    double bc = eligible ? total : INFINITY;
    long long bi = eligible ? (long long)(g + P.g_base) : 0x7fffffffffffffffLL;
    // wave arg-min: candidate indices grow with the lane, so the (cost, index) minimum is the LOWEST lane that
    // holds the minimum cost -- a min-reduction of the cost alone plus one ballot
    {
        double m = bc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off));
        const unsigned long long hit = __ballot(eligible && bc == m);
        const int src = hit ? __ffsll((long long)hit) - 1 : 0;
        bc = m;
        bi = hit ? __shfl(bi, src) : 0x7fffffffffffffffLL;
    }
    if (lane == 0) { red_cost[wave] = bc; red_idx[wave] = bi; }
    FX_STAMP(9);
    // tail mode: this wave's cost / flag / coefficient / plane stores are performed before the workgroup takes its ticket
    if (tail_mode) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Only the first wave goes on (it owns the lanes that publish); the others are done -- or, tail mode, wait to learn whether
    // theirs is the agent's last workgroup.
    if (wave != 0 && !tail_mode) return;
    if (wave == 0) {
        for (int w = 1; w < (int)blockDim.x / 64; w++)
            if (red_cost[w] < bc || (red_cost[w] == bc && red_idx[w] < bi)) { bc = red_cost[w]; bi = red_idx[w]; }
        if (lane == 0 && !defer) {   // deferred obstacle stage: fx_obstacle_kernel owns this agent's partials
            // agent-scope stores: visible to whichever XCD runs the reducing workgroup without an L2 write-back
            __hip_atomic_store(as_global(P.part_cost) + blockIdx.x, bc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(as_global(P.part_idx) + blockIdx.x, (int64_t)bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane < 2 + FX_NUM_REASONS && red_cnt[lane]) atomicAdd(&P.counters[lane], (unsigned long long)red_cnt[lane]);
        FX_STAMP(10);
        if (fuse.host_result == nullptr) return;  // a selection kernel follows

        // ---- fused selection: the last workgroup of this agent to arrive reduces and publishes ----
        // Partials and counters above are agent-scope atomics issued by THIS wave; once they are acknowledged
        // (vmcnt 0) they are performed, so a relaxed ticket is enough -- no L2 write-back in the way of the bundle's
        // store stream, no workgroup barrier.  (The other waves' cost / flag / plane stores may still be in flight
        // when the result is published: every consumer of those is ordered behind the kernel on the stream.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FX_STAMP(11);
        unsigned long long ticket = 0;
        if (lane == 0) ticket = atomicAdd(&P.counters[FX_DCNT_TICKET], 1ULL);
        ticket = __shfl(ticket, 0);
        FX_STAMP(12);
        FX_STAMP(15);
        if (TAIL && tail_mode) {   // every wave of the agent's last workgroup runs the tail (fx_tail.h): tell the others
            if (lane == 0) red_cnt[2 + FX_NUM_REASONS] = ticket == (unsigned long long)(P.n_blocks - 1) ? 1u : 0u;
        } else {
            if (ticket != (unsigned long long)(P.n_blocks - 1)) return;
            unsigned long long *out = fuse.host_result + (size_t)blockIdx.y * (FX_CNT_COUNT + 1);
            // counters: read and zero in one agent-scope exchange (the next step starts from a clean block); issued first so
            // that its round trip overlaps the loads of the partials
            unsigned long long cnt = 0ULL;
            if (lane < FX_CNT_BEST_IDX) cnt = atomicExch(&P.counters[lane], 0ULL);
            bc = INFINITY;
            bi = 0x7fffffffffffffffLL;
            // device-coherent loads cross the fabric (~1 us each): eight per lane in flight before the first comparison
            for (int b0 = lane; b0 < P.n_blocks; b0 += 8 * 64) {
                double c[8];
                long long ix[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int b = min(b0 + u * 64, P.n_blocks - 1);   // a repeated entry does not change the minimum
                    c[u] = __hip_atomic_load(as_global(P.part_cost) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ix[u] = __hip_atomic_load(as_global(P.part_idx) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (c[u] < bc || (c[u] == bc && ix[u] < bi)) { bc = c[u]; bi = ix[u]; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double oc = __shfl_xor(bc, off);
                const long long oi = __shfl_xor(bi, off);
                if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
            }
            const bool none = bi == 0x7fffffffffffffffLL;
            {   // the result block in ONE store instruction: lanes 0 .. 12 the counters, 13 .. 15 winner index, cost bits, collisions
                unsigned long long w = cnt;
                if (lane == FX_CNT_BEST_IDX) w = none ? ~0ULL : (unsigned long long)bi;
                if (lane == FX_CNT_BEST_COST) w = none ? 0ULL : (unsigned long long)__double_as_longlong(bc);
                if (lane == FX_CNT_COLLISIONS) w = 0ULL;
                if (lane < FX_CNT_COUNT) put_host(out + lane, w);
            }
            if (lane == 0) {
                if (fuse.dev_winner) {
                    fuse.dev_winner[2 * blockIdx.y] = none ? INFINITY : bc;
                    reinterpret_cast<long long *>(fuse.dev_winner)[2 * blockIdx.y + 1] = none ? -1 : bi;
                }
                __hip_atomic_store(&P.counters[FX_DCNT_TICKET], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // every lane's result words have been acknowledged (the counter is per wave) before lane 0 sends the sequence
            // word behind them (no L2 write-back fence: fx_tail.h)
            drain_stores();
            if (lane == 0) st_host(out + FX_CNT_COUNT, fuse.seq);
            return;
        }   // (single-wave publication)
    }   // (wave 0)
    if (TAIL) {
        __syncthreads();
        if (!red_cnt[2 + FX_NUM_REASONS]) return;
        fx_fused_tail(P, Pg, fuse, (int)blockIdx.y);
    }
}

}  // namespace fxk

// ---------------------------------------------------------------------------------------------------
// Generic kernel: arbitrary sampling-matrix rows (or ranges), every lane derives its own longitudinal row.
// grid = (ceil(maxC / (256/G)), n_agents), block = 256, dynamic LDS = M*64 B + FX_TP*S*8 B.
//   G      : lanes per candidate (1, 2, 4, 8; 16, 32 at WPE = 2)
//   BUNDLE : write the 14-plane SoA TrajectoryBundle + coefficients (FX_MODE_WRITE_BUNDLE)
//   OBST   : obstacles present (prediction cost and/or collision stage)
//   EXTRA  : Simpson / distance_to_obstacles costs active (windowed over steps: G == 1 only)
//   WPE    : occupancy target in waves per SIMD handed to the register allocator
// ---------------------------------------------------------------------------------------------------
template <int G, bool BUNDLE, bool OBST, bool EXTRA, int WPE>
__global__ __launch_bounds__(FX_BLOCK, WPE) void fx_eval_kernel(const DevProblem *__restrict__ probs, const FuseArgs fuse) {
    using namespace fxk;
    static_assert(!EXTRA || G == 1, "windowed costs need the whole horizon in one lane");
    constexpr int CPB = FX_BLOCK / G;  // candidates per workgroup
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];  // [M][8] knots, then the [S][FX_TP] time table
    __shared__ double red_cost[FX_BLOCK / 64];
    __shared__ long long red_idx[FX_BLOCK / 64];
    __shared__ unsigned int red_cnt[2 + FX_NUM_REASONS + 1];  // counters, then the last-workgroup flag

    __shared__ int32_t sh_cost_id[FX_NUM_COSTS];
    __shared__ double sh_cost_w[FX_NUM_COSTS];

    const DevProblem &Pg = probs[blockIdx.y];
    const ProblemRegs P = ProblemRegs::load(Pg, sh_cost_id, sh_cost_w);  // one batch of scalar loads at entry
    const int tid = threadIdx.x;
    const int64_t C = P.C;
    if ((int64_t)blockIdx.x * CPB >= C) return;  // whole workgroup beyond this agent's grid
    const int part = G == 1 ? 0 : (tid & (G - 1));
    const int64_t g_raw = (int64_t)blockIdx.x * CPB + (G == 1 ? tid : tid / G);
    const bool active = g_raw < C;
    const int64_t g = active ? g_raw : C - 1;

    const int M = P.M, S = P.S;
    // planner-sized decompositions: the obstacle records of the step and the two step masks behind the time table (every lane
    // walks another step: from global memory a dependent round trip per visited obstacle; fx_eval_grid_kernel.h has the same)
    constexpr bool LSTAGE = OBST && G >= 4;
    const int rec_n = P.K > 0 ? S_rec_doubles(S, P.K) : 0;
    const bool rec_staged = LSTAGE && rec_n > 0 && (P.mode & FX_MODE_INT_REC_LDS) != 0;   // (the host sized the LDS for it)
    // the knots' arc lengths once more as a dense array: the segment lookup bisects it (on a non-uniform reference -- the
    // config-1 route: 0.1 ... 1.0 m between knots -- the uniform-spacing guess misses and every lane walks nine levels; through
    // the 64-byte knot records that is a 16-way bank conflict per level: 43 us for the 800-row matrix against 25 us on an arc)
    double *__restrict__ rpos = lds_dyn + (size_t)M * FX_REF_FIELDS + (size_t)FX_TP * S;   // [M], padded to an even count
    double *__restrict__ rec_lds = rpos + (((size_t)M + 1) & ~(size_t)1);
    unsigned long long *__restrict__ pm_lds = reinterpret_cast<unsigned long long *>(rec_lds + rec_n);   // [S] | [S]
    if constexpr (G >= 4) {
        // every table's first batch of loads is requested before the first LDS write (a copy loop is a round trip per iteration:
        // the 409 knots of the config-1 route cost thirteen of them)
        const FX_GLOBAL fx_d2 *__restrict__ src = reinterpret_cast<const FX_GLOBAL fx_d2 *>(as_global(P.ref));   // (256-byte aligned slots)
        fx_d2 *__restrict__ kdst = reinterpret_cast<fx_d2 *>(lds_dyn);
        const int nk2 = M * FX_REF_FIELDS / 2;
        fx_d2 kv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) kv[u] = src[min(tid + u * FX_BLOCK, nk2 - 1)];
        const FX_GLOBAL double *__restrict__ tsrc = as_global(P.tpow);
        const int it = min(tid, S - 1);
        const double t1 = tsrc[it], t2 = tsrc[S + it], t3 = tsrc[2 * S + it], t4 = tsrc[3 * S + it], t5 = tsrc[4 * S + it];
        const FX_GLOBAL fx_d2 *__restrict__ rsrc = reinterpret_cast<const FX_GLOBAL fx_d2 *>(as_global(P.obs_rec));
        fx_d2 *__restrict__ rdst = reinterpret_cast<fx_d2 *>(rec_lds);
        const int n2 = rec_n / 2;
        fx_d2 rv[4];
        unsigned long long m0 = 0ULL, m1 = 0ULL;
        if (rec_staged) {
#pragma unroll
            for (int u = 0; u < 4; u++) rv[u] = rsrc[min(tid + u * FX_BLOCK, n2 - 1)];
            m0 = as_global(P.obs_pmask)[it]; m1 = as_global(P.obs_hmask)[it];
        }
        int32_t cid = 0;
        double cw = 0.0;
        if (tid < P.n_cost) { cid = Pg.cost_id[tid]; cw = Pg.cost_w[tid]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = tid + u * FX_BLOCK;
            if (e < nk2) { kdst[e] = kv[u]; if ((e & 3) == 0) rpos[e >> 2] = kv[u].x; }   // (pair 4 k holds knot k's arc length)
        }
        if (tid < S) {
            fill_time_row(lds_dyn + M * FX_REF_FIELDS + tid * FX_TP, t1, t2, t3, t4, t5);
            if (rec_staged) { pm_lds[tid] = m0; pm_lds[S + tid] = m1; }
        }
        if (rec_staged) {
#pragma unroll
            for (int u = 0; u < 4; u++) if (tid + u * FX_BLOCK < n2) rdst[tid + u * FX_BLOCK] = rv[u];
        }
        if (tid < P.n_cost) { sh_cost_id[tid] = cid; sh_cost_w[tid] = cw; }
        for (int e0 = tid + 8 * FX_BLOCK; e0 < nk2; e0 += 8 * FX_BLOCK) {
#pragma unroll
            for (int u = 0; u < 8; u++) kv[u] = src[min(e0 + u * FX_BLOCK, nk2 - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * FX_BLOCK;
                if (e < nk2) { kdst[e] = kv[u]; if ((e & 3) == 0) rpos[e >> 2] = kv[u].x; }
            }
        }
        for (int i = tid + FX_BLOCK; i < S; i += FX_BLOCK) {
            fill_time_row(lds_dyn + M * FX_REF_FIELDS + i * FX_TP, tsrc[i], tsrc[S + i], tsrc[2 * S + i], tsrc[3 * S + i], tsrc[4 * S + i]);
            if (rec_staged) { pm_lds[i] = as_global(P.obs_pmask)[i]; pm_lds[S + i] = as_global(P.obs_hmask)[i]; }
        }
        if (rec_staged) {
            for (int e0 = tid + 4 * FX_BLOCK; e0 < n2; e0 += 4 * FX_BLOCK) {
#pragma unroll
                for (int u = 0; u < 4; u++) rv[u] = rsrc[min(e0 + u * FX_BLOCK, n2 - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) if (e0 + u * FX_BLOCK < n2) rdst[e0 + u * FX_BLOCK] = rv[u];
            }
        }
    } else {
        const FX_GLOBAL double *__restrict__ src = as_global(P.ref);
        for (int i = tid; i < M * FX_REF_FIELDS; i += FX_BLOCK) {
            const double v = src[i];
            lds_dyn[i] = v;
            if ((i & (FX_REF_FIELDS - 1)) == 0) rpos[i / FX_REF_FIELDS] = v;
        }
        const FX_GLOBAL double *__restrict__ tsrc = as_global(P.tpow);
        for (int i = tid; i < S; i += FX_BLOCK)
            fill_time_row(lds_dyn + M * FX_REF_FIELDS + i * FX_TP, tsrc[i], tsrc[S + i], tsrc[2 * S + i], tsrc[3 * S + i], tsrc[4 * S + i]);
        if (tid < P.n_cost) { sh_cost_id[tid] = Pg.cost_id[tid]; sh_cost_w[tid] = Pg.cost_w[tid]; }
    }
    if (tid < 2 + FX_NUM_REASONS) red_cnt[tid] = 0;
    __syncthreads();
    const Knot *__restrict__ knots = reinterpret_cast<const Knot *>(lds_dyn);
    const double *__restrict__ tp = lds_dyn + M * FX_REF_FIELDS;

    const double dt = P.dt;
    const bool low_vel = P.low_vel_mode != 0;
    const bool D = (P.mode & FX_MODE_DRAW_TRAJ_SET) != 0;
    const bool dbg = D || (P.mode & FX_MODE_KINEMATIC_DEBUG) != 0;
    const bool do_collision = OBST && (P.mode & FX_MODE_COLLISION) != 0;
    // a batch launch is specialised for the union of its agents' modes; each agent still honours its own
    const bool bundle = BUNDLE && (P.mode & FX_MODE_WRITE_BUNDLE) != 0;
    const double a_max = P.veh.a_max;
    const int64_t ld = P.ld;

    // ---- candidate parameters (reactive_planner.py:149-171 / sampling matrix row) ----
    double T, s0, ss0, sss0, v1, a1, d0, dd0, ddd0, d1, dd1, ddd1;
    if (P.has_matrix) {
        const FX_GLOBAL double *__restrict__ r = as_global(P.matrix) + 13 * (g + P.g_base);
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
        T = as_global(P.t_samp)[it];
        v1 = as_global(P.v_samp)[iv];
        d1 = as_global(P.d_samp)[id];
        s0 = P.x0_lon[0]; ss0 = P.x0_lon[1]; sss0 = P.x0_lon[2];
        d0 = P.x0_lat[0]; dd0 = P.x0_lat[1]; ddd0 = P.x0_lat[2];
        a1 = 0.0; dd1 = 0.0; ddd1 = 0.0;
    }

    // ---- longitudinal quartic (polynomial_trajectory.py:452-488, closed form of the 2x2 solve) ----
    double cl0 = s0, cl1 = ss0, cl2 = .5 * sss0, cl3, cl4, cl5;
    lon_coeffs(P.has_matrix ? FX_LON_VELOCITY_KEEPING : P.lon_mode, s0, ss0, sss0, T, v1, a1, cl3, cl4, cl5);
    // ---- lateral quintic over time (high speed) or arclength (LOW_VEL_MODE), reactive_planner.py:161-171 ----
    double tau = T;
    if (low_vel) {
        double t2 = T * T, t3 = t2 * T, t4 = t2 * t2, t5 = t3 * t2;
        double s_lon_goal = (cl0 + cl1 * T + cl2 * t2 + cl3 * t3 + cl4 * t4 + cl5 * t5) - s0;
        if (s_lon_goal <= 0) s_lon_goal = T;
        tau = s_lon_goal;
    }
    LatPoly L;
    {
        double T2 = tau * tau, T3 = T2 * tau, T4 = T3 * tau, T5 = T4 * tau;
        double b0 = d1 - d0 - dd0 * tau - .5 * ddd0 * T2;
        double b1 = dd1 - dd0 - ddd0 * tau;
        double b2 = ddd1 - ddd0;
        L.set(d0, dd0, .5 * ddd0, fdiv(10.0 * b0 - 4.0 * b1 * tau + .5 * b2 * T2, T3),
              fdiv(-15.0 * b0 + 7.0 * b1 * tau - b2 * T2, T4), fdiv(6.0 * b0 - 3.0 * b1 * tau + .5 * b2 * T2, T5));
    }
    // len(np.arange(0, T+dt, dt)) (reactive_planner.py:296,303), clamped to the horizon
    int traj_len = (int)ceil((T + dt) / dt);
    traj_len = traj_len > S ? S : (traj_len < 1 ? 1 : traj_len);

    if (bundle && active && part == 0) {
        FX_GLOBAL double *__restrict__ co = as_global(P.coeffs) + g;
        // (write-through where the step's last workgroup gathers the winner package itself, fx_tail.h)
        const bool wt = FX_TAIL_IN_KERNEL(G, EXTRA) && fuse.host_result != nullptr && (fuse.tail() & FX_TAIL_PACKAGE);
        const double cv[FX_COEFF_ROWS] = {cl0, cl1, cl2, cl3, cl4, cl5, L.c0, L.c1, L.c2, L.c3, L.c4, L.c5,
                                          tau};  // tau: PolynomialTrajectory.delta_tau of the lateral polynomial (reactive_planner.py:161-171)
#pragma unroll
        for (int q = 0; q < FX_COEFF_ROWS; q++) st_out(co + q * ld, cv[q], wt);
        st_out(as_global(P.traj_len) + g, (int32_t)traj_len, wt);
    }

    const double rp_first = knots[0].pos, rp_last = knots[M - 1].pos;
    const double guess_scale = fdiv((double)(M - 1), rp_last - rp_first);
    // rows carry cos / sin of the reference heading when some stage needs the ego footprint
    const bool want_trig = OBST && ((do_collision && P.K > 0) || ((P.mode & FX_MODE_ROAD_BOUNDARY) && P.n_bound > 0));
    auto row_at = [&](int i) {
        return make_lon_row(i, S, M, dt, a_max, cl0, cl1, cl2, cl3, cl4, cl5, traj_len, tp, rp_first, rp_last, guess_scale,
                            want_trig, [&](int k) { return knots[k]; }, [&](int k) { return rpos[k]; },
                            (P.mode & FX_MODE_PROJ_PSEUDO_NORMAL) != 0);
    };
    auto lat_eval = [&](int i, double u_lowvel, double &d, double &dv, double &da) {
        LatU U;
        if (low_vel) U.from_parameter(u_lowvel);
        else U.from_table(tp + i * FX_TP);
        L.eval(U, d, dv, da);
    };
    // lateral value the extension holds: d[traj_len-1] (reactive_planner.py:344)
    double d_ext, dv_u, da_u;
    {
        const LonRow rl = row_at(traj_len - 1);
        lat_eval(traj_len - 1, rl.u1, d_ext, dv_u, da_u);
    }

    // ---- this lane's chunk of the horizon ----
    const int CH = G == 1 ? S : (S + G - 1) / G;
    const int i_begin = part * CH;
    const int i_end = min(S, i_begin + CH);
    // one step per lane (32 lanes per candidate, horizons up to 32 samples): the left lane hands heading, curvature and box of
    // step i - 1 over (walk_step, `neigh`) -- otherwise parts > 0 re-evaluate their carry-in step without emitting it
    const bool neigh = G == 32 && CH == 1;
    const int i_first = (G > 1 && part > 0 && !neigh) ? i_begin - 1 : i_begin;

    StepConst K;
    K.dt = dt; K.r_dt = 1.0 / dt; K.kappa_max = P.veh.kappa_max; K.a_max = a_max; K.v_switch = P.veh.v_switch;
    K.av_switch = a_max * P.veh.v_switch; K.v_des = P.v_des; K.wb = P.veh.wb_rear_axle; K.half_len = P.veh.length / 2;
    K.half_wid = P.veh.width / 2; K.S = S; K.half = S / 2; K.K = P.K; K.low_vel = low_vel; K.dbg = dbg;
    K.do_collision = do_collision; K.store_wt = (P.mode & FX_MODE_INT_STORE_WT) != 0;
    K.n_bound = (OBST && (P.mode & FX_MODE_ROAD_BOUNDARY)) ? P.n_bound : 0; K.bound_d_reach = P.bound_d_reach;
    K.ox = P.hot_origin[0]; K.oy = P.hot_origin[1]; K.gap_margin = P.hot_gap_margin;
    K.cull_r0 = (float)(1.41423 * sqrt(P.veh.length * P.veh.length + P.veh.width * P.veh.width) * 0.5);
    K.atan_k = nullptr;
    const BoundView Bv{as_global(P.bound_piece), as_global(P.bound_bin), as_global(P.bound_item)};
    const FX_GLOBAL double *__restrict__ obs_rec = as_global(P.obs_rec);
    const FX_GLOBAL unsigned long long *__restrict__ obs_pmask = as_global(P.obs_pmask);
    const FX_GLOBAL unsigned long long *__restrict__ obs_hmask = as_global(P.obs_hmask);

    StepCarry Cy;
    Cy.th_prev = P.x0_orientation; Cy.kap_prev = 0.0; Cy.bx_prev = Cy.by_prev = Cy.ux_prev = Cy.uy_prev = 0.0;
    // one step per lane (32 lanes per candidate): the lanes ARE the candidate's steps, so one ballot of "my step moves" answers
    // every lane's "last moving step in front of mine" (fx_eval_grid_kernel.h has the same; a lane-by-lane scan made the
    // workgroups of rows that end in the extension at the slowest sampled velocity the last to take their ticket)
    unsigned long long moving_lanes = 0ULL;
    if (neigh) {
        bool mv = false;
        if (!low_vel && i_begin < S) {
            const double *te = tp + (i_begin < traj_len ? i_begin : traj_len - 1) * FX_TP;
            double sv = cl1 + 2. * cl2 * te[0] + 3. * cl3 * te[1] + 4. * cl4 * te[2] + 5. * cl5 * te[3];
            if (fabs(sv) < FX_EPS) sv = 0.0;
            mv = sv > 0.001;
        }
        moving_lanes = __ballot(mv);
    }
    if (G > 1 && !low_vel && i_first > 0 && i_first < S) {
        // the carry-in step keeps the previous heading when it stands still: scan back to the last moving step (:447).  The scan
        // only needs each step's LON_MOVING bit -- the longitudinal velocity, make_lon_row's own expression -- and builds ONE row,
        // the one it stops at (a full row per scanned step made a braking ego's step 42 us where a cruising one took 23:
        // BASELINE config 1's 800-row matrix, v0 = 5.6 m/s)
        auto moving_at = [&](int i) {
            const double *te = tp + (i < traj_len ? i : traj_len - 1) * FX_TP;
            double sv = cl1 + 2. * cl2 * te[0] + 3. * cl3 * te[1] + 4. * cl4 * te[2] + 5. * cl5 * te[3];
            if (fabs(sv) < FX_EPS) sv = 0.0;
            return sv > 0.001;
        };
        if (!moving_at(i_first)) {
            int j = i_first - 1;
            if (neigh) {   // one step per lane: the last moving step in front of this one from the candidate's ballot (below)
                const unsigned int grp = (unsigned int)(moving_lanes >> ((tid & 63) & 32));
                const unsigned int below = grp & ((1u << part) - 1u);
                j = below ? 31 - __clz((int)below) : -1;
            } else {
                while (j >= 0 && !moving_at(j)) j--;
            }
            if (j >= 0) {
                const LonRow rj = row_at(j);
                double d_j, dv_j = 0.0, da_j;
                if (j < traj_len) lat_eval(j, rj.u1, d_j, dv_j, da_j);
                Cy.th_prev = heading_of_moving_step(rj, dv_j);
            }
        }
    }
    StepAcc A;
    A.neg = A.acc_viol = A.collided = false;
    A.step_reasons = 0; A.first_key = 0xffffffffu; A.fail_step = 0x7fffffff; A.bound_step = 0x7fffffff;
    A.sum_abs_d = A.sum_voff = A.pred = A.d_end = A.v_end = 0.0;
    StepOut O;
    Simpson sim_acc, sim_jerk, sim_orient, sim_path;
    double a_prev = 0.0, thcl_prev = 0.0, dto = 0.0;
    if (EXTRA) { sim_acc.init(); sim_jerk.init(); sim_orient.init(); sim_path.init(); }
    const int n_dto = EXTRA ? P.n_dto : 0;
    const FX_GLOBAL double *__restrict__ dto_pos = as_global(P.dto_pos);
    bool lane_on = false;   // lane_center_offset among the step's cost terms
    if (EXTRA)
        for (int n = 0; n < P.n_cost; n++) lane_on |= P.cost_id[n] == FX_COST_LANE_CENTER_OFFSET;
    double lane_off = 0.0;
    FX_GLOBAL double *__restrict__ planes = as_global(P.planes);
    const int64_t ps = (int64_t)S * ld;

#pragma unroll 1
    for (int i = i_first; i < i_end; i++) {
        const bool emit = i >= i_begin;  // false only for the carry-in step of parts > 0
        const LonRow r = row_at(i);
        if (LSTAGE && rec_staged)   // (workgroup-uniform: the step's obstacle records and masks come from LDS)
            walk_step<OBST, false>(K, r, L, tp, i, traj_len, d_ext, emit, bundle && active && emit, planes + (int64_t)i * ld + g, 0u, ps,
                                   Cy, A, O, (const double *)rec_lds, (const unsigned long long *)pm_lds,
                                   (const unsigned long long *)(pm_lds + S), Bv, nullptr, -1, neigh, part > 0);
        else
        walk_step<OBST, G == 1>(K, r, L, tp, i, traj_len, d_ext, emit, bundle && active && emit, planes + (int64_t)i * ld + g, 0u, ps,
                                Cy, A, O, obs_rec, obs_pmask, obs_hmask, Bv, nullptr, -1, neigh, part > 0);
        if (EXTRA) {
            sim_acc.push(O.a * O.a, S);                                 // partial_cost_functions.py:29-31
            sim_path.push(O.v, S);                                      // :194-195
            if (i > 0) {
                const double j = (O.a - a_prev) / dt;                   // :41-44
                const double w = (O.th_cl - thcl_prev) / dt;            // :146-149
                sim_jerk.push(j * j, S - 1);
                sim_orient.push(w * w, S - 1);
            }
            a_prev = O.a;
            thcl_prev = O.th_cl;
            for (int o = 0; o < n_dto; o++) {                           // :177-184
                const double ex = O.x - dto_pos[2 * o], ey = O.y - dto_pos[2 * o + 1];
                const double dist = sqrt(ex * ex + ey * ey);
                dto += 1.0 / (dist * dist);
            }
            if (lane_on) lane_off += lane_center_distance(P, O.x, O.y);     // :106-115, a running sum over the points
        }
    }

    // ---- combine parts, flags, cost, outputs, workgroup reductions ----
    WalkResult W;
    W.neg = A.neg; W.acc_viol = A.acc_viol; W.collided = A.collided;
    W.step_reasons = A.step_reasons; W.first_key = A.first_key; W.fail_step = A.fail_step; W.bound_step = A.bound_step;
    W.sum_abs_d = A.sum_abs_d; W.sum_voff = A.sum_voff; W.pred = A.pred; W.dto = dto; W.lane_off = lane_off; W.d_end = A.d_end; W.v_end = A.v_end;
    W.cl3 = cl3; W.cl4 = cl4; W.cl5 = cl5; W.ct3 = L.c3; W.ct4 = L.c4; W.ct5 = L.c5;
    if (EXTRA) { W.sim_acc = sim_acc; W.sim_jerk = sim_jerk; W.sim_orient = sim_orient; W.sim_path = sim_path; }
    finish_candidate<G, BUNDLE, OBST, EXTRA>(P, Pg, W, g, active, part, i_begin, i_end, bundle, do_collision, dbg, D, red_cost,
                                             red_idx, red_cnt, fuse);
}
