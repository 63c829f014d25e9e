// fx_tail.h -- the end of a plan step inside the kernel that finishes it: winner, collision count, winner package, publication.
//
// A plan step ends with (a) the lexicographic (cost, index) arg-min over the selectable collision-free candidates -- the first
// collision-free entry of the reference's stable-sorted list (trajectories.py:524-561, planner.py:329-392) --, (b) the number of
// colliding candidates that list holds in front of it (planner.py:336-357 `_collision_counter`), (c) the arrays of the chosen
// trajectory (planner.py:394-447 reads them) and (d) the counters of reactive_planner.py:229-235.  Steps of 50 000 candidates do
// (a)/(b) in fx_select_kernel (32 - 512 workgroups scan the candidates) and (c) in its publishing workgroup; for PLANNER-SIZED
// steps -- the reference's own operating point: 630 / 800 candidates (planning.yaml:34-35), a few thousand per agent in a batch --
// a launch costs more than the work, so the workgroup of the evaluation kernel that draws the agent's LAST completion ticket does
// all four itself: one launch per step.
//
// Visibility between workgroups (they may sit on different XCDs, whose L2s are not coherent with each other): everything the
// tail reads was stored with agent-scope (write-through, `sc1`) stores and acknowledged (`s_waitcnt vmcnt(0)`) by its wave before
// that wave's workgroup took its ticket; the tail reads with agent-scope loads.  Nothing is fenced, no L2 is written back.
#pragma once

#include "fx_device.h"

#ifdef FX_TAIL_STAMPS
#define FX_TSTAMP(k) FX_STAMP(k)
#else
#define FX_TSTAMP(k) do { } while (0)
#endif

namespace fxk {

struct fx_d2c { double x, y; };

template <typename T>
__device__ __forceinline__ T ld_agent(const FX_GLOBAL T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void st_agent(FX_GLOBAL T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The pinned, mapped result / package blocks are published WITHOUT a release fence: a system-scope release is `buffer_wbl2 sc0 sc1`,
// a write-back pass over the whole XCD L2 (~1.7 us each), and the fence-per-wave + fence + release-store sequence of round 4
// paid three of them in a row.  Instead every word goes out with a system-scope (`sc0 sc1`, write-through) store, every wave waits
// until its own stores are acknowledged (`s_waitcnt vmcnt(0)`), the workgroup meets at a barrier, and the sequence word goes out
// behind them.  (A wave's system-scope stores issue one behind the other, ~0.5 us each -- tools/probe_timeline.py --: the
// package is written with as few store instructions per wave as the layout allows.)
// payload word of a host block.  NOT a plain store: those are acknowledged by the L2 -- with `s_waitcnt vmcnt(0)` and no write-back
// fence a result block reached the host AFTER its sequence word (test_work_decomposition..., test_adapter_replay failed on it)
template <typename T>
__device__ __forceinline__ void put_host(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// Sequence word of a host block.  Payload-before-sequence ordering here is NOT the memory model's (relaxed atomics carry none): it
// is the hardware's -- system-scope stores of gfx942 / gfx950 are write-through and `s_waitcnt vmcnt(0)` returns when they are
// performed, and every wave has waited for its own before the barrier in front of this store (measured:
// test_published_blocks_are_never_torn).  Any other target gets the portable form: a system-scope RELEASE.
template <typename T>
__device__ __forceinline__ void st_host(T *p, T v) {
#if defined(__gfx942__) || defined(__gfx950__)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
// two adjacent payload words with ONE 16-byte system-scope store (p 16-byte aligned): the number of store INSTRUCTIONS a wave issues
// is what the publication costs
__device__ __forceinline__ void put_host2(double *p, double a, double b) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t v = {a, b};
#if defined(__gfx942__) || defined(__gfx950__)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");   // (the gfx940+ spelling of a system-scope store)
#else
    put_host(p, a); put_host(p + 1, b);
#endif
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// a per-candidate output: write-through where a later workgroup of the SAME launch reads it (the tail), plain otherwise
template <typename T>
__device__ __forceinline__ void st_out(FX_GLOBAL T *p, T v, bool wt) {
    if (wt) st_agent(p, v);
    else *p = v;
}

// Winner package: everything the planner reads of the chosen trajectory -- planes [14][S], coefficients, raw partial costs,
// cost, horizon, flag word -- gathered from the SoA outputs straight into pinned host memory, so that a plan step ends with ONE
// wait instead of a second round of strided copies and a stream synchronisation (reactive_planner_cpp.py:355-357 reads the
// optimal trajectory's arrays, planner.py:394-447 packages them).  Layout per agent (doubles): planes | lon6 lat6 | raw[FX_NUM_COSTS]
// | cost | traj_len | flags | index | found | tau_lat, then the sequence word at stride - 1.
// The gather itself, for one agent, by the `nthreads` lanes of a workgroup (tid = 0 .. nthreads - 1); every wave leaves with its
// stores acknowledged (st_host / drain_stores above; the caller orders the sequence word behind a barrier).  COHERENT: the sources were written by
// other workgroups of the launch that is still running (agent-scope loads); false behind a kernel boundary.
template <bool COHERENT, typename PR>
__device__ __forceinline__ void fx_package_gather(const PR &P, long long gi, double *out, int plane_rows, int tid, int nthreads) {
    double *tail = out + plane_rows;
    const bool found = gi >= 0 && (P.mode & FX_MODE_WRITE_BUNDLE);
    if (found) {
        const int64_t l = gi - P.g_base, ld = P.ld;
        const int n_pl = FX_NUM_PLANES * P.S;
        const FX_GLOBAL double *pl = as_global(P.planes);
        auto rd = [&](const FX_GLOBAL double *p) { return COHERENT ? ld_agent(p) : *p; };
        // ONE round trip for a planner-sized workgroup: two PAIRS of adjacent plane values per lane (128 lanes cover the 217 pairs
        // of 14 x 31 values) and the lane's word of the package's tail are requested together and stored with two 16-byte stores
        // + one 8-byte store; only larger horizons loop on.  The tail's words come from five arrays: the lane picks an ADDRESS
        // (a load behind each divergent branch is a round trip per branch -- the six of the first version cost 3.8 us), the two
        // integer words are read by every lane.
        auto tail_addr = [&](int j) -> const FX_GLOBAL double * {
            if (j < 12) return as_global(P.coeffs) + (size_t)j * ld + l;
            if (j < 12 + FX_NUM_COSTS) return as_global(P.costmap) + (size_t)(j - 12) * ld + l;
            if (j == 17 + FX_NUM_COSTS) return as_global(P.coeffs) + (size_t)12 * ld + l;
            return as_global(P.cost) + l;
        };
        static_assert(FX_PKG_TAIL <= 64 && FX_NUM_PLANES % 2 == 0, "one tail word per lane of the first wave; plane values in pairs");
        const int nq = n_pl / 2;
        double va[2], vb[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = tid + u * nthreads;
            va[u] = q < nq ? rd(pl + (size_t)(2 * q) * ld + l) : 0.0;
            vb[u] = q < nq ? rd(pl + (size_t)(2 * q + 1) * ld + l) : 0.0;
        }
        // (the tail's words are written by the LAST wave, the result block by the first: a wave's system-scope stores issue one
        // behind the other, ~0.75 us each, so the publication's store instructions are spread over the waves)
        const bool has_cm = (P.mode & FX_MODE_WRITE_COSTMAP) != 0;
        const int tt = tid - (nthreads - 64);   // lane of the last wave
        const int jt = tt >= 0 && tt < FX_PKG_TAIL ? tt : 0;
        const bool t_lane = tt >= 0 && tt < FX_PKG_TAIL;
        const bool t_load = t_lane && (jt < 12 || (jt < 12 + FX_NUM_COSTS && jt - 12 < P.n_cost && has_cm) || jt == 12 + FX_NUM_COSTS ||
                                                  jt == 17 + FX_NUM_COSTS);
        double tw = t_load ? rd(tail_addr(jt)) : 0.0;
        const int32_t tl_w = COHERENT ? ld_agent(as_global(P.traj_len) + l) : as_global(P.traj_len)[l];
        const uint32_t fl_w = COHERENT ? ld_agent(as_global(P.flags) + l) : as_global(P.flags)[l];
        if (!t_lane) tw = 0.0;
        if (jt == 13 + FX_NUM_COSTS) tw = (double)tl_w;
        if (jt == 14 + FX_NUM_COSTS) tw = (double)fl_w;
        if (jt == 15 + FX_NUM_COSTS) tw = (double)gi;
        if (jt == 16 + FX_NUM_COSTS) tw = 1.0;   // found
        FX_TSTAMP(5);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = tid + u * nthreads;
            if (q < nq) put_host2(out + 2 * q, va[u], vb[u]);
        }
        FX_TSTAMP(6);
        if (t_lane) put_host(tail + jt, tw);
        FX_TSTAMP(7);
        for (int q0 = tid + 2 * nthreads; q0 < nq; q0 += 2 * nthreads) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int q = q0 + u * nthreads;
                va[u] = q < nq ? rd(pl + (size_t)(2 * q) * ld + l) : 0.0;
                vb[u] = q < nq ? rd(pl + (size_t)(2 * q + 1) * ld + l) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int q = q0 + u * nthreads;
                if (q < nq) put_host2(out + 2 * q, va[u], vb[u]);
            }
        }
    } else if (tid == 16 + FX_NUM_COSTS) {
        put_host(tail + 16 + FX_NUM_COSTS, 0.0);   // found = 0: nothing else of the block is read
    }
    drain_stores();
    FX_TSTAMP(8);
}

// The last workgroup of agent `agent` (ALL its lanes that are still alive: blockDim.x, a multiple of 64, at most 1 024): reduce
// the agent's partials, count the colliding candidates in front of the winner, gather the winner's package, publish.
// The caller guarantees: every workgroup of the agent has stored its partial (cost, index), its counters and -- tail & FX_TAIL_COUNT:
// cost[] / flags[]; tail & FX_TAIL_PACKAGE: everything fx_package_gather reads -- with agent-scope stores that were acknowledged
// before its ticket; this workgroup drew ticket n_blocks - 1 and all its waves are here.
template <typename PR>
__device__ __forceinline__ void fx_fused_tail(const PR &P, const DevProblem &Pg, const FuseArgs &fuse, int agent) {
    __shared__ double t_cost[16];
    __shared__ long long t_idx[16];
    __shared__ unsigned int t_cnt;
    const int tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    unsigned long long *out = fuse.host_result + (size_t)agent * (FX_CNT_COUNT + 1);
    // counters: read and zero in one agent-scope exchange (the next step starts from a clean block); issued first so that its
    // round trip overlaps the loads of the partials
    unsigned long long cnt = 0ULL;
    if (tid < FX_CNT_BEST_IDX) cnt = atomicExch(&P.counters[tid], 0ULL);
    // the step's flag words: the first FPRE pairs per lane are requested together with the partials (a planner-sized step's
    // whole flag array) -- the tail is a chain of round trips, this one rides with the first
    const bool count_mode = (fuse.tail() & FX_TAIL_COUNT) && (P.mode & FX_MODE_COLLISION);
    const int64_t C = P.C;
    const int64_t n2 = (C + 1) / 2;   // pairs of flag words (ld is a multiple of 64: the last pair exists)
    const FX_GLOBAL unsigned long long *fl2 = reinterpret_cast<const FX_GLOBAL unsigned long long *>(as_global(P.flags));
    const FX_GLOBAL double *co = as_global(P.cost);
    constexpr int FPRE = 24;   // flag pairs per lane requested up front: 48 x nthreads candidates (12 288 with 256 lanes) in ONE round trip
    unsigned long long f_pre[FPRE];
    fx_d2c c_pre[4];   // ... and the costs of the first four pairs (8 x nthreads candidates: config 1's 630 / 800 with 128 lanes)
    c_pre[0] = c_pre[1] = c_pre[2] = c_pre[3] = fx_d2c{0.0, 0.0};
    if (count_mode) {
#pragma unroll
        for (int u = 0; u < FPRE; u++) {
            const int64_t j = (int64_t)tid + (int64_t)u * nthreads;
            f_pre[u] = j < n2 ? ld_agent(fl2 + j) : 0ULL;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t j = (int64_t)tid + (int64_t)u * nthreads;
            if (j < n2) { c_pre[u].x = ld_agent(co + 2 * j); c_pre[u].y = ld_agent(co + 2 * j + 1); }   // (ld is a multiple of 64)
        }
    }
    double bc = INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    for (int b0 = tid; b0 < P.n_blocks; b0 += 4 * nthreads) {
        double c[4];
        long long ix[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = min(b0 + u * nthreads, P.n_blocks - 1);   // a repeated entry does not change the minimum
            c[u] = ld_agent(as_global(P.part_cost) + b);
            ix[u] = (long long)ld_agent(as_global(P.part_idx) + b);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c[u] < bc || (c[u] == bc && ix[u] < bi)) { bc = c[u]; bi = ix[u]; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oc = __shfl_xor(bc, off);
        const long long oi = __shfl_xor(bi, off);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (nw > 1) {
        if (lane == 0) { t_cost[wave] = bc; t_idx[wave] = bi; }
        if (tid == 0) t_cnt = 0;
        __syncthreads();
        bc = t_cost[0]; bi = t_idx[0];
        for (int w = 1; w < nw; w++)
            if (t_cost[w] < bc || (t_cost[w] == bc && t_idx[w] < bi)) { bc = t_cost[w]; bi = t_idx[w]; }
    }
    const bool none = bi == 0x7fffffffffffffffLL;
    FX_STAMP(13);
    // colliding selectable candidates ordered before the winner (all of them when nothing is collision-free): flag words first,
    // the cost only of what collides
    unsigned long long collisions = 0ULL;
    if (count_mode) {
        unsigned int mine = 0;
        // one batch = eight flag pairs per lane.  The costs of what collides are requested for the WHOLE batch before the first
        // comparison (a load inside the per-candidate branch is a round trip per candidate: 8 us for 11 000 candidates); lanes
        // with nothing to ask read word 0 -- no branch around a load.  The first four pairs' costs came with the flags.
        auto need = [&](unsigned long long two, int64_t j, int h) {
            const uint32_t f = (uint32_t)(two >> (32 * h));
            return 2 * j + h < C && (f & FX_FLAG_SELECTABLE) && (f & FX_FLAG_COLLISION);
        };
        auto before = [&](double c, int64_t g) { return c < bc || (c == bc && g + P.g_base < bi); };
        auto batch = [&](const unsigned long long *f2, int64_t j0, const fx_d2c *have) {
            bool nd[16];
            bool any = false;
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int h = 0; h < 2; h++) { nd[2 * u + h] = need(f2[u], j0 + (int64_t)u * nthreads, h); any |= nd[2 * u + h]; }
            if (!__any(any)) return;
            if (none) {
#pragma unroll
                for (int k = 0; k < 16; k++) mine += nd[k] ? 1u : 0u;
                return;
            }
            double c[16];
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int64_t g = 2 * (j0 + (int64_t)u * nthreads) + h;
                    if (have && u < 4) c[2 * u + h] = h ? have[u].y : have[u].x;
                    else c[2 * u + h] = ld_agent(co + (nd[2 * u + h] ? g : 0));
                }
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (nd[2 * u + h] && before(c[2 * u + h], 2 * (j0 + (int64_t)u * nthreads) + h)) mine++;
        };
        batch(f_pre, (int64_t)tid, c_pre);
#pragma unroll
        for (int q = 1; q < FPRE / 8; q++)
            if ((int64_t)q * 8 * nthreads < n2) batch(f_pre + 8 * q, (int64_t)tid + (int64_t)q * 8 * nthreads, nullptr);
        for (int64_t j0 = (int64_t)tid + FPRE * (int64_t)nthreads; j0 < n2; j0 += 8 * (int64_t)nthreads) {
            unsigned long long f2[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t j = j0 + (int64_t)u * nthreads;
                f2[u] = j < n2 ? ld_agent(fl2 + j) : 0ULL;
            }
            batch(f2, j0, nullptr);
        }
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
        if (nw > 1) {
            if (lane == 0 && mine) atomicAdd(&t_cnt, mine);
            __syncthreads();
            mine = t_cnt;
        }
        collisions = mine;
    }
    FX_STAMP(14);
    // publish the step's result straight into pinned host memory (the host polls the sequence word)
    {   // the result block in ONE store instruction: lanes 0 .. 12 the counters, 13 .. 15 winner index, cost bits, collisions
        static_assert(FX_CNT_BEST_IDX == 13 && FX_CNT_BEST_COST == 14 && FX_CNT_COLLISIONS == 15 && FX_CNT_COUNT == 16, "result block layout");
        unsigned long long w = cnt;
        if (tid == FX_CNT_BEST_IDX) w = none ? ~0ULL : (unsigned long long)bi;
        if (tid == FX_CNT_BEST_COST) w = none ? 0ULL : (unsigned long long)__double_as_longlong(bc);
        if (tid == FX_CNT_COLLISIONS) w = collisions;
        if (tid < FX_CNT_COUNT) put_host(out + tid, w);
    }
    if (tid == 0) {
        if (fuse.dev_winner) {
            fuse.dev_winner[2 * agent] = none ? INFINITY : bc;
            reinterpret_cast<long long *>(fuse.dev_winner)[2 * agent + 1] = none ? -1 : bi;
        }
        __hip_atomic_store(&P.counters[FX_DCNT_TICKET], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the obstacle kernel's candidate list (deferred obstacle stage) starts the next step empty
        if (P.mode & FX_MODE_INT_DEFER_OBST) __hip_atomic_store(&P.counters[FX_DCNT_LIVE], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double *pkg = (fuse.tail() & FX_TAIL_PACKAGE) ? Pg.pkg_out : nullptr;
    if (pkg) fx_package_gather<true>(P, none ? -1LL : bi, pkg, Pg.pkg_plane_rows, tid, nthreads);   // (ends with this wave's stores drained)
    else if (wave == 0) drain_stores();
    if (nw > 1) __syncthreads();
    FX_TSTAMP(9);
    // every wave's words have left the chip: the sequence words go out behind them (both in one store instruction)
    if (tid == 0 || (tid == 1 && pkg)) st_host(tid == 0 ? out + FX_CNT_COUNT : Pg.pkg_seq, fuse.seq);
    FX_STAMP(15);
}

}  // namespace fxk
