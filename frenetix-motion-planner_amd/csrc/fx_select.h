// fx_select.h -- the end of a plan step whose collision stage ran outside a tail kernel: winner, collision count, publication.
#pragma once

#include "fx_tail.h"

// ---------------------------------------------------------------------------------------------------
// Selection kernel: one workgroup per agent.  Reduces the per-workgroup partials to the winner and
// counts the colliding candidates that the reference's cost-ordered walk would have visited before it
// (planner.py:336-357 `_collision_counter`).
// ---------------------------------------------------------------------------------------------------
// grid = (slices, n_agents): every workgroup reduces the (few hundred) partials to the winner on its own, counts
// the colliding candidates ordered before the winner in its slice of the candidates (loads of four iterations in
// flight), adds its count to the agent's device counter and takes a ticket; the workgroup that draws the last ticket
// publishes the result block.  One workgroup scanning 50 000 candidates took ~30 us; the slices take ~5.
// The number of slices grows with the candidate count (host: fx_launch_select): 32 for planner-sized and 50 000-candidate steps,
// 256 at a million candidates -- with a fixed 32 every workgroup scanned 31 000 cost / flag pairs there while 224 CUs idled.
#define FX_SELECT_SLICES_MIN 32
#define FX_SELECT_SLICES_MAX 512
// The body, for workgroup `slice` of `n_slices` of agent `agent` (256 lanes).  COH (fx_step_kernel.h: the selection as the last
// phase of a one-launch step): partials, flag words and costs were written by other workgroups of the launch that is still
// running (agent-scope stores, acknowledged before the grid barrier) -- agent-scope loads; `n_part` = how many partials exist.
template <bool COH>
__device__ __forceinline__ void fx_select_body(const DevProblem &P, const int agent, const int slice, const int n_slices, const int n_part,
                                               unsigned long long *host_result, unsigned long long seq, double *dev_winner,
                                               double *host_pkg, int pkg_stride, int pkg_plane_rows) {
    __shared__ double sc[4];
    __shared__ long long si[4];
    __shared__ unsigned int scnt;
    __shared__ unsigned long long s_ticket;
    auto LD = [](auto p) { return COH ? fxk::ld_agent(p) : *p; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double bc = INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    // Everything that does not depend on the winner is requested NOW, together with the partials -- the first eight (flag, cost)
    // pairs per thread of this workgroup's slice (the whole slice up to 2 048 candidates: planner-sized steps, config 3) and the
    // step's counters: the kernel is a chain of memory round trips, and these two used to be links of their own.
    const bool count_mode = (P.mode & FX_MODE_COLLISION) != 0;
    const int64_t per = (P.C + n_slices - 1) / n_slices;
    const int64_t g0 = min(P.C, (int64_t)slice * per), g1 = min(P.C, g0 + per);
    uint32_t f_pre[8];
    double c_pre[8];
    if (count_mode) {
        const FX_GLOBAL uint32_t *__restrict__ fl = as_global(P.flags);
        const FX_GLOBAL double *__restrict__ co = as_global(P.cost);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t gu = g0 + tid + u * 256;
            f_pre[u] = gu < g1 ? LD(fl + gu) : 0u;
            c_pre[u] = gu < g1 ? LD(co + gu) : 0.0;
        }
    }
    unsigned long long cnt_pre = 0ULL;
    if (tid < FX_CNT_BEST_IDX) cnt_pre = LD(as_global(P.counters) + tid);   // (accumulated by the evaluation kernel, which is complete)
    {
        // the partials were written by the evaluation kernel, which is complete: plain loads, four per thread in flight
        // (device-coherent atomic loads, as the in-kernel selection needs them, serialise at ~1 us each)
        const FX_GLOBAL double *__restrict__ pc = as_global(P.part_cost);
        const FX_GLOBAL int64_t *__restrict__ pi = as_global(P.part_idx);
        for (int b0 = tid; b0 < n_part; b0 += 4 * 256) {
            double c[4];
            long long ix[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int b = b0 + u * 256;
                c[u] = b < n_part ? LD(pc + b) : INFINITY;
                ix[u] = b < n_part ? (long long)LD(pi + b) : 0x7fffffffffffffffLL;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (c[u] < bc || (c[u] == bc && ix[u] < bi)) { bc = c[u]; bi = ix[u]; }
        }
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
    for (int w = 1; w < 4; w++)
        if (sc[w] < bc || (sc[w] == bc && si[w] < bi)) { bc = sc[w]; bi = si[w]; }
    const bool none = bi == 0x7fffffffffffffffLL;
    // colliding selectable candidates ordered before the winner (all of them when nothing is collision-free)
    if (count_mode) {
        const FX_GLOBAL uint32_t *__restrict__ fl = as_global(P.flags);
        const FX_GLOBAL double *__restrict__ co = as_global(P.cost);
        unsigned int cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {   // the pairs requested at entry
            const int64_t gu = g0 + tid + u * 256;
            if ((f_pre[u] & FX_FLAG_SELECTABLE) && (f_pre[u] & FX_FLAG_COLLISION) &&
                (none || c_pre[u] < bc || (c_pre[u] == bc && gu + P.g_base < bi))) cnt++;
        }
        for (int64_t g = g0 + tid + 8 * 256; g < g1; g += 4 * 256) {
            uint32_t f[4];
            double c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t gu = g + u * 256;
                f[u] = gu < g1 ? LD(fl + gu) : 0u;
                c[u] = gu < g1 ? LD(co + gu) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t gu = g + u * 256;
                if ((f[u] & FX_FLAG_SELECTABLE) && (f[u] & FX_FLAG_COLLISION) && (none || c[u] < bc || (c[u] == bc && gu + P.g_base < bi))) cnt++;
            }
        }
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane == 0 && cnt) atomicAdd(&scnt, cnt);
        __syncthreads();
    }
    // The workgroup that draws the last ticket of this agent publishes.  ONE device-scope atomic per workgroup carries both
    // the ticket (low 16 bits) and the slice's count (upper bits): no second atomic, no fence between them -- the chain of
    // device-coherent round trips is what this kernel's 8 us are made of.
    static_assert(FX_SELECT_SLICES_MAX < 65536, "the ticket lives in the low 16 bits");
    if (tid == 0) s_ticket = atomicAdd(&P.counters[FX_DCNT_TICKET], ((unsigned long long)scnt << 16) | 1ULL);
    __syncthreads();
    if ((s_ticket & 0xffffULL) != (unsigned long long)(n_slices - 1)) return;
    const unsigned long long collisions = (s_ticket >> 16) + scnt;
    // Publish the step's result straight into pinned host memory (the host polls the sequence word instead of
    // paying for a D2H copy and a stream synchronisation) and leave the device counters zeroed for the next step.  The
    // counters were accumulated by the evaluation kernel, which is complete: plain loads and stores.
    unsigned long long *out = host_result + (size_t)agent * (FX_CNT_COUNT + 1);
    // (host words: system-scope stores, drained per wave, the sequence word behind a barrier -- no L2 write-back fence: fx_tail.h)
    {   // the result block in ONE store instruction (a wave's system-scope stores issue one behind the other): lanes 0 .. 12 the
        // counters, 13 .. 15 winner index, cost bits, collisions
        unsigned long long w = cnt_pre;
        if (tid == FX_CNT_BEST_IDX) w = none ? ~0ULL : (unsigned long long)bi;
        if (tid == FX_CNT_BEST_COST) w = none ? 0ULL : (unsigned long long)__double_as_longlong(bc);
        if (tid == FX_CNT_COLLISIONS) w = collisions;
        if (tid < FX_CNT_COUNT) fxk::put_host(out + tid, w);
    }
    if (tid < FX_CNT_BEST_IDX) as_global(P.counters)[tid] = 0ULL;
    if (tid == 0 && dev_winner) {  // (cost, index bits) of the winner, device-resident for the multi-GPU exchange
        dev_winner[2 * agent] = none ? INFINITY : bc;
        reinterpret_cast<long long *>(dev_winner)[2 * agent + 1] = none ? -1 : bi;
    }
    if (tid == 0) {
        __hip_atomic_store(&P.counters[FX_DCNT_TICKET], 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the obstacle kernel's candidate list (deferred obstacle stage) starts the next step empty
        if (P.mode & FX_MODE_INT_DEFER_OBST) as_global(P.counters)[FX_DCNT_LIVE] = 0ULL;
    }
    // the winner package (fx_set_package): the evaluation kernel is complete, so the publishing workgroup gathers the chosen
    // trajectory right here -- no further launch; its sequence word goes out behind the result block's
    double *pkg = host_pkg ? host_pkg + (size_t)agent * pkg_stride : nullptr;
    if (pkg) fxk::fx_package_gather<COH>(P, none ? -1LL : bi, pkg, pkg_plane_rows, tid, 256);
    else if (tid < 64) fxk::drain_stores();
    __syncthreads();
    if (tid == 0) {
        fxk::st_host(out + FX_CNT_COUNT, seq);
        if (pkg) fxk::st_host(reinterpret_cast<unsigned long long *>(pkg + pkg_stride - 1), seq);
    }
}

__global__ __launch_bounds__(256) void fx_select_kernel(const DevProblem *__restrict__ probs, unsigned long long *host_result,
                                                        unsigned long long seq, double *dev_winner, double *host_pkg, int pkg_stride,
                                                        int pkg_plane_rows) {
    const DevProblem &P = probs[blockIdx.y];
    fx_select_body<false>(P, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, P.n_blocks, host_result, seq, dev_winner, host_pkg, pkg_stride,
                          pkg_plane_rows);
}
