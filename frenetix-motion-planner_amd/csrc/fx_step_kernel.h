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
//   phase 2  the obstacle stage: fx_obstacle_kernel.h's single-wave (tile of 64 listed candidates, chunk of CH steps) items dealt to
//            ALL waves of the launch (fx_obstacle_body<CH, false, true>: the chunks of a tile meet through global memory; input tables
//            through the constant address space so that they stay scalar loads behind the barrier); CH is picked so that the items
//            fit the launch's waves in one round where possible;
//   barrier
//   phase 3  the selection (fx_select_body<true>): every workgroup reduces the tiles' partials to the winner, counts the colliding
//            candidates in front of it in ITS slice of the candidates and takes a ticket; the last one publishes (and gathers the
//            winner package).
//
// Results are those of the three launches bit for bit at three steps per item (the phases are the kernels' bodies and expression
// trees; tests/test_step_kernel.py).  Measured (tools/c3_step_kernel.py, tools/probe_step_kernel.py; profiles/r6/NOTES.md): walk
// done at 38 us, each barrier ~1 us, selection 5 - 9 us -- and the obstacle phase 40 - 45 us against the obstacle kernel's 32:
// config 3 takes 94 us this way, 86 us as three launches (the same with a first version of phase 2 that parked EVERY operand of
// the visit loops in LDS and walked several tiles per wave: 94 - 99 us).  Why: the launch runs with the walk's 165 registers -- three waves per SIMD,
// 3 072 waves for 5 030 three-step items --, so every wave runs TWO items' latency chains back to back (entry round trips, visits with
// their LDS waits, hand-off, closing), where the obstacle kernel's 74 registers give every item its own wave at six per SIMD and
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

// Dynamic LDS of one wave's obstacle item: fx_obstacle_kernel.h's wave-private operand table, (cu, cw) [CH][K][2] | hull circles [CH][K][4]
#define FX_STEP_ITEM_DOUBLES(CH, K) ((size_t)6 * (size_t)(CH) * (size_t)(K))

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
        // items = (tile, chunk), tile-major: the waves of a workgroup share a tile's list entries, flag words and rows
        const int n_items = n_tiles * NC;
        for (int item = (int)blockIdx.x * n_wave + wave; item < n_items; item += n_w) {
            const int grp = item / NC;
            fx_obstacle_body<CH, false, true>(P, my_lds, grp, item - grp * NC, n_live);
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

