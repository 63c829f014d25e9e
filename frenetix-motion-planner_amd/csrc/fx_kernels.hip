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
#include <hip/hip_ext.h>

#include <atomic>

#include "fx_eval_kernel.h"
#include "fx_eval_grid_kernel.h"
#include "fx_obstacle_kernel.h"
#include "fx_step_kernel.h"

using fxk::wave_count;

// the vector unit's row_bcast data-parallel controls (fx_walk.h wave_max_u32, wave_min_u64 below) exist on the GFX9 family only
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "libfxplan's kernels are written for gfx950 (row_bcast DPP controls, wave64): build with --offload-arch=gfx950"
#endif

// slot of the calling thread's current device in the per-device tables of the launchers
#define FX_MAX_DEVICES 64
static inline int fx_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d % FX_MAX_DEVICES;
}

#ifdef FX_CULL_STATS
__device__ unsigned long long fx_cull_stats[16];
__device__ double fx_probe_bound = 1e300;
extern "C" int fx_probe_bound_set(double v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(fx_probe_bound), &v, sizeof(v), 0, hipMemcpyHostToDevice); }
extern "C" int fx_cull_stats_read(unsigned long long *out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fx_cull_stats), sizeof(fx_cull_stats), 0, hipMemcpyDeviceToHost);
    if (reset) {
        unsigned long long z[16] = {0};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(fx_cull_stats), z, sizeof(z), 0, hipMemcpyHostToDevice);
    }
    return rc;
}
#endif
#ifdef FX_PROBE
__device__ unsigned long long fx_probe_stamps[FX_PROBE_WAVES * FX_PROBE_SLOTS];
extern "C" int fx_probe_read(unsigned long long *out, size_t n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fx_probe_stamps), n_words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
__device__ unsigned long long fx_probe_stamps_obs[FX_PROBE_WAVES * FX_PROBE_SLOTS];
extern "C" int fx_probe_read_obs(unsigned long long *out, size_t n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fx_probe_stamps_obs), n_words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// (fx_package_gather: fx_tail.h)
__global__ __launch_bounds__(256) void fx_package_kernel(const DevProblem *__restrict__ probs, const double *__restrict__ winner,
                                                         double *host_pkg, int stride, int plane_rows, unsigned long long seq) {
    double *out = host_pkg + (size_t)blockIdx.x * stride;
    fxk::fx_package_gather<false>(probs[blockIdx.x], reinterpret_cast<const long long *>(winner)[2 * blockIdx.x + 1], out, plane_rows,
                                  (int)threadIdx.x, 256);
    __syncthreads();   // (every wave's package words are acknowledged: fx_tail.h, st_host / drain_stores)
    if (threadIdx.x == 0) fxk::st_host(reinterpret_cast<unsigned long long *>(out + stride - 1), seq);
}

extern "C" hipError_t fx_launch_package(const DevProblem *d_probs, int n_agents, const double *winner, double *host_pkg, int stride,
                                        int plane_rows, unsigned long long seq, hipStream_t stream) {
    hipLaunchKernelGGL(fx_package_kernel, dim3(n_agents), dim3(256), 0, stream, d_probs, winner, host_pkg, stride, plane_rows, seq);
    return hipGetLastError();
}

#include "fx_select.h"   // fx_select_kernel

// ---------------------------------------------------------------------------------------------------
// Top-k: the k best selectable collision-free candidates in (cost, index) order (k <= 64).
// Two small launches: FX_TOPK_SLICES workgroups per agent each extract the k best of a contiguous slice
// (k rounds of arg-min with a strict lower bound, over 1/64 of the candidates), then one workgroup per agent
// merges the 64 x k survivors.  Only the multi-GPU exchange and the host-side road-boundary walk need it.
// ---------------------------------------------------------------------------------------------------
#define FX_TOPK_SLICES 64

__device__ __forceinline__ void block_argmin(double &bc, long long &bi, double *sc, long long *si) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double oc = __shfl_xor(bc, off);
        const long long oi = __shfl_xor(bi, off);
        if (oc < bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sc[wave] = bc; si[wave] = bi; }
    __syncthreads();
    bc = sc[0]; bi = si[0];
    for (int w = 1; w < nw; w++)
        if (sc[w] < bc || (sc[w] == bc && si[w] < bi)) { bc = sc[w]; bi = si[w]; }
}

__global__ __launch_bounds__(256) void fx_topk_slice_kernel(const DevProblem *__restrict__ probs, int k, double *scr_cost,
                                                            long long *scr_idx) {
    __shared__ double sc[4];
    __shared__ long long si[4];
    const DevProblem &P = probs[blockIdx.y];
    const int tid = threadIdx.x;
    const int64_t per = (P.C + FX_TOPK_SLICES - 1) / FX_TOPK_SLICES;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(P.C, lo + per);
    const FX_GLOBAL uint32_t *__restrict__ flags = as_global(P.flags);
    const FX_GLOBAL double *__restrict__ cost = as_global(P.cost);
    const long long NONE = 0x7fffffffffffffffLL;
    double lb_c = -INFINITY;
    long long lb_i = -1;
    const size_t out = ((size_t)blockIdx.y * FX_TOPK_SLICES + blockIdx.x) * k;
    for (int r = 0; r < k; r++) {
        double bc = INFINITY;
        long long bi = NONE;
        for (int64_t g = lo + tid; g < hi; g += 256) {
            const uint32_t f = flags[g];
            if ((f & FX_FLAG_SELECTABLE) && !(f & (FX_FLAG_COLLISION | FX_FLAG_BOUNDARY))) {
                const double c = cost[g];
                const long long gg = (long long)(g + P.g_base);
                const bool after = c > lb_c || (c == lb_c && gg > lb_i);
                if (after && (c < bc || (c == bc && gg < bi))) { bc = c; bi = gg; }
            }
        }
        block_argmin(bc, bi, sc, si);
        const bool none = bi == NONE;
        if (tid == 0) { scr_cost[out + r] = none ? INFINITY : bc; scr_idx[out + r] = none ? -1 : bi; }
        if (none) { lb_c = INFINITY; lb_i = NONE; } else { lb_c = bc; lb_i = bi; }
    }
}

__global__ __launch_bounds__(256) void fx_topk_merge_kernel(int k, const double *__restrict__ scr_cost,
                                                            const long long *__restrict__ scr_idx, double *out_cost,
                                                            long long *out_idx) {
    __shared__ double sc[4];
    __shared__ long long si[4];
    const int tid = threadIdx.x;
    const int n = FX_TOPK_SLICES * k;
    const size_t base = (size_t)blockIdx.x * n;
    const long long NONE = 0x7fffffffffffffffLL;
    double lb_c = -INFINITY;
    long long lb_i = -1;
    for (int r = 0; r < k; r++) {
        double bc = INFINITY;
        long long bi = NONE;
        for (int e = tid; e < n; e += 256) {
            const long long gg = scr_idx[base + e];
            if (gg >= 0) {
                const double c = scr_cost[base + e];
                const bool after = c > lb_c || (c == lb_c && gg > lb_i);
                if (after && (c < bc || (c == bc && gg < bi))) { bc = c; bi = gg; }
            }
        }
        block_argmin(bc, bi, sc, si);
        const bool none = bi == NONE;
        if (tid == 0) { out_cost[(size_t)blockIdx.x * k + r] = none ? INFINITY : bc; out_idx[(size_t)blockIdx.x * k + r] = none ? -1 : bi; }
        if (none) { lb_c = INFINITY; lb_i = NONE; } else { lb_c = bc; lb_i = bi; }
    }
}

// One-wave variants (slices / survivor sets of at most 64 x FX_TOPK_R entries: grids up to 131 072 candidates per agent, k <= 32):
// ONE pass over memory into registers, then k rounds of a wave-level arg-min -- no workgroup barrier anywhere; every lane keeps
// the minimum of its own entries and only the owner of a round's winner retires it and looks again.  The k block-wide
// reductions of the general kernels above (two barriers each, eight workgroups per CU contending) cost 135 + 112 us per step for
// config 5's 32 agents x k = 32; the (cost, index) order is the same.
#define FX_TOPK_R 32   // entries per lane
// order-preserving key of a (non-NaN, no negative zero) double
__device__ __forceinline__ unsigned long long f64_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}

// min over the wave of a 64-bit key, wave-uniform result: row shifts and row broadcasts of the vector unit's data-parallel
// primitives on the two halves (no LDS crossbar: a __shfl_xor chain on 64-bit values costs ~3 000 cycles per reduction)
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
#define FX_DPP_MIN64(ctrl, rows)                                                                              \
    {                                                                                                          \
        const unsigned oh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, ctrl, rows, 0xf, false);   \
        const unsigned ol = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, ctrl, rows, 0xf, false);   \
        const bool lt = oh < hi || (oh == hi && ol < lo);                                                      \
        hi = lt ? oh : hi; lo = lt ? ol : lo;                                                                  \
    }
    FX_DPP_MIN64(0x111, 0xf);  // row_shr:1
    FX_DPP_MIN64(0x112, 0xf);  // row_shr:2
    FX_DPP_MIN64(0x114, 0xf);  // row_shr:4
    FX_DPP_MIN64(0x118, 0xf);  // row_shr:8   -> lane 15 of every row holds the row's minimum
    FX_DPP_MIN64(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    FX_DPP_MIN64(0x143, 0xc);  // row_bcast:31 into rows 2 and 3
#undef FX_DPP_MIN64
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

__device__ __forceinline__ long long readlane_i64(long long v, int lane) {
    return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((unsigned long long)v >> 32), lane) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane));
}

// One round's winner over the wave: minimum cost over the lanes' own minima, the lowest `id` among the lanes that hold it.
// Returns the owner lane (wave-uniform), -1 when no lane has anything left.
__device__ __forceinline__ int wave_round_owner(const double mc, const bool have, const long long id) {
    const unsigned long long key = have ? f64_key(mc) : ~0ULL;
    const unsigned long long kmin = wave_min_u64(key);
    unsigned long long tie = __builtin_amdgcn_ballot_w64(have && key == kmin);
    if (tie == 0ULL) return -1;
    int owner = __builtin_ctzll(tie);
    tie &= tie - 1;
    if (tie) {   // equal costs on several lanes (rare): the lowest index
        long long bi = readlane_i64(id, owner);
        while (tie) {
            const int l = __builtin_ctzll(tie);
            tie &= tie - 1;
            const long long oi = readlane_i64(id, l);
            if (oi < bi) { bi = oi; owner = l; }
        }
    }
    return owner;
}

__global__ __launch_bounds__(64) void fx_topk_slice_wave_kernel(const DevProblem *__restrict__ probs, int k, double *scr_cost,
                                                                long long *scr_idx) {
    const DevProblem &P = probs[blockIdx.y];
    const int lane = threadIdx.x;
    const int64_t per = (P.C + FX_TOPK_SLICES - 1) / FX_TOPK_SLICES;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(P.C, lo + per);
    const FX_GLOBAL uint32_t *__restrict__ flags = as_global(P.flags);
    const FX_GLOBAL double *__restrict__ cost = as_global(P.cost);
    // slot u of lane l holds candidate lo + l + 64 u: its cost, or NaN when it is not eligible (a NaN cost is never selected:
    // the comparisons of the general kernel drop it the same way)
    double c[FX_TOPK_R];
#pragma unroll
    for (int u = 0; u < FX_TOPK_R; u++) {
        const int64_t g = lo + lane + (int64_t)u * 64;
        c[u] = __builtin_nan("");
        if (g < hi) {
            const uint32_t f = flags[g];
            const double cg = cost[g];
            if ((f & FX_FLAG_SELECTABLE) && !(f & (FX_FLAG_COLLISION | FX_FLAG_BOUNDARY))) c[u] = cg + 0.0;   // (+ 0.0: no negative zero in the keys)
        }
    }
    const size_t out = ((size_t)blockIdx.y * FX_TOPK_SLICES + blockIdx.x) * k;
    // the lane's minimum is kept per group of eight slots: retiring an entry re-scans its group only (the rounds are bound by
    // the instructions of that re-scan: 2 048 waves x k rounds)
    double gm[FX_TOPK_R / 8];
    int gu[FX_TOPK_R / 8];
    auto scan_group = [&](auto Q) {
        constexpr int q = decltype(Q)::value;
        double m = INFINITY;
        int mu_ = -1;
#pragma unroll
        for (int u = 8 * q; u < 8 * q + 8; u++)
            if (c[u] < m || (mu_ < 0 && c[u] == c[u])) { m = c[u]; mu_ = u; }   // (an infinite cost is still a candidate)
        gm[q] = m; gu[q] = mu_;
    };
    auto retire_in_group = [&](auto Q, int slot) {
        constexpr int q = decltype(Q)::value;
#pragma unroll
        for (int u = 8 * q; u < 8 * q + 8; u++)
            if (u == slot) c[u] = __builtin_nan("");
        scan_group(Q);
    };
    static_assert(FX_TOPK_R == 32, "four groups of eight slots below");
    scan_group(std::integral_constant<int, 0>{}); scan_group(std::integral_constant<int, 1>{});
    scan_group(std::integral_constant<int, 2>{}); scan_group(std::integral_constant<int, 3>{});
    for (int r = 0; r < k; r++) {
        // first group that holds the smallest cost: slots grow with the group, so this is the lane's (cost, index) minimum
        double mc = gm[0];
        int mu = gu[0];
#pragma unroll
        for (int q = 1; q < FX_TOPK_R / 8; q++)
            if (gu[q] >= 0 && (mu < 0 || gm[q] < mc)) { mc = gm[q]; mu = gu[q]; }
        const long long id = (long long)(lo + lane + (int64_t)mu * 64 + P.g_base);
        const int owner = wave_round_owner(mc, mu >= 0, id);
        if (owner < 0) {
            if (lane == 0) { scr_cost[out + r] = INFINITY; scr_idx[out + r] = -1; }
            continue;
        }
        if (lane == owner) {   // publish, retire the entry, look again in its group (only this lane does anything here)
            scr_cost[out + r] = mc; scr_idx[out + r] = id;
            switch (mu >> 3) {
            case 0: retire_in_group(std::integral_constant<int, 0>{}, mu); break;
            case 1: retire_in_group(std::integral_constant<int, 1>{}, mu); break;
            case 2: retire_in_group(std::integral_constant<int, 2>{}, mu); break;
            default: retire_in_group(std::integral_constant<int, 3>{}, mu); break;
            }
        }
    }
}

// Merge of the 64 slices' sorted lists: lane l walks slice l's list through a head pointer (the lists sit in LDS), one round per
// output -- no rescans at all.
__global__ __launch_bounds__(64) void fx_topk_merge_wave_kernel(int k, const double *__restrict__ scr_cost, const long long *__restrict__ scr_idx,
                                                                double *out_cost, long long *out_idx) {
    __shared__ double l_cost[FX_TOPK_SLICES * FX_TOPK_R];
    __shared__ long long l_idx[FX_TOPK_SLICES * FX_TOPK_R];
    const int lane = threadIdx.x;
    const int n = FX_TOPK_SLICES * k;   // host: k <= FX_TOPK_R
    const size_t base = (size_t)blockIdx.x * n;
    for (int e = lane; e < n; e += 64) { l_cost[e] = scr_cost[base + e] + 0.0; l_idx[e] = scr_idx[base + e]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int head = 0;
    double mc = l_cost[lane * k];
    long long mi = l_idx[lane * k];   // -1: the slice's list has ended
    for (int r = 0; r < k; r++) {
        const int owner = wave_round_owner(mc, mi >= 0, mi);
        if (owner < 0) {
            if (lane == 0) { out_cost[(size_t)blockIdx.x * k + r] = INFINITY; out_idx[(size_t)blockIdx.x * k + r] = -1; }
            continue;
        }
        if (lane == owner) {
            out_cost[(size_t)blockIdx.x * k + r] = mc; out_idx[(size_t)blockIdx.x * k + r] = mi;
            head++;
            mi = head < k ? l_idx[lane * k + head] : -1;
            mc = head < k ? l_cost[lane * k + head] : INFINITY;
        }
    }
}

// Copy a small device buffer (the all-gathered survivors) into pinned host memory and publish a sequence word:
// the host polls instead of paying for a D2H copy + stream synchronisation.
__global__ __launch_bounds__(256) void fx_publish_kernel(const double *__restrict__ src, int n, double *host_dst,
                                                         unsigned long long *host_seq, unsigned long long seq) {
    for (int i = threadIdx.x; i < n; i += 256) fxk::put_host(host_dst + i, src[i]);
    fxk::drain_stores();  // per wave: its stores are acknowledged before the barrier lets thread 0 send the sequence word
    __syncthreads();
    if (threadIdx.x == 0) fxk::st_host(host_seq, seq);
}

extern "C" hipError_t fx_launch_publish(const double *src, int n, double *host_dst, unsigned long long *host_seq,
                                        unsigned long long seq, hipStream_t stream) {
    hipLaunchKernelGGL(fx_publish_kernel, dim3(1), dim3(256), 0, stream, src, n, host_dst, host_seq, seq);
    return hipGetLastError();
}

// Staging copy of a plan step's rewritten inputs: the kernel reads the pinned (mapped) staging block over the bus and writes
// the device copy, 16 B per lane.  For the 1 - 150 KB a state update touches this lands in a few microseconds behind the
// launch, where a DMA-engine copy of the same bytes costs its submission latency first (measured: tools/upload_step.py).
__global__ __launch_bounds__(256) void fx_stage_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

extern "C" hipError_t fx_launch_stage(const void *src_mapped, void *dst, size_t bytes, hipStream_t stream) {
    const int n16 = (int)(bytes / 16);
    const int blocks = std::min(256, std::max(1, (n16 + 255) / 256));
    hipLaunchKernelGGL(fx_stage_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const uint4 *>(src_mapped),
                       reinterpret_cast<uint4 *>(dst), n16);
    return hipGetLastError();
}

// element-wise check of the fx_math kernels (tests/test_hip_math.py)
// probe of the host-write path (fx_api.hip, probe_host_writes): every workgroup -- they are spread over all XCDs -- copies the same
// 64 bytes of the arena into its own slot, so that a stale copy of the line in ANY XCD's L2 shows
__global__ void fx_probe_read_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst) {
    if (threadIdx.x < 4) dst[blockIdx.x * 4 + threadIdx.x] = src[threadIdx.x];
}
extern "C" hipError_t fx_launch_probe_read(const void *src, void *dst, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(fx_probe_read_kernel, dim3(blocks), dim3(64), 0, stream, reinterpret_cast<const uint4 *>(src), reinterpret_cast<uint4 *>(dst));
    return hipGetLastError();
}

__global__ void fx_math_test_kernel(int n, const double *__restrict__ x, double *__restrict__ at, double *__restrict__ sn,
                                    double *__restrict__ cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    at[i] = fxm::atan(x[i]);
    fxm::sincos(x[i], &sn[i], &cs[i]);
}

extern "C" hipError_t fx_launch_math_test(int n, const double *x, double *at, double *sn, double *cs, hipStream_t stream) {
    hipLaunchKernelGGL(fx_math_test_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, x, at, sn, cs);
    return hipGetLastError();
}

// ---- launchers (called from fx_api.hip) ----
// Launch the evaluation kernel specialised for (G lanes per candidate, bundle, obstacles, extra costs, occupancy target).
extern "C" hipError_t fx_launch_eval(const DevProblem *d_probs, int n_agents, int max_blocks, size_t lds_bytes, int G,
                                     bool bundle, bool obst, bool extra, int wpe, hipEvent_t ev_start, hipEvent_t ev_stop,
                                     FuseArgs fuse, hipStream_t stream) {
    dim3 grid(max_blocks, n_agents), block(FX_BLOCK);
#define FX_LAUNCH(Gv, B, O, E, W)                                                                                \
    do {                                                                                                        \
        /* largest dynamic LDS size this specialisation has been enabled for, PER DEVICE (the attribute is per device; the call */ \
        /* costs microseconds: once per size, not per launch); relaxed atomics: two threads at worst both set the attribute */  \
        static std::atomic<size_t> lds_set_[FX_MAX_DEVICES];                                                    \
        std::atomic<size_t> &hw_ = lds_set_[fx_device_slot()];                                                  \
        if (lds_bytes > 48 * 1024 && lds_bytes > hw_.load(std::memory_order_relaxed)) {                         \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&fx_eval_kernel<Gv, B, O, E, W>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);    \
            if (e_ != hipSuccess) return e_;                                                                    \
            hw_.store(lds_bytes, std::memory_order_relaxed);                                                    \
        }                                                                                                       \
        hipExtLaunchKernelGGL((fx_eval_kernel<Gv, B, O, E, W>), grid, block, lds_bytes, stream, ev_start, ev_stop, 0, d_probs, fuse); \
        return hipGetLastError();                                                                               \
    } while (0)
#define FX_BO(Gv, W)                                                          \
    do {                                                                      \
        if (bundle && obst) FX_LAUNCH(Gv, true, true, false, W);              \
        if (bundle) FX_LAUNCH(Gv, true, false, false, W);                     \
        if (obst) FX_LAUNCH(Gv, false, true, false, W);                       \
        FX_LAUNCH(Gv, false, false, false, W);                                \
    } while (0)
    // four waves per SIMD (128 VGPRs) only exists for the plain select-only walk: with the bundle stores or the obstacle
    // stage that budget spills hundreds of bytes per lane to scratch -- slower than three waves, and kernels with that much
    // private memory faulted on this platform when a process used a second stream
#define FX_W(Gv)                                                          \
    do {                                                                  \
        if (wpe >= 4 && !bundle && !obst) FX_LAUNCH(Gv, false, false, false, 4); \
        if (wpe >= 3) FX_BO(Gv, 3);                                       \
        FX_BO(Gv, 2);                                                     \
    } while (0)
    if (extra) {  // windowed costs: one lane per candidate
        if (bundle && obst) FX_LAUNCH(1, true, true, true, 2);
        if (bundle) FX_LAUNCH(1, true, false, true, 2);
        if (obst) FX_LAUNCH(1, false, true, true, 2);
        FX_LAUNCH(1, false, false, true, 2);
    }
    switch (G) {
    case 32: FX_BO(32, 1);   // planner-sized sampling matrices: one or two steps per lane; one wave per SIMD is all they fill, so the
                             // allocator may use the whole register file (at two waves per SIMD: 12 registers spilled to scratch)
    case 16: FX_BO(16, 2);
    case 8: FX_W(8);
    case 4: FX_W(4);
    case 2: FX_W(2);
    default: FX_W(1);
    }
#undef FX_W
#undef FX_BO
#undef FX_LAUNCH
}

// Grid (t x v x d) specialisation with the shared longitudinal table; lds_bytes includes the rows.
extern "C" hipError_t fx_launch_eval_grid(const DevProblem *d_probs, int n_agents, int max_blocks, int block_size,
                                          size_t lds_bytes, int G, bool bundle, bool obst, int wpe, bool wsplit,
                                          hipEvent_t ev_start, hipEvent_t ev_stop, FuseArgs fuse, hipStream_t stream) {
    dim3 grid(max_blocks, n_agents), block(block_size);
#define FX_LAUNCH(Gv, B, O, W, WS)                                                                                 \
    do {                                                                                                          \
        static std::atomic<size_t> lds_set_[FX_MAX_DEVICES];   /* per device, see fx_launch_eval */                 \
        std::atomic<size_t> &hw_ = lds_set_[fx_device_slot()];                                                    \
        if (lds_bytes > 48 * 1024 && lds_bytes > hw_.load(std::memory_order_relaxed)) {                           \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&fx_eval_grid_kernel<Gv, B, O, W, WS>), \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);      \
            if (e_ != hipSuccess) return e_;                                                                      \
            hw_.store(lds_bytes, std::memory_order_relaxed);                                                      \
        }                                                                                                         \
        hipExtLaunchKernelGGL((fx_eval_grid_kernel<Gv, B, O, W, WS>), grid, block, lds_bytes, stream, ev_start, ev_stop, 0, d_probs, fuse); \
        return hipGetLastError();                                                                                 \
    } while (0)
#define FX_BO(Gv, W, WS)                                           \
    do {                                                           \
        if (bundle && obst) FX_LAUNCH(Gv, true, true, W, WS);      \
        if (bundle) FX_LAUNCH(Gv, true, false, W, WS);             \
        if (obst) FX_LAUNCH(Gv, false, true, W, WS);               \
        FX_LAUNCH(Gv, false, false, W, WS);                        \
    } while (0)
#define FX_W(Gv, WS)                                                          \
    do {                                                                      \
        if (wpe >= 4 && !bundle && !obst) FX_LAUNCH(Gv, false, false, 4, WS); \
        if (wpe >= 3) FX_BO(Gv, 3, WS);                                       \
        FX_BO(Gv, 2, WS);                                                     \
    } while (0)
    if (wsplit && G > 1) {
        switch (G) {
        case 4: FX_W(4, true);
        default: FX_W(2, true);
        }
    }
    switch (G) {
    case 32: FX_BO(32, 2, false);   // planner-sized grids: one or two steps per lane, one occupancy target
    case 16: FX_BO(16, 2, false);
    case 8: FX_W(8, false);
    case 4: FX_W(4, false);
    case 2: FX_W(2, false);
    default: FX_W(1, false);
    }
#undef FX_W
#undef FX_BO
#undef FX_LAUNCH
}

// The obstacle stage as its own kernel (fx_obstacle_kernel.h): grid = (max tiles x chunks, n_agents), one wave per item,
// dynamic LDS = CH * K * 48 B.
extern "C" hipError_t fx_launch_obstacle(const DevProblem *d_probs, int n_agents, int max_items, size_t lds_bytes, int CH,
                                         hipEvent_t ev_start, hipEvent_t ev_stop, hipStream_t stream, int wg_waves, int max_tiles) {
    // wg_waves > 0: one workgroup of wg_waves waves per tile (the chunks meet in LDS), grid = (max_tiles, n_agents)
#define FX_LAUNCH(CHv)                                                                                                                \
    do {                                                                                                                            \
        if (wg_waves > 0) {                                                                                                         \
            static std::atomic<size_t> lds_set_[FX_MAX_DEVICES];                                                                    \
            std::atomic<size_t> &hw_ = lds_set_[fx_device_slot()];                                                                  \
            if (lds_bytes > 48 * 1024 && lds_bytes > hw_.load(std::memory_order_relaxed)) {                                         \
                hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&fxk::fx_obstacle_kernel<CHv, 4, true>),          \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                    \
                if (e_ != hipSuccess) return e_;                                                                                    \
                hw_.store(lds_bytes, std::memory_order_relaxed);                                                                    \
            }                                                                                                                       \
            hipExtLaunchKernelGGL((fxk::fx_obstacle_kernel<CHv, 4, true>), dim3(max_tiles, n_agents), dim3(64 * wg_waves), lds_bytes, stream, \
                                  ev_start, ev_stop, 0, d_probs);                                                                   \
        } else {                                                                                                                    \
            hipExtLaunchKernelGGL((fxk::fx_obstacle_kernel<CHv, 4, false>), dim3(max_items, n_agents), dim3(64), lds_bytes, stream, ev_start, \
                                  ev_stop, 0, d_probs);                                                                             \
        }                                                                                                                           \
        return hipGetLastError();                                                                                                   \
    } while (0)
    if (CH == 2) FX_LAUNCH(2);
    if (CH == 3) FX_LAUNCH(3);
    if (CH == 5) FX_LAUNCH(5);
#undef FX_LAUNCH
    return hipErrorInvalidValue;
}

// The whole step in one launch (fx_step_kernel.h): grid = (blocks, n_agents), 256 lanes.  fx_step_kernel_capacity: how many of
// its workgroups the device holds at once with `lds_bytes` of dynamic LDS (the grid barrier needs them all resident).
#define FX_STEP_CASES(X) X(3) X(5) X(8)
extern "C" hipError_t fx_step_kernel_capacity(int CH, size_t lds_bytes, int *blocks_out) {
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
#define FX_CAP(CHv)                                                                                                              \
    if (CH == CHv) {                                                                                                             \
        if (lds_bytes > 48 * 1024) {                                                                                             \
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fx_step_kernel<CHv>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds_bytes);                                                                             \
            if (e != hipSuccess) return e;                                                                                       \
        }                                                                                                                        \
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fx_step_kernel<CHv>, FX_BLOCK, lds_bytes);                     \
        if (e != hipSuccess) return e;                                                                                           \
        *blocks_out = per_cu * cus;                                                                                              \
        return hipSuccess;                                                                                                       \
    }
    FX_STEP_CASES(FX_CAP)
#undef FX_CAP
    return hipErrorInvalidValue;
}
extern "C" hipError_t fx_launch_step(const DevProblem *d_probs, int n_agents, int blocks, size_t lds_bytes, int CH, hipEvent_t ev_start,
                                     hipEvent_t ev_stop, FuseArgs fuse, StepArgs sa, hipStream_t stream) {
#define FX_LAUNCH(CHv)                                                                                                           \
    if (CH == CHv) {                                                                                                             \
        hipExtLaunchKernelGGL((fx_step_kernel<CHv>), dim3(blocks, n_agents), dim3(FX_BLOCK), lds_bytes, stream, ev_start, ev_stop, 0, \
                              d_probs, fuse, sa);                                                                                \
        return hipGetLastError();                                                                                                \
    }
    FX_STEP_CASES(FX_LAUNCH)
#undef FX_LAUNCH
    return hipErrorInvalidValue;
}

extern "C" hipError_t fx_launch_select(const DevProblem *d_probs, int n_agents, int64_t max_candidates, unsigned long long *host_result,
                                       unsigned long long seq, double *dev_winner, double *host_pkg, int pkg_stride, int pkg_plane_rows,
                                       hipStream_t stream) {
    // one slice per ~4 096 candidates of the largest agent, at least 32 (the small-step tuning), a power of two, at most 512
    int slices = FX_SELECT_SLICES_MIN;
    while (slices < FX_SELECT_SLICES_MAX && (int64_t)slices * 4096 < max_candidates) slices *= 2;
    // batched launches: keep the grid around a thousand workgroups (every workgroup reduces all of an agent's partials)
    while (slices > FX_SELECT_SLICES_MIN && (int64_t)slices * n_agents > 2048) slices /= 2;
    hipLaunchKernelGGL(fx_select_kernel, dim3(slices, n_agents), dim3(256), 0, stream, d_probs, host_result, seq, dev_winner,
                       host_pkg, pkg_stride, pkg_plane_rows);
    return hipGetLastError();
}

extern "C" hipError_t fx_launch_topk(const DevProblem *d_probs, int n_agents, int64_t max_candidates, int k, double *scr_cost,
                                     long long *scr_idx, double *out_cost, long long *out_idx, hipStream_t stream) {
    // one wave per slice / per agent where the entries fit its registers (fx_topk_*_wave_kernel), the general kernels beyond
    const int64_t per = (max_candidates + FX_TOPK_SLICES - 1) / FX_TOPK_SLICES;
    if (per <= 64 * FX_TOPK_R)
        hipLaunchKernelGGL(fx_topk_slice_wave_kernel, dim3(FX_TOPK_SLICES, n_agents), dim3(64), 0, stream, d_probs, k, scr_cost, scr_idx);
    else
        hipLaunchKernelGGL(fx_topk_slice_kernel, dim3(FX_TOPK_SLICES, n_agents), dim3(256), 0, stream, d_probs, k, scr_cost, scr_idx);
    if (FX_TOPK_SLICES * k <= 64 * FX_TOPK_R)
        hipLaunchKernelGGL(fx_topk_merge_wave_kernel, dim3(n_agents), dim3(64), 0, stream, k, scr_cost, scr_idx, out_cost, out_idx);
    else
        hipLaunchKernelGGL(fx_topk_merge_kernel, dim3(n_agents), dim3(256), 0, stream, k, scr_cost, scr_idx, out_cost, out_idx);
    return hipGetLastError();
}
