// fx_device.h -- device-side problem description shared by the kernels and the host API.
//
// One DevProblem per agent lives in device memory; every field is wave-uniform, so the kernels read it
// with scalar loads.  Pointers are device pointers into the context's arenas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fxplan.h"

#define FX_BLOCK 256            // candidates per workgroup (4 wave64)
#define FX_MODE_INT_STORE_WT (1u << 30)  // internal DevProblem.mode bit: write-through plane stores (fx_set_store_mode)
// internal DevProblem.mode bit: the obstacle stage of this agent runs as its own (candidate x step)-parallel kernel behind the
// walk (fx_obstacle_kernel.h): the walk leaves the cost sum up to the prediction term in cost[], the terms behind it in
// cost_tail[], and no arg-min partial; the obstacle kernel completes cost / flags / cost map and writes the partials
#define FX_MODE_INT_DEFER_OBST (1u << 29)
// internal DevProblem.mode bit: the agent's obstacle record table rec[S][K][12] and its two step masks are staged in the dynamic
// LDS of the lane-split evaluation kernels (the host has sized the launch's LDS for them: fx_api.hip)
#define FX_MODE_INT_REC_LDS (1u << 28)
#define FX_HOT_STRIDE 10        // doubles per (step, obstacle) entry of the hot obstacle table (80 B)
#define FX_HOT_PRE 4            // table elements per lane prefetched one step ahead (covers K <= 25 obstacles)
#define FX_TP 14                // doubles per step of the time table in LDS: t .. t^5, then 2t, 3t^2, 4t^3, 5t^4, 6t, 12t^2, 20t^3, then the
                                // step's obstacle masks (pmask, hmask bit patterns; grid kernel with the staged obstacle stage, else 0)
#define FX_REF_FIELDS 8         // per knot: pos, theta, curv, curv_d, x, y, nx, ny  (64 B, AoS in LDS)
#define FX_MAX_SAMPLES 128      // N+1 <= 128
// lane-split kernels keep the agent's obstacle record table rec[S][K][12] (+ the two step masks) in LDS up to this many bytes
// (fx_eval_grid_kernel.h; K = 5 .. 15 obstacles of a planner-sized step at S = 31: 15 .. 45 KB)
#define FX_REC_LDS_MAX (48 * 1024)
__host__ __device__ inline int S_rec_doubles(int S, int K) { return S * K * 12; }

struct DevProblem {
    // ---- scalars (FxProblem) ----
    int32_t N, S;
    uint32_t mode;
    int32_t low_vel_mode;
    double dt;
    double x0_lon[3], x0_lat[3];
    double x0_orientation, v_des;
    FxVehicle veh;
    int32_t nT, nV, nD;
    int32_t has_matrix;
    int32_t lon_mode;  // FX_LON_*: v_samp holds end velocities (quartic) or end positions (quintic to (s, 0, 0))
    int64_t C;       // candidates of this agent evaluated here (the shard)
    int64_t g_base;  // global index of local candidate 0 (multi-GPU candidate sharding)
    int64_t ld;      // leading dimension of every per-candidate output (C rounded up to 64)
    int32_t M, K, P;
    int32_t n_cost, n_dto;
    int32_t cost_id[FX_NUM_COSTS];
    double cost_w[FX_NUM_COSTS];
    double simpson_corr[3];
    // ---- inputs (device) ----
    const double *tpow;        // [5][S]
    const double *t_samp, *v_samp, *d_samp;
    const double *matrix;      // [C][13] or null
    const double *ref;         // [M][FX_REF_FIELDS]
    const double *obs_pos;     // [K][P][2]
    const double *obs_cov_inv; // [K][P][4]
    const int32_t *obs_npred;  // [K]
    const double *obs_hull;    // [K][P-1][6]
    const int32_t *obs_nhull;  // [K]
    // step-major packed obstacle records for the walk: rec[S][K][12] = {mu_x, mu_y, iv00, iv01, iv10, iv11 of
    // prediction i-1 ; hull (cx, cy, ex, ey, h1, h2) of hull i-2}; pmask/hmask[S]: bit k set <=> obstacle k has a
    // prediction / a hull that ego step i meets.  One contiguous 96-byte record per (step, obstacle).
    const double *obs_rec;
    const unsigned long long *obs_pmask, *obs_hmask;
    // what every (step, obstacle) visit needs, in the form that costs the fewest operations (fx_walk.h, ObsHot), 80 B:
    // hot[S][K][10] = {l11, l12, cu, l22, cw (Cholesky factor of the inverse covariance and the transformed centre),
    // -2 hx, -2 hy, -2 r, |h|^2 - r^2 (hull centre h and circle radius r with slack), 0}, coordinates relative to
    // hot_origin; a wave copies its step's block to LDS with one coalesced load
    const double *obs_hot;
    double hot_origin[2];
    double hot_gap_margin;
    const double *dto_pos;     // [n_dto][2]
    // road boundary: pieces (mid x, mid y, half dx, half dy) and per reference knot the pieces in reach (CSR)
    int32_t n_bound;
    const double *bound_piece;   // [n_bound][4]
    const int32_t *bound_bin;    // [M + 1]
    const int32_t *bound_item;   // [bound_bin[M]]
    double bound_d_reach;
    // ---- outputs (device) ----
    int32_t *bound_step;       // [ld] first step that meets the road boundary, -1 if never (FX_MODE_ROAD_BOUNDARY)
    double *cost;              // [ld]
    double *cost_tail;         // [ld] weighted cost terms ordered behind the prediction term (FX_MODE_INT_DEFER_OBST)
    // obstacle kernel: partial prediction sums [chunks][ld], collision ballots [chunks][tiles], tile tickets
    double *obs_part;
    unsigned long long *obs_colm;
    unsigned int *obs_ticket;
    int32_t *obs_list;         // [ld] local indices of the costed candidates, appended by the walk (counters[FX_DCNT_LIVE] entries)
    uint32_t *flags;           // [ld]
    double *costmap;           // [n_cost][ld]      (FX_MODE_WRITE_COSTMAP)
    double *planes;            // [14][S][ld]       (FX_MODE_WRITE_BUNDLE)
    double *coeffs;            // [FX_COEFF_ROWS][ld]: lon6 | lat6 | delta_tau of the lateral polynomial (FX_MODE_WRITE_BUNDLE)
    int32_t *traj_len;         // [ld]              (FX_MODE_WRITE_BUNDLE)
    // ---- selection scratch ----
    double *part_cost;         // [n_blocks]
    int64_t *part_idx;         // [n_blocks]
    unsigned long long *counters; // [FX_CNT_COUNT]
    int32_t n_blocks;
    // lane_center_offset cost (generic kernel, windowed-cost path): the lanelets in network order (fxplan.h FxProblem.n_lane)
    int32_t n_lane;
    const double *lane_bbox;        // [n_lane][4] = (x min, x max, y min, y max)
    const int32_t *lane_poly_off;   // [n_lane + 1]
    const double *lane_poly;        // [.][2] closed outlines
    const int32_t *lane_ctr_off;    // [n_lane + 1]
    const double *lane_ctr;         // [.][2] centre polylines
    // winner package of this agent in pinned + mapped host memory (fx_tail.h reads these three from the problem in memory, the
    // evaluation kernels never hold them in registers): [pkg_plane_rows] plane values | FX_PKG_TAIL doubles, and its sequence word
    double *pkg_out;
    unsigned long long *pkg_seq;
    int32_t pkg_plane_rows;
};

// The same fields held in registers: a kernel loads them all at entry (one batch of scalar loads, one latency).
// The dynamically indexed arrays become pointers (the kernels stage cost ids / weights in LDS).
struct ProblemRegs {
    int32_t N, S;
    uint32_t mode;
    int32_t low_vel_mode;
    double dt;
    double x0_lon[3], x0_lat[3];
    double x0_orientation, v_des;
    FxVehicle veh;
    int32_t nT, nV, nD;
    int32_t has_matrix;
    int32_t lon_mode;
    int64_t C, g_base, ld;
    int32_t M, K, P;
    int32_t n_cost, n_dto;
    const int32_t *cost_id;
    const double *cost_w;
    const double *simpson_corr;
    const double *tpow;
    const double *t_samp, *v_samp, *d_samp;
    const double *matrix;
    const double *ref;
    const double *obs_pos;
    const double *obs_cov_inv;
    const int32_t *obs_npred;
    const double *obs_rec;
    const unsigned long long *obs_pmask, *obs_hmask;
    const double *obs_hot;
    double hot_origin[2];
    double hot_gap_margin;
    const double *dto_pos;
    int32_t n_bound;
    const double *bound_piece;
    const int32_t *bound_bin;
    const int32_t *bound_item;
    double bound_d_reach;
    int32_t *bound_step;
    double *cost;
    double *cost_tail;
    int32_t *obs_list;
    uint32_t *flags;
    double *costmap;
    double *planes;
    double *coeffs;
    int32_t *traj_len;
    double *part_cost;
    int64_t *part_idx;
    unsigned long long *counters;
    int32_t n_blocks;
    int32_t n_lane;
    const double *lane_bbox;
    const int32_t *lane_poly_off;
    const double *lane_poly;
    const int32_t *lane_ctr_off;
    const double *lane_ctr;

    __device__ __forceinline__ static ProblemRegs load(const DevProblem &g, const int32_t *cost_id, const double *cost_w) {
        ProblemRegs r;
        r.N = g.N; r.S = g.S; r.mode = g.mode; r.low_vel_mode = g.low_vel_mode; r.dt = g.dt;
        r.x0_lon[0] = g.x0_lon[0]; r.x0_lon[1] = g.x0_lon[1]; r.x0_lon[2] = g.x0_lon[2];
        r.x0_lat[0] = g.x0_lat[0]; r.x0_lat[1] = g.x0_lat[1]; r.x0_lat[2] = g.x0_lat[2];
        r.x0_orientation = g.x0_orientation; r.v_des = g.v_des; r.veh = g.veh;
        r.nT = g.nT; r.nV = g.nV; r.nD = g.nD; r.has_matrix = g.has_matrix; r.lon_mode = g.lon_mode;
        r.C = g.C; r.g_base = g.g_base; r.ld = g.ld; r.M = g.M; r.K = g.K; r.P = g.P;
        r.n_cost = g.n_cost; r.n_dto = g.n_dto;
        r.cost_id = cost_id; r.cost_w = cost_w; r.simpson_corr = g.simpson_corr;
        r.tpow = g.tpow; r.t_samp = g.t_samp; r.v_samp = g.v_samp; r.d_samp = g.d_samp; r.matrix = g.matrix;
        r.ref = g.ref; r.obs_pos = g.obs_pos; r.obs_cov_inv = g.obs_cov_inv; r.obs_npred = g.obs_npred;
        r.obs_rec = g.obs_rec; r.obs_pmask = g.obs_pmask; r.obs_hmask = g.obs_hmask; r.obs_hot = g.obs_hot; r.dto_pos = g.dto_pos;
        r.hot_origin[0] = g.hot_origin[0]; r.hot_origin[1] = g.hot_origin[1]; r.hot_gap_margin = g.hot_gap_margin;
        r.n_bound = g.n_bound; r.bound_piece = g.bound_piece; r.bound_bin = g.bound_bin; r.bound_item = g.bound_item;
        r.bound_d_reach = g.bound_d_reach; r.bound_step = g.bound_step;
        r.cost = g.cost; r.cost_tail = g.cost_tail; r.obs_list = g.obs_list; r.flags = g.flags; r.costmap = g.costmap; r.planes = g.planes; r.coeffs = g.coeffs;
        r.traj_len = g.traj_len; r.part_cost = g.part_cost; r.part_idx = g.part_idx; r.counters = g.counters;
        r.n_blocks = g.n_blocks;
        r.n_lane = g.n_lane; r.lane_bbox = g.lane_bbox; r.lane_poly_off = g.lane_poly_off; r.lane_poly = g.lane_poly;
        r.lane_ctr_off = g.lane_ctr_off; r.lane_ctr = g.lane_ctr;
        return r;
    }
};

// Developer probe (-DFX_PROBE): per-wave cycle stamps at the phase boundaries of the evaluation kernels, read back
// with fx_probe_read (tools/probe_phases.py).  Compiles to nothing in the product build.
#ifdef FX_PROBE
#define FX_PROBE_SLOTS 16
#define FX_PROBE_WAVES (1 << 16)
extern __device__ unsigned long long fx_probe_stamps[FX_PROBE_WAVES * FX_PROBE_SLOTS];
#define FX_STAMP(k)                                                                                              \
    do {                                                                                                         \
        const unsigned w_ = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));                               \
        if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && w_ < FX_PROBE_WAVES)                                   \
            fx_probe_stamps[(size_t)w_ * FX_PROBE_SLOTS + (k)] = (k) == 0 || (k) == 15 || FX_PROBE == 2 ? wall_clock64() : clock64(); \
    } while (0)
// the obstacle kernel's own stamp block (it runs behind the walk and would overwrite the walk's stamps)
extern __device__ unsigned long long fx_probe_stamps_obs[FX_PROBE_WAVES * FX_PROBE_SLOTS];
#define FX_OSTAMP(k)                                                                                             \
    do {                                                                                                         \
        const unsigned w_ = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));                               \
        if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && w_ < FX_PROBE_WAVES && (FX_PROBE != 4 || (k) == 0 || (k) == 15 || (k) == 4))  \
            fx_probe_stamps_obs[(size_t)w_ * FX_PROBE_SLOTS + (k)] = (k) == 0 || (k) == 15 || FX_PROBE >= 3 ? wall_clock64() : clock64(); \
    } while (0)
// the one-launch step's phase boundaries (fx_step_kernel.h), wall clock, in the obstacle kernel's block
#define FX_MSTAMP(k)                                                                                             \
    do {                                                                                                         \
        const unsigned w_ = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));                               \
        if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && w_ < FX_PROBE_WAVES)                                   \
            fx_probe_stamps_obs[(size_t)w_ * FX_PROBE_SLOTS + (k)] = wall_clock64();                             \
    } while (0)
#else
#define FX_STAMP(k) do { } while (0)
#define FX_OSTAMP(k) do { } while (0)
#define FX_MSTAMP(k) do { } while (0)
#endif

// rows of the per-candidate coefficient table: lon[6] | lat[6] | tau_lat (the lateral polynomial's delta_tau: t, or s_lon_goal in
// LOW_VEL_MODE -- reactive_planner.py:161-171)
#define FX_COEFF_ROWS 13
// winner package tail behind the [14][S] planes (doubles): lon6 lat6 | raw[FX_NUM_COSTS] | cost | traj_len | flags | index | found | tau_lat
#define FX_PKG_TAIL (12 + FX_NUM_COSTS + 6)

// counters[] layout
enum {
    FX_CNT_RETURNED = 0, FX_CNT_FEASIBLE, FX_CNT_HIST0, /* 11 entries */
    FX_CNT_BEST_IDX = FX_CNT_HIST0 + FX_NUM_REASONS, FX_CNT_BEST_COST, FX_CNT_COLLISIONS, FX_CNT_COUNT
};

// The device-side counters[] slot FX_CNT_BEST_IDX is free (the winner only exists in the published block): the
// evaluation kernel's workgroups take completion tickets from it when the selection is fused into the kernel.
#define FX_DCNT_TICKET FX_CNT_BEST_IDX
// ... and FX_CNT_BEST_COST counts the entries of the obstacle kernel's candidate list (deferred obstacle stage): the walk's waves
// append their costed candidates, the obstacle kernel reads the count, the selection's publishing workgroup zeroes it
#define FX_DCNT_LIVE FX_CNT_BEST_COST
// Fused selection (no agent asks for the collision stage): the LAST workgroup of an agent to finish reduces the
// per-workgroup partials and publishes the step's result block, so the step is one launch.  host_result == nullptr:
// a separate fx_select_kernel follows.
// tail (fx_tail.h): what the agent's last workgroup does beyond the arg-min -- FX_TAIL_COUNT: the colliding candidates in front of
// the winner (planner.py:336-357; the step's kernels store cost[] / flags[] write-through), FX_TAIL_PACKAGE: the winner package
// gathered into host_pkg (the bundle, coefficients and cost map are stored write-through).  0 = the single-wave publication of
// steps without a collision stage and without a package.
#define FX_TAIL_COUNT 1u
#define FX_TAIL_PACKAGE 2u
// which evaluation kernels carry the tail: the planner-sized decompositions (>= 4 lanes per candidate: steps below 200 waves) and
// the windowed-cost kernel (one lane per candidate, small by construction); the host asks for a tail only with those
#define FX_TAIL_IN_KERNEL(G, EXTRA) ((G) >= 4 || (EXTRA))
struct FuseArgs {
    unsigned long long *host_result;  // pinned + mapped [n_agents][FX_CNT_COUNT + 1]
    unsigned long long seq;           // sequence word the host polls for
    double *dev_winner;               // optional device copy of (cost, index) per agent
    int32_t k_max;                    // largest obstacle count of the launch's agents (sizes the per-wave hot blocks)
                                      // ... and, in bits 16 / 17, FX_TAIL_* (only with host_result): the struct is a kernel argument
                                      // that the walk keeps in scalar registers, every further word costs the big kernels a spill
    __host__ __device__ uint32_t tail() const { return (uint32_t)k_max >> 16; }
    __host__ __device__ int32_t kmax() const { return k_max & 0xffff; }
};

// arguments of the one-launch step (fx_step_kernel.h) beyond the walk's
struct StepArgs {
    unsigned long long *bar;        // two barrier blocks of FX_BAR_WORDS words (fx_step_kernel.h), monotonic: never reset
    unsigned long long bar_base;    // their value before this launch (the host counts: + workgroups of the launch per step)
    unsigned long long *host_result;
    unsigned long long seq;
    double *dev_winner;
    double *host_pkg;
    int32_t pkg_stride, pkg_plane_rows;
    int32_t walk_blocks;            // workgroups [0, walk_blocks) of every agent walk candidates
};

// Pointers stored inside DevProblem are loaded from memory, so the compiler only knows them as generic ("flat")
// pointers: every access would be a flat_load/flat_store and wave-uniform reads could not become scalar loads.
// They all point into hipMalloc'ed memory -- say so.
#define FX_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ FX_GLOBAL T *as_global(T *p) {
    return (FX_GLOBAL T *)p;
}
// Input tables that NO kernel of the step writes, as pointers into the constant address space: a wave-uniform read stays a scalar
// load even behind an atomic, a fenced barrier or a wait loop.  (The compiler turns a uniform load from the GLOBAL address space into
// a scalar load only while nothing in front of it in the kernel may have written memory; behind the one-launch step's grid barrier
// every uniform table read became a vector load with a round trip and a register pair of its own -- profiles/r6/NOTES.md.)
#define FX_CONSTAS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const FX_CONSTAS T *as_const(const T *p) {
    return (const FX_CONSTAS T *)p;
}
template <bool CONSTANT, typename T>
__device__ __forceinline__ auto as_table(const T *p) {
    if constexpr (CONSTANT) return as_const(p);
    else return as_global(p);
}
