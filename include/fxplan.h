/*
 * fxplan.h -- C-ABI of libfxplan.so, the MI355X (gfx950) Frenet sampling-and-evaluation engine.
 *
 * The reference (TUM-AVS/Frenetix-Motion-Planner @ 2024_10_08) has no C-ABI: its hot path is
 * reached through the pybind11 module `frenetix` (C++ wheel, not in tree) that
 * frenetix_motion_planner/reactive_planner_cpp.py drives, or through the pure-Python
 * frenetix_motion_planner/reactive_planner.py.  This header is the boundary a maintainer binds
 * instead (ctypes stub: INTEGRATION.md).  Every entry point cites the reference interface it
 * replaces.  Plain pointers and sizes only; all floating point is IEEE binary64.
 *
 * Semantics follow the reference *Python* path (reactive_planner.py:132-577,
 * trajectories.py:524-561, cost_function.py:78-91, planner.py:329-392); the `frenetix` handler is
 * only the calling convention.  See DESIGN.md for the normative definitions of the two pieces
 * of third-party arithmetic that are not in the reference tree (curvilinear->Cartesian projection
 * and the OBB-sum collision test).
 */
#ifndef FXPLAN_H
#define FXPLAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_ABI_VERSION 9

/* ---- status codes (planner.py / reactive_planner_cpp.py raise Python exceptions; the shim maps
 *      <0 -> ValueError, >0 -> RuntimeError, see SURVEY 8b "Error conventions") ---- */
#define FX_OK 0
#define FX_ERR_INVALID_ARGUMENT (-1)
#define FX_ERR_NOT_READY (-2)     /* reference / vehicle / sampling not set before a plan step */
#define FX_ERR_CAPACITY (-3)      /* problem larger than the context was created for */
#define FX_ERR_HIP 1              /* a HIP runtime call failed; text in fx_last_error() */
#define FX_ERR_NO_DEVICE 2
#define FX_ERR_TIMEOUT 3          /* no answer from the device within the context's time bound (fx_set_timeout_ms): a peer that never
                                   * joined a collective or a faulted kernel.  The context refuses further steps; destroy it (and, in a
                                   * multi-rank job, end the process: a fresh child may be started, a process that touched the GPU is
                                   * never re-exec'ed).  The reference bounds its inter-process hand-offs the same way, TIMEOUT = 20 s:
                                   * cr_scenario_handler/simulation/simulation.py:637,655, agent_batch.py:98 */

/* ---- trajectory planes of the structure-of-arrays TrajectoryBundle.
 *      trajectories.py:56-197 (CartesianSample) and :200-334 (CurviLinearSample).
 *      Device layout: plane[p][step][ld] (ld = candidate count rounded up to 64), f64. ---- */
enum {
    FX_PL_X = 0, FX_PL_Y, FX_PL_THETA, FX_PL_V, FX_PL_A, FX_PL_KAPPA, FX_PL_KAPPA_DOT,
    FX_PL_S, FX_PL_D, FX_PL_THETA_CL, FX_PL_S_DOT, FX_PL_S_DDOT, FX_PL_D_DOT, FX_PL_D_DDOT,
    FX_NUM_PLANES
};

/* ---- partial cost functions (partial_cost_functions.py).  Ids are in alphabetical order of the
 *      reference names because cost_function.py:55-60 sorts the active names; callers list the
 *      active ids ascending. ---- */
enum {
    FX_COST_ACCELERATION = 0,            /* :24-33  */
    FX_COST_DISTANCE_TO_OBSTACLES,       /* :172-186 */
    FX_COST_DISTANCE_TO_REFERENCE_PATH,  /* :154-169 */
    FX_COST_JERK,                        /* :36-46  */
    FX_COST_LANE_CENTER_OFFSET,          /* :91-117 (needs the lanelets: FxProblem.n_lane ...; DESIGN.md 4.4, parity unpinned) */
    FX_COST_LATERAL_JERK,                /* :49-55  */
    FX_COST_LONGITUDINAL_JERK,           /* :58-64  */
    FX_COST_ORIENTATION_OFFSET,          /* :141-151 */
    FX_COST_PATH_LENGTH,                 /* :189-196 */
    FX_COST_PREDICTION,                  /* :341-356 + collision_probability.py:264-299 */
    FX_COST_VELOCITY_OFFSET,             /* :120-130 */
    FX_NUM_COSTS
};

/* ---- per-candidate flag word ---- */
#define FX_FLAG_VALID      (1u << 0)  /* reactive_planner.py:293,350-351,544 */
#define FX_FLAG_FEASIBLE   (1u << 1)  /* :292 and the five constraints :480-533 */
#define FX_FLAG_COLLISION  (1u << 2)  /* planner.py:348-357 (dynamic-obstacle prediction check) */
#define FX_FLAG_RETURNED   (1u << 3)  /* member of check_feasibility's return list (:353,379,385,567) */
#define FX_FLAG_COSTED     (1u << 4)  /* cost evaluated (:244-253: all returned in debug, feasible otherwise) */
#define FX_FLAG_SELECTABLE (1u << 5)  /* walked by trajectory_collision_check (:248,251,258) */
#define FX_FLAG_BOUNDARY   (1u << 6)  /* planner.py:362-381: the ego footprint leaves the road (boundary_harm != 0) */
#define FX_REASON_SHIFT 8             /* bits 8..18: reasons 0..10 flagged as in :352,378,384,418,485-531,545 */
#define FX_REASON_MASK  (0x7FFu << FX_REASON_SHIFT)
#define FX_NUM_REASONS 11

/* evaluation mode bits (configurations/frenetix_motion_planner/debug.yaml) */
#define FX_MODE_DRAW_TRAJ_SET   (1u << 0)  /* debug.draw_traj_set : evaluate everything, no pre-filter */
#define FX_MODE_KINEMATIC_DEBUG (1u << 1)  /* debug.kinematic_debug: keep checking after first violation */
#define FX_MODE_WRITE_BUNDLE    (1u << 2)  /* materialise the 14-plane SoA TrajectoryBundle in HBM */
#define FX_MODE_WRITE_COSTMAP   (1u << 3)  /* keep the per-name raw costs (TrajectorySample.costMap) */
#define FX_MODE_COLLISION       (1u << 4)  /* run the OBB collision stage (planner.use_prediction) */
#define FX_MODE_ROAD_BOUNDARY   (1u << 5)  /* test the ego footprint against the road boundary (planner.py:362-381) */
/* (s, d) -> (x, y) offsets d along the UN-normalised interpolated vertex normal (d is a pseudo-distance) instead of along its
 * unit vector -- the two readings of CCosy's construction differ by up to 7 mm on the ZAM_Tjunction route (DESIGN.md 4.1;
 * tests/golden/pin_third_party.py decides which one the reference's CCosy is, once commonroad_dc is importable) */
#define FX_MODE_PROJ_PSEUDO_NORMAL (1u << 6)

/* longitudinal sampling mode.
 * FX_LON_VELOCITY_KEEPING: v_samp = sampled end velocities, quartic to (v, 0) -- _create_trajectory_bundle,
 *   reactive_planner.py:132-182.
 * FX_LON_STOP_POINT: v_samp = sampled end POSITIONS s, quintic to (s, 0, 0) -- stop-point sampling,
 *   _create_end_point_trajectory_bundle, reactive_planner.py:628-671 (cpp: generate_stopping_trajectories,
 *   reactive_planner_cpp.py:258-290).  Ranges only (the C x 13 matrix has no end-position column). */
#define FX_MAX_OBSTACLES 256   /* predicted obstacles per agent (FxProblem.K) */
#define FX_LON_VELOCITY_KEEPING 0
#define FX_LON_STOP_POINT 1

/* vehicle parameters: configuration.py:58-83 (VehicleConfiguration); kappa_max is
 * tan(delta_max)/wheelbase as computed at reactive_planner.py:492 (host computes it once). */
typedef struct FxVehicle {
    double a_max, v_switch, delta_max, wheelbase, length, width, wb_rear_axle, kappa_max;
} FxVehicle;

/* Everything a plan step needs that is shared by all candidates of one agent.
 * reactive_planner.py:67-130 (plan), planner.py:172-217 (update_externals). */
typedef struct FxProblem {
    /* horizon: planner.py:63-65 */
    int32_t N;              /* steps; every trajectory has S = N+1 samples */
    double dt;
    uint32_t mode;          /* FX_MODE_* */
    int32_t low_vel_mode;   /* planner.py:222-230 */
    int32_t lon_mode;       /* FX_LON_*: what the longitudinal samples (v_samp) are */
    double x0_lon[3];       /* x_cl[0] = [s, s_dot, s_ddot]   planner.py:567-635 */
    double x0_lat[3];       /* x_cl[1] = [d, d_dot, d_ddot] */
    double x0_orientation;  /* x_0.orientation, used at reactive_planner.py:447 */
    double v_des;           /* desired_velocity (cost_function.py:70) */
    FxVehicle veh;

    /* time grid: t[i]=round(arange(0,..,dt),5) and its powers rounded to 10 dp
     * (reactive_planner.py:296-300).  tpow[k*S + i] = t_i^(k+1), k = 0..4, i = 0..S-1.
     * Host-computed with the reference's own expressions so the device never calls pow(). */
    const double *tpow;

    /* sampling: ordered ranges in *iteration order* (sampling_matrix.py:141-195; CPython set
     * order, reactive_planner.py:149-158).  Candidate g = (it*nV + iv)*nD + id.
     * v_samp holds end velocities or, with FX_LON_STOP_POINT, end positions (:641-643). */
    int32_t nT, nV, nD;
    const double *t_samp, *v_samp, *d_samp;
    /* ...or an explicit C x 13 matrix (sampling_matrix.py:85-121, reactive_planner_cpp.py:228-253)
     * when sampling_matrix != NULL; then nT/nV/nD are ignored and C = n_rows. */
    const double *sampling_matrix;
    int64_t n_rows;
    /* candidate shard [shard_begin, shard_begin + shard_count) of the global grid evaluated by this
     * context (multi-GPU candidate sharding); shard_count == 0 means the whole grid.
     * Per-candidate outputs are indexed locally (0..shard_count), best_index / top-k stay global so
     * that the (cost, index) tie-break is the same on every rank. */
    int64_t shard_begin, shard_count;

    /* reference path: utils_coordinate_system.py:189-207 (ref_pos/ref_theta/ref_curv/ref_curv_d),
     * plus vertices and vertex normals for the projection (DESIGN.md "projection"). */
    int32_t M;
    const double *ref_x, *ref_y, *ref_nx, *ref_ny, *ref_pos, *ref_theta, *ref_curv, *ref_curv_d;

    /* cost function: cost_function.py:40-64.  ids ascending (== name-sorted). */
    int32_t n_cost;
    const int32_t *cost_id;
    const double *cost_w;
    double simpson_corr[3]; /* alpha, beta, eta of scipy.integrate.simpson's even-N correction for h=[dt,dt] */

    /* predictions (prediction_helpers.py:209-261 dict -> packed): K obstacles, P steps each.
     * obs_pos[K][P][2], obs_cov_inv[K][P][4] (np.linalg.inv(cov_list), row-major 2x2),
     * obs_npred[K] = len(pos_list): the REAL length of the prediction -- it decides which ego steps see the obstacle
 * (collision_probability.py:287) and may exceed the stride P, which only bounds what is stored and read (a predictor with a longer
 * horizon than the planner's does not enlarge the tables; fx_pack_predictions).  collision_probability.py:264-299.
     * K <= FX_MAX_OBSTACLES; up to 64 obstacles the grid kernel applies, beyond that the step runs on the generic kernel
     * (the per-step obstacle masks take one 64-bit word per 64 obstacles). */
    int32_t K, P;
    const double *obs_pos, *obs_cov_inv;
    const int32_t *obs_npred;
    /* collision stage: OBB hulls of consecutive predicted boxes, obs_hull[K][P-1][6] =
     * (cx, cy, ex, ey, h1, h2) with unit axis e and half extents h; obs_nhull[K] = number of hulls
     * (0 when the obstacle is skipped, collision_check.py:165-168).  Built by fx_build_obstacle_hulls. */
    const double *obs_hull;
    const int32_t *obs_nhull;
    /* distance_to_obstacles cost: obstacle positions at the current step, dto_pos[n_dto][2]
     * (partial_cost_functions.py:179-184). */
    int32_t n_dto;
    const double *dto_pos;
    /* road boundary (planner.py:362-381; built once per scenario, :550-565): n_bound straight pieces
     * bound_piece[n_bound][4] = (mid x, mid y, half dx, half dy), and per reference knot k the pieces that can
     * touch an ego footprint whose foot point lies on segment k: bound_item[bound_bin[k] .. bound_bin[k+1])
     * (CSR, M+1 offsets).  Built by fx_build_boundary_bins.  DESIGN.md 4.3 is the normative test. */
    int32_t n_bound;
    const double *bound_piece;
    const int32_t *bound_bin;
    const int32_t *bound_item;
    double bound_d_reach;   /* lateral reach the bins were built for; |d| beyond it counts as off the road */
    /* lane_center_offset cost (partial_cost_functions.py:91-117): the lanelets in network order.  Per trajectory point the
     * FIRST lanelet whose outline contains it (lane_poly[lane_poly_off[l] .. lane_poly_off[l+1]) = left vertices, then the
     * right ones reversed; ray casting; lane_bbox[l] = (x min, x max, y min, y max) settles most lanelets) and the distance
     * to that lanelet's centre polyline lane_ctr[lane_ctr_off[l] .. lane_ctr_off[l+1]) -- the closest point of its segments;
     * 5 when no lanelet contains the point; the cost is the mean over the trajectory's points.  The reference asks
     * commonroad-io (find_lanelet_by_position) and shapely (project / interpolate), neither of which is in the reference
     * tree: which lanelet is "first" where lanelets overlap, and points exactly on an outline, are this header's definition. */
    int32_t n_lane;
    const double *lane_bbox;       /* [n_lane][4] */
    const int32_t *lane_poly_off;  /* [n_lane + 1] */
    const double *lane_poly;       /* [lane_poly_off[n_lane]][2] */
    const int32_t *lane_ctr_off;   /* [n_lane + 1] */
    const double *lane_ctr;        /* [lane_ctr_off[n_lane]][2] */
} FxProblem;

/* Result of one plan step (what _get_optimal_trajectory returns plus the counters it sets,
 * reactive_planner.py:184-272; planner.py:329-392). */
typedef struct FxResult {
    int64_t n_candidates;
    int64_t best_index;        /* uniqueId of the selected trajectory, -1 if none */
    double best_cost;
    int64_t n_returned;        /* len(trajectories_all) */
    int64_t n_feasible;        /* len(feasible_trajectories) (valid and feasible) */
    int64_t n_infeasible;      /* _infeasible_count_kinematics[0] */
    int64_t n_collisions;      /* _collision_counter: colliding candidates walked before the winner */
    int64_t reason_hist[FX_NUM_REASONS]; /* infeasible_invalid_count_kinematics (queue_2 payload) */
    double feasible_percentage;/* infeasible_kinematics_percentage (is the feasible %, :235) */
    double kernel_ms;          /* HIP-event time of the device work of this step; -1 when the step was not timed or
                                * its events were still pending when fx_finish returned (fx_last_kernel_ms waits) */
} FxResult;

typedef struct FxContext FxContext;

/* ---- library ---- */
int32_t fx_abi_version(void);
const char *fx_last_error(void);          /* thread-local text of the last failure */
int32_t fx_device_count(int32_t *count);  /* no context needed */

/* ---- context: owns device buffers sized for up to max_candidates x (max_steps+1); one per
 *      planner instance (frenetix.TrajectoryHandler(dt=) at reactive_planner_cpp.py:49).
 *      stream: a hipStream_t as void* (0 -> the context creates its own). ---- */
int32_t fx_create(FxContext **out, int32_t device, int64_t max_candidates, int32_t max_steps,
                  int32_t max_ref_knots, int32_t max_obstacles, int32_t max_pred_steps);
int32_t fx_destroy(FxContext *ctx);
int32_t fx_set_stream(FxContext *ctx, void *hip_stream);
/* tuning override (0 = automatic): lanes that share one candidate's horizon (1, 2, 4, 8, 16, 32), the occupancy
 * target in waves per SIMD (2..4) of the evaluation kernel, and the kernel variant (1 = generic per-candidate
 * kernel, 2 = grid kernel with the per-(t,v) longitudinal table).  Results do not depend on any of them. */
int32_t fx_set_tuning(FxContext *ctx, int32_t lanes_per_candidate, int32_t waves_per_simd, int32_t kernel_variant);
/* device buffer [n_agents][2] = (cost f64, global index i64) that every evaluated step also leaves its winner in,
 * so a k = 1 survivor exchange needs no extra kernel (NULL to disable); e.g. a torch tensor's data_ptr() */
int32_t fx_set_winner_buffer(FxContext *ctx, void *d_winner);
/* publish n doubles of a small device buffer (the all-gathered survivors) to the host through pinned memory on the
 * context stream, and wait for / copy them out: avoids a D2H copy + stream synchronisation per step */
int32_t fx_publish(FxContext *ctx, const void *d_src, int32_t n);
int32_t fx_wait_published(FxContext *ctx, double *out);
/* how the parts of a split horizon map to lanes: 0 auto, 1 adjacent lanes (shuffle combine), 2 lane-groups of the
 * workgroup ("wave split", LDS combine; keeps every plane store a contiguous row segment per wave) */
int32_t fx_set_part_mapping(FxContext *ctx, int32_t mapping);
/* workgroup size of the grid kernel: 0 (auto), 64, 128 or 256 lanes */
int32_t fx_set_block_size(FxContext *ctx, int32_t block_size);
/* how the bundle's plane stores leave the CU: 0 auto (by bundle size), 1 plain write-back stores, 2 write-through
 * (agent-scope) stores -- nothing dirty is left in the L2s for the end-of-kernel write-back, which pays off while the
 * whole bundle is small (measured on MI355X: 42 vs 45 us at 175 MB, 780 vs 650 us at 3.5 GB).  Results are unaffected. */
int32_t fx_set_store_mode(FxContext *ctx, int32_t store_mode);

/* where the obstacle stage (prediction cost collision_probability.py:264-299 + OBB-sum collision walk planner.py:329-392) runs:
 * 0 auto, 1 fused into the horizon walk, 2 as its own (candidate x step)-parallel kernel behind the walk -- it reads x / y /
 * theta back from the materialised planes, so it needs FX_MODE_WRITE_BUNDLE, at most 64 obstacles, no road-boundary stage and
 * no windowed cost term (FX_ERR_INVALID_ARGUMENT at the next upload otherwise).  steps_per_item (0 auto, 2, 3, 5): horizon steps
 * one wave of that kernel visits.  Decisions (flags, counters, winner) do not depend on either; the prediction cost is summed
 * in a different order (agreement 1e-13 relative). */
int32_t fx_set_obstacle_stage(FxContext *ctx, int32_t stage, int32_t steps_per_item);

/* ---- staging: copy the shared inputs of a plan step to the device (borrowed for the call).
 *      Replaces handler.generate_trajectories(matrix, low_vel_mode) + the functor registration
 *      at reactive_planner_cpp.py:96-178,256. ---- */
int32_t fx_upload(FxContext *ctx, const FxProblem *prob);

/* ---- evaluation: fused sample -> solve -> evaluate -> Frenet->Cartesian -> feasibility -> cost
 *      -> collision launch plus the (cost, index) selection.  Replaces
 *      handler.evaluate_all_current_functions(True) (:347-349) and, in the Python path,
 *      check_feasibility + TrajectoryBundle.sort + trajectory_collision_check.
 *      fx_evaluate only enqueues on the context stream; fx_finish synchronises and fills *res. ---- */
int32_t fx_evaluate(FxContext *ctx);
int32_t fx_finish(FxContext *ctx, FxResult *res);
/* convenience: upload + evaluate + finish */
int32_t fx_plan_step(FxContext *ctx, const FxProblem *prob, FxResult *res);
/* evaluate + finish for inputs that are already resident (the handler re-evaluating its current trajectory set,
 * reactive_planner_cpp.py:345-349): one call across the boundary per plan step; res[n_agents] */
int32_t fx_step(FxContext *ctx, FxResult *res);

/* ---- per-step update of a planner that keeps its reference path, grid shape and cost function: the new ego state,
 *      desired velocity, sampling values and predictions of planner.py:172-217 (update_externals) / reactive_planner_cpp.py:
 *      56-86 (set_predictions) rewritten in place in the context's pinned staging block; the range that
 *      changed is staged in front of the next evaluation (a copy kernel reading the mapped block; the DMA engine above 1 MiB).  NULL pointers, NaN doubles and a negative low_vel_mode keep the uploaded
 *      value; array lengths (nT, nV, nD, K, P) are those of the upload.  fx_update_step = fx_update_state(agent 0) + fx_step. */
typedef struct FxStateUpdate {
    const double *x0_lon, *x0_lat;             /* [3] each */
    double x0_orientation, v_des;
    int32_t low_vel_mode;
    const double *t_samp, *v_samp, *d_samp;
    const double *obs_pos, *obs_cov_inv;       /* [K][P][2], [K][P][4] */
    const int32_t *obs_npred;                  /* [K] */
    const double *obs_hull;                    /* [K][P-1][6] */
    const int32_t *obs_nhull;                  /* [K] */
    /* the array shapes as the CALLER holds them (0 = not stated, nothing is checked): the library copies nT / nV / nD doubles and
     * K x P obstacle rows out of the borrowed pointers with the counts of the UPLOAD -- a caller whose arrays were built for another
     * grid or another (K, P) would be read past their end.  A stated count that differs from the upload's is refused with
     * FX_ERR_INVALID_ARGUMENT before anything is rewritten (upload again). */
    int32_t nT, nV, nD, K, P;
} FxStateUpdate;
int32_t fx_update_state(FxContext *ctx, int32_t agent, const FxStateUpdate *upd);
int32_t fx_update_step(FxContext *ctx, const FxStateUpdate *upd, FxResult *res);

/* ---- multi-GPU survivor exchange inside the library (one process per GPU; the reference fans candidate chunks / agent batches
 *      out to processes and pickles results back through Queues, reactive_planner.py:197-224, simulation.py:449-470) ----
 *      The context gets an RCCL communicator of its own: rank 0 draws the 128-byte id (fx_comm_unique_id), the host program
 *      broadcasts it by whatever means it has, every rank calls fx_comm_init.  fx_step_exchange = fx_evaluate + ONE all-gather of
 *      every rank's winner (cost f64, global index i64) per agent on the context's stream + publication to pinned host memory +
 *      fx_finish_batch: cost / index [world][agent rows], index -1 where a rank found nothing.  RCCL is bound at run time
 *      (librccl.so.1); without it these return FX_ERR_NOT_READY and everything else works. */
int32_t fx_comm_unique_id(uint8_t *id128);
/* local preconditions of fx_comm_init (RCCL present, capacity) WITHOUT entering anything collective: every rank calls it and the
 * ranks agree (all-reduce MIN over the host program's own group) before any of them calls fx_comm_init -- a rank that would fail
 * there never reaches ncclCommInitRank and its peers would wait for it forever */
int32_t fx_comm_check(const FxContext *ctx, int32_t world);
int32_t fx_comm_init(FxContext *ctx, const uint8_t *id128, int32_t rank, int32_t world);
int32_t fx_comm_destroy(FxContext *ctx);
/* the agent rows EVERY rank contributes to an exchange (default: the context's max_agents).  The element count of the all-gather
 * is a property of the communicator, agreed between the ranks and fixed here -- not of a rank's current upload: a rank that
 * uploaded fewer agents sends "no survivor" (inf, -1) in the remaining rows, a rank that uploaded more gets FX_ERR_CAPACITY
 * from the exchange (after having entered it) */
int32_t fx_comm_set_agents(FxContext *ctx, int32_t n_agents);
/* where the exchanges' all-gather lands: 0 (default) a device buffer + the publication kernel, 1 straight in the pinned, mapped
 * block with a stream-ordered write of the sequence word (no launch behind the collective).  Use mode 1 only after it has
 * agreed with an independent exchange on every rank (distributed.ShardedEvaluator does that). */
int32_t fx_set_exchange_mode(FxContext *ctx, int32_t mode);
/* out[0] rank, [1] world, [2] the ranks RCCL itself reports for the communicator (ncclCommCount, -1 if unavailable), [3] agent
 * rows per rank */
int32_t fx_comm_info(const FxContext *ctx, int32_t *out4);
/* A = the communicator's agent rows (fx_comm_set_agents).  A rank whose own evaluation fails, or whose upload does not fit A
 * rows, still enters the all-gather (with cost inf / index -1 in its rows) and returns its error afterwards: nothing that can
 * fail locally returns ahead of the collective, the peers are never left waiting */
int32_t fx_step_exchange(FxContext *ctx, FxResult *res, double *cost /*[world][A]*/, int64_t *index /*[world][A]*/);
/* the same for the k <= 64 best survivors per agent (agent sharding with a top-k gather, BASELINE config 5): evaluation,
 * selection, top-k, ONE all-gather of 16 k bytes per rank and agent row, publication -- no host code between the launches; k is
 * the same on every rank */
int32_t fx_step_exchange_topk(FxContext *ctx, int32_t k, FxResult *res, double *cost /*[world][A][k]*/,
                              int64_t *index /*[world][A][k]*/);
/* every host wait on device work (fx_finish, fx_wait_published, the exchanges, fx_read_package) is bounded in TIME: default
 * 20 000 ms, the reference's TIMEOUT (simulation.py:637); FX_ERR_TIMEOUT when it runs out.  fx_wait_word is the wait itself
 * (pure host code): returns FX_OK once *word == expected, FX_ERR_TIMEOUT after timeout_ms. */
int32_t fx_set_timeout_ms(FxContext *ctx, int32_t timeout_ms);
int32_t fx_wait_word(const volatile unsigned long long *word, unsigned long long expected, int32_t timeout_ms);

/* ---- the chosen trajectory, packaged (planner.py:394-447 _compute_trajectory_pair; reactive_planner_cpp.py:355-357 reads the
 *      optimal trajectory's arrays; frenet_interface.py:243-277 consumes the pair) ----
 *      With fx_set_package(ctx, 1) every evaluation that writes the bundle is followed by a gather of the winner's data into
 *      pinned host memory on the same stream, and fx_finish returns once it has arrived: no strided read-back, no second
 *      synchronisation.  fx_read_package hands it out: `pkg` = index (global), cost, flag word, horizon, coefficients, raw
 *      partial costs; `block` (may be NULL) = [FX_PKG_ROWS][S] doubles: the FX_NUM_PLANES planes in PLANE order, then
 *      yaw_rate (yaw_rate0, then backward differences of theta over dt), steering_angle atan2(wheelbase * kappa, 1) and the
 *      heading shifted into [x0_orientation - pi, x0_orientation + pi].  found = 0: no selectable collision-free candidate.
 *      fx_plan_and_package = fx_update_state(agent 0, upd; upd may be NULL) + fx_evaluate + fx_finish + fx_read_package in ONE
 *      call: a planner's closed-loop plan step. */
#define FX_PKG_ROWS (FX_NUM_PLANES + 3)
typedef struct FxPackage {
    int32_t found, S, traj_len;
    uint32_t flags;
    int64_t index;
    double cost;
    double coeff_lon[6], coeff_lat[6];
    int32_t n_cost, reserved;
    double raw_costs[FX_NUM_COSTS];
    double tau_lat;   /* delta_tau of the lateral polynomial: t, or in LOW_VEL_MODE s_lon_goal (reactive_planner.py:161-171, :650-659);
                       * the longitudinal polynomial's delta_tau is always t = sampling_parameters[1] - [0] */
} FxPackage;
int32_t fx_set_package(FxContext *ctx, int32_t enabled);
int32_t fx_read_package(FxContext *ctx, int32_t agent, double yaw_rate0, FxPackage *pkg, double *block /*[FX_PKG_ROWS][S] or NULL*/);
int32_t fx_plan_and_package(FxContext *ctx, const FxStateUpdate *upd, double yaw_rate0, FxResult *res, FxPackage *pkg, double *block);
/* the plan steps of a batch of agents in ONE call (agent_batch.py:140-189 runs one planner after the other): fx_update_state for
 * every agent whose upd[a] is not NULL (upd itself may be NULL), one evaluation with the winner package on, the results, every
 * agent's package.  n_agents = the uploaded batch's; yaw_rate0 [n_agents] (NULL: zeros); blocks [n_agents] pointers to
 * [FX_PKG_ROWS][S_a] doubles each (NULL, or NULL entries: no block).  A refused update leaves the context as the earlier
 * agents' updates left it and nothing is evaluated. */
int32_t fx_plan_batch_packaged(FxContext *ctx, int32_t n_agents, const FxStateUpdate *const *upd, const double *yaw_rate0, FxResult *res,
                               FxPackage *pkg, double *const *blocks);
/* the same in two halves, so that a host with several contexts keeps one evaluating while it prepares the next one's inputs and
 * consumes the previous one's results: _begin = the state updates + the evaluation's launches (returns without waiting),
 * _end = the wait, the results and the packages of THAT evaluation. */
int32_t fx_plan_batch_begin(FxContext *ctx, int32_t n_agents, const FxStateUpdate *const *upd);
int32_t fx_plan_batch_end(FxContext *ctx, int32_t n_agents, const double *yaw_rate0, FxResult *res, FxPackage *pkg, double *const *blocks);

/* ---- host geometry of the callers either side of the path (plain C, no device) ----
 *      fx_cs_to_curvilinear: (s, d) of a Cartesian point along a reference polyline with per-vertex normals (the projection
 *      behind planner.py:574-578); ref_xy / normals [M][2], ref_pos [M]; FX_ERR_INVALID_ARGUMENT outside the projection domain.
 *      fx_build_obstacle_hulls_batch: fx_build_obstacle_hulls for K obstacles stored with stride P. */
int32_t fx_cs_to_curvilinear(int32_t M, const double *ref_xy, const double *normals, const double *ref_pos, double x, double y,
                             double *sd /*[2]*/);
/* the same with the projection variant spelled out: pseudo_normal != 0 inverts the FX_MODE_PROJ_PSEUDO_NORMAL map */
int32_t fx_cs_to_curvilinear_ex(int32_t M, const double *ref_xy, const double *normals, const double *ref_pos, double x, double y,
                                int32_t pseudo_normal, double *sd);
/* n 2x2 matrices (row-major, 4 doubles each) inverted with the arithmetic of np.linalg.inv, bit for bit (collision_probability.py:281) */
int32_t fx_invert_cov2(int32_t n, const double *m, double *out);
/* K predicted obstacles -> the obstacle arrays of FxProblem / FxStateUpdate in one call (covariance inverses, hulls, padding
 * to the stride P); pos[k] [n[k]][2], cov[k] [n[k]][4], yaw[k] [n[k]] or NULL (no hulls for that obstacle). */
int32_t fx_pack_predictions(int32_t K, int32_t P, int32_t n_samples, const int32_t *n, const double *const *pos, const double *const *cov,
                            const double *const *yaw, const double *length, const double *width, double *pos_out, double *cov_inv_out,
                            int32_t *npred, double *hull, int32_t *nhull);
int32_t fx_build_obstacle_hulls_batch(int32_t K, int32_t P, const int32_t *n_use, const double *pos, const double *yaw,
                                      const double *length, const double *width, double *hull, int32_t *n_hull);

/* ---- read-back (TrajectorySample views are materialised lazily from the SoA bundle;
 *      reactive_planner_cpp.py:353 get_sorted_trajectories, trajectories.py:337-477) ---- */
int32_t fx_read_costs(FxContext *ctx, double *cost /*[C]*/, uint32_t *flags /*[C]*/);
int32_t fx_read_costmap(FxContext *ctx, double *raw /*[n_cost][C]*/);
int32_t fx_read_coeffs(FxContext *ctx, int64_t index, double *lon6, double *lat6, int32_t *traj_len);
int32_t fx_read_sample(FxContext *ctx, int64_t index, double *planes /*[FX_NUM_PLANES][S]*/);
int32_t fx_read_plane(FxContext *ctx, int32_t plane, double *out /*[S][C]*/);
/* k best selectable, collision-free candidates in (cost, index) order; returns count in *n_out.
 * Used for the host-side road-boundary walk (planner.py:362-390) and the multi-GPU exchange. */
int32_t fx_read_topk(FxContext *ctx, int32_t k, double *cost, int64_t *index, int32_t *n_out);
/* same, but left in device memory (caller-provided device pointers, e.g. torch tensors) so the
 * survivors can go straight into an RCCL all-gather; enqueued on the context stream. */
int32_t fx_topk_to_device(FxContext *ctx, int32_t k, void *d_cost /*f64[k]*/, void *d_index /*i64[k]*/);

/* ---- host-side helpers (pure CPU, no context) ---- */
/* OBB hulls of consecutive predicted boxes (collision_check.py:170-186 create_tvobstacle +
 * trajectory_preprocess_obb_sum; normative definition in DESIGN.md). pos[P][2], yaw[P]. */
int32_t fx_build_obstacle_hulls(int32_t n_pred, const double *pos, const double *yaw,
                                double length, double width, double *hull /*[n_pred-1][6]*/,
                                int32_t *n_hull);

/* ---- multi-agent batching: several agents evaluated by ONE launch (grid.y = agent).  The reference steps
 *      agents sequentially inside AgentBatch processes (cr_scenario_handler/simulation/agent_batch.py:186-189,
 *      simulation.py:449-470); agents are independent given the frozen predictions, so the batch is a pure
 *      concatenation.  max_candidates_total bounds the sum over agents.  res[n_agents]. ---- */
int32_t fx_create_batch(FxContext **out, int32_t device, int32_t max_agents, int64_t max_candidates_total,
                        int32_t max_steps, int32_t max_ref_knots, int32_t max_obstacles, int32_t max_pred_steps);
int32_t fx_upload_batch(FxContext *ctx, int32_t n_agents, const FxProblem *probs);
int32_t fx_finish_batch(FxContext *ctx, FxResult *res);
int32_t fx_read_costs_agent(FxContext *ctx, int32_t agent, double *cost, uint32_t *flags);
int32_t fx_read_costmap_agent(FxContext *ctx, int32_t agent, double *raw);
int32_t fx_read_coeffs_agent(FxContext *ctx, int32_t agent, int64_t index, double *lon6, double *lat6, int32_t *traj_len);
/* PolynomialTrajectory.delta_tau of one candidate's LATERAL polynomial as the device used it (polynomial_trajectory.py:17-60):
 * the sampled t, or -- LOW_VEL_MODE -- the arc length s_lon_goal = s(t) - s(0) of its longitudinal polynomial, t when that is
 * not positive (reactive_planner.py:161-171; stop-point bundle :650-659) */
int32_t fx_read_lat_tau_agent(FxContext *ctx, int32_t agent, int64_t index, double *tau_lat);
int32_t fx_read_sample_agent(FxContext *ctx, int32_t agent, int64_t index, double *planes);
int32_t fx_read_plane_agent(FxContext *ctx, int32_t agent, int32_t plane, double *out);
/* Everything a trajectory object of ONE candidate exposes (frenetix TrajectorySample: cartesian / curvilinear arrays,
 * coefficients, costMap, cost, feasibility -- reactive_planner_cpp.py:355-357,456,470-482) in one call with one stream
 * synchronisation: planes [FX_NUM_PLANES][S], coeffs13 = lon[6] | lat[6] | tau_lat (fx_read_lat_tau_agent), traj_len, raw partial costs [n_cost], total
 * cost and flag word.  Any output pointer may be NULL.  What the planner reads back for the chosen trajectory. */
int32_t fx_read_candidate_agent(FxContext *ctx, int32_t agent, int64_t index, double *planes, double *coeffs13,
                                int32_t *traj_len, double *raw_costs, double *cost, uint32_t *flags);
int32_t fx_read_topk_batch(FxContext *ctx, int32_t k, double *cost /*[n_agents][k]*/, int64_t *index /*[n_agents][k]*/);

/* ---- road boundary (replaces create_road_boundary_obstacle + trajectories_collision_static_obstacles,
 *      planner.py:362-381,550-565; commonroad-drivability-checker, not in the reference tree) ----
 * fx_build_boundary_bins: host-side geometry, no GPU.  Splits the n_seg boundary segments seg[n_seg][4] =
 * (ax, ay, bx, by) into pieces no longer than max_len, and lists for every reference knot k the pieces whose
 * midpoint lies within reach + half length of knot k.  piece_out[piece_cap][4], bin_out[M + 1],
 * item_out[item_cap]; *n_piece / *n_item receive the counts (FX_ERR_CAPACITY when a buffer is too small; the
 * counts are still set so the caller can re-allocate).  reach must cover the ego footprint around its foot point:
 * longest reference segment + largest |d| + wb_rear_axle + half diagonal of the vehicle. */
int32_t fx_build_boundary_bins(int32_t M, const double *ref_x, const double *ref_y, int32_t n_seg, const double *seg,
                               double max_len, double reach, int32_t piece_cap, double *piece_out, int32_t *n_piece,
                               int32_t *bin_out, int32_t item_cap, int32_t *item_out, int32_t *n_item);
/* first step at which each candidate's footprint meets the boundary, -1 if never (FX_MODE_ROAD_BOUNDARY);
 * boundary_harm = logistic regression of cartesian.v at that step (planner.py:369-375) is left to the caller */
int32_t fx_read_boundary_steps(FxContext *ctx, int32_t *steps);
int32_t fx_read_boundary_steps_agent(FxContext *ctx, int32_t agent, int32_t *steps);

/* device pointers of one agent's outputs for callers that keep working on the GPU:
 * cost f64[ld], flags u32[ld], planes f64[14][S][ld] (NULL without FX_MODE_WRITE_BUNDLE) */
int32_t fx_device_views(FxContext *ctx, int32_t agent, void **cost, void **flags, void **planes, int64_t *ld);

/* ---- measurement hooks (bench.py): device bytes owned; HIP-event time of the last step's device work
 *      (evaluation + selection kernels) and of the evaluation kernel alone ---- */
int64_t fx_device_bytes(const FxContext *ctx);
double fx_last_kernel_ms(const FxContext *ctx);
double fx_last_eval_kernel_ms(const FxContext *ctx);
/* selection fused into the evaluation kernel (default on): the evaluation kernel's last workgroup of an agent reduces the
 * partial arg-mins and publishes the result, so a plan step is a single launch.  With FX_MODE_COLLISION it also counts the
 * colliding candidates in front of the winner (planner.py:336-357) -- for agents of at most 8 192 candidates whose obstacle
 * stage runs inside the evaluation kernel (planner-sized steps; larger ones and steps whose obstacle stage is its own kernel keep
 * fx_select_kernel) -- and with fx_set_package it gathers the winner package: ReactivePlanner.plan() at the reference's operating
 * point (planning.yaml:34-35, 630 / 800 candidates) is ONE launch.  0 = always run the separate selection kernel (same results;
 * used by the parity tests), 1 = automatic, 2 = in-kernel whatever the candidate count (tests).  Takes effect at the next upload. */
int32_t fx_set_fused_selection(FxContext *ctx, int32_t enabled);
/* The whole plan step in ONE launch (csrc/fx_step_kernel.h; ABI 9), OPT-IN: steps whose obstacle stage runs as its own kernel
 * behind the walk (200 ... 3 072 waves with a materialised bundle: BASELINE config 3, config 4's batch) can run walk | grid barrier |
 * obstacle items | grid barrier | sliced selection (+ winner package) as phases of one kernel.  Same results (bit for bit at three
 * steps per item).  Measured slower on the MI355X -- config 3: 94 us against 86 us -- because the obstacle phase then runs with the
 * walk's register allocation (3 072 waves for ~5 000 work items: every wave runs two items' latency chains back to back), so it is off unless asked
 * for.  Needs every workgroup of the launch resident at once: the library sizes the launch by the occupancy query and keeps the three
 * launches where the walk alone would not fit.  The query assumes the device to itself: two such launches at a time (two contexts, two
 * processes) can hold each other's slots until the in-kernel barrier gives up after 2 s and the step ends in FX_ERR_TIMEOUT -- use it
 * from ONE context per device.  mode 0 / 1 = off (three launches), 2 = on where applicable (FX_STEP_KERNEL=1 in
 * the environment does the same for every context); steps_per_item 0 = automatic, or 3 / 5 / 8 steps of the horizon per obstacle
 * work item.  Takes effect at the next upload.  fx_step_info_ex [15] bit 16 reports that the last step ran this way. */
int32_t fx_set_step_kernel(FxContext *ctx, int32_t mode, int32_t steps_per_item);
/* how the last evaluation was launched: grid kernel, lanes per candidate, waves per SIMD, workgroup size, wave split, fused
 * selection, workgroups per agent, agents, winner package, dynamic LDS bytes */
int32_t fx_step_info(const FxContext *ctx, int64_t *out10);
/* the same ten values, then: [10] obstacle stage ran as its own kernel, [11] its steps per work item, [12] work items (waves) per
 * agent (max), [13] dynamic LDS bytes, [14] waves per workgroup when the chunks of a tile share one workgroup (0: one wave per
 * (tile, chunk) item), [15] what the agent's last workgroup did beyond the arg-min: bit 0 counted the collisions in front of the
 * winner, bit 1 gathered the winner package (0 with a separate selection kernel); bits 8-9 how the latest inputs (upload or state
 * update) reached the device: 1 DMA copy, 2 staging kernel reading the pinned block, 3 written by the host straight into device
 * memory -- where the device memory is mapped into the process (large BAR; probed at fx_create without risking a fault) a state
 * update needs no staging launch: posted writes, ordered in front of the evaluation launch.  FX_STAGE=kernel|dma|bar forces a path */
int32_t fx_step_info_ex(const FxContext *ctx, int64_t *out16);
/* HIP-event time of the obstacle kernel of the latest timed step / of the most recent <= max_n timed steps (FX_TIMING_KERNEL;
 * 0 where the stage ran fused into the walk) */
double fx_last_obstacle_kernel_ms(const FxContext *ctx);
int32_t fx_read_obstacle_kernel_times(FxContext *ctx, int32_t max_n, double *obst_ms, int32_t *n_out);
/* per-step HIP-event timing (default FX_TIMING_OFF: fx_finish only polls the result block the kernel publishes
 * into pinned host memory).  FX_TIMING_STREAM -- stream events around the kernels (the evaluation figure includes
 * the dispatch gap in front of the kernel); FX_TIMING_KERNEL -- start/stop events attached to the evaluation
 * kernel itself (hipExtLaunchKernel): the figure a kernel trace reports, at a few microseconds more host time per
 * timed launch.  Events live in a ring of 256 steps and are read on request only: fx_last_*_ms (latest timed
 * step) and fx_read_kernel_times (the most recent <= max_n timed steps, oldest first; n_out = how many) wait for
 * the events they read, fx_finish never does.  fx_set_timing_interval: time every n-th step only. */
enum { FX_TIMING_OFF = 0, FX_TIMING_STREAM = 1, FX_TIMING_KERNEL = 2 };
int32_t fx_set_timing(FxContext *ctx, int32_t mode);
int32_t fx_set_timing_interval(FxContext *ctx, int32_t every);
int32_t fx_read_kernel_times(FxContext *ctx, int32_t max_n, double *eval_ms, double *step_ms, int32_t *n_out);
/* device self-test of the kernel's elementary functions (atan, sin, cos) on n host values */
int32_t fx_math_selftest(int32_t n, const double *x, double *atan_out, double *sin_out, double *cos_out);

#ifdef __cplusplus
}
#endif
#endif /* FXPLAN_H */
