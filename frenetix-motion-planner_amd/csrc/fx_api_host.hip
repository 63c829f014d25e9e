// fx_api_host.hip -- host geometry of the callers either side of the path (no GPU involved) and the read-back of a step's
// per-candidate outputs (header: include/fxplan.h; context: fx_context.h).
#include "fx_context.h"

extern "C" {

// ---- host geometry of the callers either side of the path ----
// (s, d) of a Cartesian point along the reference polyline (planner.py:574-578 convert_to_curvilinear_coords): on every
// segment k the foot point P_k + lam b and the interpolated normal n_k + lam dn are collinear with the point where
// cross(a + lam b, n_k + lam dn) = 0, a quadratic in lam; of all roots in [0, 1] the one with the smallest |d| wins.
int32_t fx_cs_to_curvilinear(int32_t M, const double *ref_xy, const double *normals, const double *ref_pos, double x, double y, double *sd) {
    return fx_cs_to_curvilinear_ex(M, ref_xy, normals, ref_pos, x, y, 0, sd);
}
int32_t fx_cs_to_curvilinear_ex(int32_t M, const double *ref_xy, const double *normals, const double *ref_pos, double x, double y,
                                int32_t pseudo_normal, double *sd) {
    if (M < 2 || !ref_xy || !normals || !ref_pos || !sd) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_cs_to_curvilinear: bad argument");
    bool have = false;
    double best_s = 0, best_d = 0;
    for (int k = 0; k + 1 < M; k++) {
        const double ax = ref_xy[2 * k] - x, ay = ref_xy[2 * k + 1] - y;
        const double bx = ref_xy[2 * k + 2] - ref_xy[2 * k], by = ref_xy[2 * k + 3] - ref_xy[2 * k + 1];
        const double n0x = normals[2 * k], n0y = normals[2 * k + 1];
        const double dnx = normals[2 * k + 2] - n0x, dny = normals[2 * k + 3] - n0y;
        const double c2 = bx * dny - by * dnx;
        const double c1 = ax * dny - ay * dnx + bx * n0y - by * n0x;
        const double c0 = ax * n0y - ay * n0x;
        double roots[2];
        int nr = 0;
        if (std::fabs(c2) < 1e-14) {
            if (!(std::fabs(c1) < 1e-300)) roots[nr++] = -c0 / c1;
        } else {
            const double disc = c1 * c1 - 4 * c2 * c0;
            if (disc < 0) continue;
            const double sq = std::sqrt(disc);
            roots[nr++] = (-c1 + sq) / (2 * c2);
            roots[nr++] = (-c1 - sq) / (2 * c2);
        }
        for (int r = 0; r < nr; r++) {
            double lam = roots[r];
            if (!(lam >= -1e-12 && lam <= 1 + 1e-12)) continue;
            lam = std::fmin(std::fmax(lam, 0.0), 1.0);
            const double fx = ref_xy[2 * k] + lam * bx, fy = ref_xy[2 * k + 1] + lam * by;
            double nx = n0x + lam * dnx, ny = n0y + lam * dny;
            // (x, y) = foot + d n / |n|  (or foot + d n: the pseudo-distance variant)  =>  d = (p - foot) . n / |n|  (/ |n|^2)
            const double nn = pseudo_normal ? nx * nx + ny * ny : std::sqrt(nx * nx + ny * ny);
            nx = nx / nn; ny = ny / nn;
            const double dd = (x - fx) * nx + (y - fy) * ny;
            if (!have || std::fabs(dd) < std::fabs(best_d)) {
                have = true;
                best_d = dd;
                best_s = ref_pos[k] + lam * (ref_pos[k + 1] - ref_pos[k]);
            }
        }
    }
    if (!have) return set_err(FX_ERR_INVALID_ARGUMENT, "point outside projection domain");
    sd[0] = best_s; sd[1] = best_d;
    return FX_OK;
}

// Inverses of n 2x2 matrices with the arithmetic of np.linalg.inv (LAPACK gesv on the identity as OpenBLAS executes it:
// partial pivoting, the multiplier and both divisions through reciprocals, one fused multiply-add in the back substitution)
// -- bit-identical to NumPy's result (collision_probability.py:281 inverts the prediction covariances with it; the CPU
// suite compares the two on random matrices).  Returns FX_ERR_INVALID_ARGUMENT for a singular matrix (NumPy: LinAlgError).
int32_t fx_invert_cov2(int32_t n, const double *m, double *out) {
    if (n < 0 || (n > 0 && (!m || !out))) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_invert_cov2: bad argument");
    for (int i = 0; i < n; i++) {
        const double a = m[4 * i], b = m[4 * i + 1], c = m[4 * i + 2], d = m[4 * i + 3];
        const bool sw = std::fabs(c) > std::fabs(a);
        const double p0a = sw ? c : a, p0b = sw ? d : b, p1a = sw ? a : c, p1b = sw ? b : d;
        if (p0a == 0.0) return set_err(FX_ERR_INVALID_ARGUMENT, "singular matrix (%d)", i);
        const double rp = 1.0 / p0a;
        const double l = p1a * rp;
        const double u11 = p1b - l * p0b;
        if (u11 == 0.0) return set_err(FX_ERR_INVALID_ARGUMENT, "singular matrix (%d)", i);
        const double ru = 1.0 / u11;
        for (int col = 0; col < 2; col++) {
            const double r0 = sw ? (col == 1) : (col == 0), r1 = sw ? (col == 0) : (col == 1);
            const double x1 = (r1 - l * r0) * ru;
            out[4 * i + col] = std::fma(-p0b, x1, r0) * rp;
            out[4 * i + 2 + col] = x1;
        }
    }
    return FX_OK;
}

// Packing of K predicted obstacles (prediction_helpers.py:209-261 dict entries) into the arrays FxProblem / FxStateUpdate take,
// in one call: obstacle k has n[k] predictions at pos[k] ([n][2]), cov[k] ([n][4]) and -- when yaw[k] is not NULL -- headings
// yaw[k] ([n]) with the box length[k] x width[k].  Outputs with stride P (zero-filled here): pos_out [K][P][2], cov_inv_out
// [K][P][4] (fx_invert_cov2), npred [K] = n[k] (the real length decides which ego steps see the obstacle,
// collision_probability.py:287), hull [K][P-1][6], nhull [K] (hulls over the first min(n_samples, n[k], P) boxes,
// collision_check.py:150).
int32_t fx_pack_predictions(int32_t K, int32_t P, int32_t n_samples, const int32_t *n, const double *const *pos, const double *const *cov,
                            const double *const *yaw, const double *length, const double *width, double *pos_out, double *cov_inv_out,
                            int32_t *npred, double *hull, int32_t *nhull) {
    if (K < 0 || P < 2 || (K > 0 && (!n || !pos || !cov || !yaw || !length || !width || !pos_out || !cov_inv_out || !npred || !hull || !nhull)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_pack_predictions: bad argument");
    memset(pos_out, 0, sizeof(double) * 2 * (size_t)K * P);
    memset(cov_inv_out, 0, sizeof(double) * 4 * (size_t)K * P);
    memset(hull, 0, sizeof(double) * 6 * (size_t)K * (P - 1));
    for (int k = 0; k < K; k++) {
        npred[k] = n[k];
        nhull[k] = 0;
        const int m = std::min(n[k], P);
        if (m <= 0) continue;
        if (!pos[k] || !cov[k]) return set_err(FX_ERR_INVALID_ARGUMENT, "obstacle %d: NULL arrays", k);
        memcpy(pos_out + (size_t)2 * P * k, pos[k], sizeof(double) * 2 * m);
        int rc = fx_invert_cov2(m, cov[k], cov_inv_out + (size_t)4 * P * k);
        if (rc) return rc;
        if (yaw[k]) {
            rc = fx_build_obstacle_hulls(std::min(n_samples, m), pos_out + (size_t)2 * P * k, yaw[k], length[k], width[k],
                                         hull + (size_t)6 * (P - 1) * k, nhull + k);
            if (rc) return rc;
        }
    }
    return FX_OK;
}

// fx_build_obstacle_hulls for K obstacles in one call: pos [K][P][2], yaw [K][P], n_use [K] predictions that count,
// length / width [K]; hull [K][P-1][6], n_hull [K].
int32_t fx_build_obstacle_hulls_batch(int32_t K, int32_t P, const int32_t *n_use, const double *pos, const double *yaw,
                                      const double *length, const double *width, double *hull, int32_t *n_hull) {
    if (K < 0 || P < 2 || (K > 0 && (!n_use || !pos || !yaw || !length || !width || !hull || !n_hull)))
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_build_obstacle_hulls_batch: bad argument");
    for (int k = 0; k < K; k++) {
        if (n_use[k] > P) return set_err(FX_ERR_INVALID_ARGUMENT, "obstacle %d: %d predictions, stride %d", k, n_use[k], P);
        int rc = fx_build_obstacle_hulls(n_use[k], pos + (size_t)2 * P * k, yaw + (size_t)P * k, length[k], width[k],
                                         hull + (size_t)6 * (P - 1) * k, n_hull + k);
        if (rc) return rc;
    }
    return FX_OK;
}

// ---- read-back ----

int32_t fx_read_costs_agent(FxContext *c, int32_t agent, double *cost, uint32_t *flags) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    if (cost) HIP_TRY(hipMemcpy(cost, c->d_cost + s.cand_off, sizeof(double) * s.C, hipMemcpyDeviceToHost));
    if (flags) HIP_TRY(hipMemcpy(flags, c->d_flags + s.cand_off, sizeof(uint32_t) * s.C, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_costs(FxContext *c, double *cost, uint32_t *flags) { return fx_read_costs_agent(c, 0, cost, flags); }

int32_t fx_read_costmap_agent(FxContext *c, int32_t agent, double *raw) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_COSTMAP)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_COSTMAP");
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    HIP_TRY(hipMemcpy2D(raw, sizeof(double) * s.C, c->d_costmap + (size_t)FX_NUM_COSTS * s.cand_off, sizeof(double) * s.ld,
                        sizeof(double) * s.C, s.n_cost, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_costmap(FxContext *c, double *raw) { return fx_read_costmap_agent(c, 0, raw); }

static int32_t read_coeff_rows(FxContext *c, int32_t agent, int64_t index, double *lon6, double *lat6, double *tau_lat, int32_t *traj_len) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_BUNDLE)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (index < 0 || index >= s.C) return set_err(FX_ERR_INVALID_ARGUMENT, "candidate %lld out of range", (long long)index);
    double tmp[FX_COEFF_ROWS];
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    HIP_TRY(hipMemcpy2D(tmp, sizeof(double), c->d_coeffs + (size_t)FX_COEFF_ROWS * s.cand_off + index, sizeof(double) * s.ld,
                        sizeof(double), FX_COEFF_ROWS, hipMemcpyDeviceToHost));
    if (lon6) memcpy(lon6, tmp, 6 * sizeof(double));
    if (lat6) memcpy(lat6, tmp + 6, 6 * sizeof(double));
    if (tau_lat) *tau_lat = tmp[12];
    if (traj_len) HIP_TRY(hipMemcpy(traj_len, c->d_trajlen + s.cand_off + index, sizeof(int32_t), hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_coeffs_agent(FxContext *c, int32_t agent, int64_t index, double *lon6, double *lat6, int32_t *traj_len) {
    return read_coeff_rows(c, agent, index, lon6, lat6, nullptr, traj_len);
}
int32_t fx_read_lat_tau_agent(FxContext *c, int32_t agent, int64_t index, double *tau_lat) {
    if (!tau_lat) return set_err(FX_ERR_INVALID_ARGUMENT, "tau_lat is NULL");
    return read_coeff_rows(c, agent, index, nullptr, nullptr, tau_lat, nullptr);
}
int32_t fx_read_boundary_steps_agent(FxContext *c, int32_t agent, int32_t *steps) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    if (!steps) return set_err(FX_ERR_INVALID_ARGUMENT, "steps is NULL");
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_ROAD_BOUNDARY)) return set_err(FX_ERR_NOT_READY, "the step ran without FX_MODE_ROAD_BOUNDARY");
    HIP_TRY(hipMemcpyAsync(steps, c->d_bstep + s.cand_off, sizeof(int32_t) * s.C, hipMemcpyDeviceToHost, c->stream));
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    return FX_OK;
}
int32_t fx_read_boundary_steps(FxContext *c, int32_t *steps) { return fx_read_boundary_steps_agent(c, 0, steps); }

int32_t fx_read_coeffs(FxContext *c, int64_t index, double *lon6, double *lat6, int32_t *traj_len) {
    return fx_read_coeffs_agent(c, 0, index, lon6, lat6, traj_len);
}

int32_t fx_read_sample_agent(FxContext *c, int32_t agent, int64_t index, double *planes) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_BUNDLE)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (index < 0 || index >= s.C) return set_err(FX_ERR_INVALID_ARGUMENT, "candidate %lld out of range", (long long)index);
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    // one strided gather: 14*S elements, pitch = ld doubles
    HIP_TRY(hipMemcpy2D(planes, sizeof(double), c->h_probs[agent].planes + index, sizeof(double) * s.ld, sizeof(double),
                        (size_t)FX_NUM_PLANES * s.S, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_sample(FxContext *c, int64_t index, double *planes) { return fx_read_sample_agent(c, 0, index, planes); }

int32_t fx_read_candidate_agent(FxContext *c, int32_t agent, int64_t index, double *planes, double *coeffs13, int32_t *traj_len,
                                double *raw_costs, double *cost, uint32_t *flags) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (index < 0 || index >= s.C) return set_err(FX_ERR_INVALID_ARGUMENT, "candidate %lld out of range", (long long)index);
    const bool bundle = (s.mode & FX_MODE_WRITE_BUNDLE) != 0, cmap = (s.mode & FX_MODE_WRITE_COSTMAP) != 0;
    if ((planes || coeffs13 || traj_len) && !bundle) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (raw_costs && !cmap) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_COSTMAP");
    // all pieces go to one pinned block with asynchronous copies; ONE synchronisation
    double *hp = c->h_cand;
    const size_t n_pl = (size_t)FX_NUM_PLANES * s.S;
    double *h_co = hp + n_pl, *h_rc = h_co + FX_COEFF_ROWS, *h_c = h_rc + FX_NUM_COSTS;
    int32_t *h_tl = reinterpret_cast<int32_t *>(h_c + 1);
    uint32_t *h_fl = reinterpret_cast<uint32_t *>(h_c + 2);
    if (n_pl + FX_COEFF_ROWS + FX_NUM_COSTS + 4 > c->h_cand_doubles) return set_err(FX_ERR_CAPACITY, "candidate staging block too small");
    if (planes)
        HIP_TRY(hipMemcpy2DAsync(hp, sizeof(double), c->h_probs[agent].planes + index, sizeof(double) * s.ld, sizeof(double), n_pl,
                                 hipMemcpyDeviceToHost, c->stream));
    if (coeffs13)
        HIP_TRY(hipMemcpy2DAsync(h_co, sizeof(double), c->d_coeffs + (size_t)FX_COEFF_ROWS * s.cand_off + index, sizeof(double) * s.ld,
                                 sizeof(double), FX_COEFF_ROWS, hipMemcpyDeviceToHost, c->stream));
    if (traj_len) HIP_TRY(hipMemcpyAsync(h_tl, c->d_trajlen + s.cand_off + index, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (raw_costs && s.n_cost > 0)
        HIP_TRY(hipMemcpy2DAsync(h_rc, sizeof(double), c->d_costmap + (size_t)FX_NUM_COSTS * s.cand_off + index, sizeof(double) * s.ld,
                                 sizeof(double), s.n_cost, hipMemcpyDeviceToHost, c->stream));
    if (cost) HIP_TRY(hipMemcpyAsync(h_c, c->d_cost + s.cand_off + index, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (flags) HIP_TRY(hipMemcpyAsync(h_fl, c->d_flags + s.cand_off + index, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    if (planes) memcpy(planes, hp, sizeof(double) * n_pl);
    if (coeffs13) memcpy(coeffs13, h_co, sizeof(double) * FX_COEFF_ROWS);
    if (traj_len) *traj_len = *h_tl;
    if (raw_costs) memcpy(raw_costs, h_rc, sizeof(double) * s.n_cost);
    if (cost) *cost = *h_c;
    if (flags) *flags = *h_fl;
    return FX_OK;
}

int32_t fx_read_plane_agent(FxContext *c, int32_t agent, int32_t plane, double *out) {
    int rc = check_agent(c, agent);
    if (rc) return rc;
    const FxAgentSlot &s = c->slots[agent];
    if (!(s.mode & FX_MODE_WRITE_BUNDLE)) return set_err(FX_ERR_NOT_READY, "plan step ran without FX_MODE_WRITE_BUNDLE");
    if (plane < 0 || plane >= FX_NUM_PLANES) return set_err(FX_ERR_INVALID_ARGUMENT, "plane %d out of range", plane);
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    HIP_TRY(hipMemcpy2D(out, sizeof(double) * s.C, c->h_probs[agent].planes + (size_t)plane * s.S * s.ld, sizeof(double) * s.ld,
                        sizeof(double) * s.C, s.S, hipMemcpyDeviceToHost));
    return FX_OK;
}
int32_t fx_read_plane(FxContext *c, int32_t plane, double *out) { return fx_read_plane_agent(c, 0, plane, out); }

int32_t fx_topk_to_device(FxContext *c, int32_t k, void *d_cost, void *d_index) {
    if (!c) return set_err(FX_ERR_INVALID_ARGUMENT, "context is NULL");
    if (!c->evaluated) return set_err(FX_ERR_NOT_READY, "no evaluated plan step");
    if (k < 1 || k > 64) return set_err(FX_ERR_INVALID_ARGUMENT, "k=%d outside [1,64]", k);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(fx_launch_topk(c->d_probs, c->n_agents, max_candidates_of(c), k, c->d_topk_scr_cost, c->d_topk_scr_idx, reinterpret_cast<double *>(d_cost),
                           reinterpret_cast<long long *>(d_index), c->stream));
    c->in_flight = true; c->tail_work = true;
    return FX_OK;
}

int32_t fx_read_topk_batch(FxContext *c, int32_t k, double *cost, int64_t *index) {
    int rc = fx_topk_to_device(c, k, c->d_topk_cost, c->d_topk_idx);
    if (rc) return rc;
    const size_t n = (size_t)k * c->n_agents;
    HIP_TRY(hipMemcpyAsync(c->h_topk_cost, c->d_topk_cost, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_topk_idx, c->d_topk_idx, sizeof(long long) * n, hipMemcpyDeviceToHost, c->stream));
    { HIP_TRY(hipStreamSynchronize(c->stream)); c->tail_work = c->user_stream; }
    memcpy(cost, c->h_topk_cost, sizeof(double) * n);
    for (size_t i = 0; i < n; i++) index[i] = (int64_t)c->h_topk_idx[i];
    return FX_OK;
}

int32_t fx_read_topk(FxContext *c, int32_t k, double *cost, int64_t *index, int32_t *n_out) {
    if (c && c->n_agents != 1) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_read_topk is single-agent; use fx_read_topk_batch");
    int rc = fx_read_topk_batch(c, k, cost, index);
    if (rc) return rc;
    int n = 0;
    while (n < k && index[n] >= 0) n++;
    if (n_out) *n_out = n;
    return FX_OK;
}

int32_t fx_build_obstacle_hulls(int32_t n_pred, const double *pos, const double *yaw, double length, double width,
                                double *hull, int32_t *n_hull) {
    if (!n_hull || (n_pred > 0 && (!pos || !yaw))) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_build_obstacle_hulls: NULL argument");
    // collision_check.py:165-168: an obstacle with <= 2 predicted steps is skipped
    if (n_pred <= 2) { *n_hull = 0; return FX_OK; }
    if (!hull) return set_err(FX_ERR_INVALID_ARGUMENT, "hull output is NULL");
    const double hl = length / 2, hw = width / 2;
    double u1x, u1y;   // box j + 1's heading is box j's of the next hull: one cos / sin per box
    ::sincos(yaw[0], &u1y, &u1x);
    for (int j = 0; j + 1 < n_pred; j++) {
        const double c0x = pos[2 * j], c0y = pos[2 * j + 1], c1x = pos[2 * j + 2], c1y = pos[2 * j + 3];
        const double u0x = u1x, u0y = u1y;
        ::sincos(yaw[j + 1], &u1y, &u1x);
        double mx = u0x + u1x, my = u0y + u1y;
        double mn = std::sqrt(mx * mx + my * my);
        double ex, ey;
        if (mn < 1e-12) { ex = u0x; ey = u0y; } else { ex = mx / mn; ey = my / mn; }
        const double fx = -ey, fy = ex;
        double lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
        for (int b = 0; b < 2; b++) {
            const double cx = b ? c1x : c0x, cy = b ? c1y : c0y, ux = b ? u1x : u0x, uy = b ? u1y : u0y;
            const double p1 = cx * ex + cy * ey, p2 = cx * fx + cy * fy;
            const double r1 = hl * std::fabs(ux * ex + uy * ey) + hw * std::fabs(-uy * ex + ux * ey);
            const double r2 = hl * std::fabs(ux * fx + uy * fy) + hw * std::fabs(-uy * fx + ux * fy);
            if (b == 0) { lo1 = p1 - r1; hi1 = p1 + r1; lo2 = p2 - r2; hi2 = p2 + r2; }
            else {
                lo1 = std::fmin(lo1, p1 - r1); hi1 = std::fmax(hi1, p1 + r1);
                lo2 = std::fmin(lo2, p2 - r2); hi2 = std::fmax(hi2, p2 + r2);
            }
        }
        const double m1 = 0.5 * (lo1 + hi1), m2 = 0.5 * (lo2 + hi2);
        double *o = hull + 6 * j;
        o[0] = m1 * ex + m2 * fx;
        o[1] = m1 * ey + m2 * fy;
        o[2] = ex;
        o[3] = ey;
        o[4] = 0.5 * (hi1 - lo1);
        o[5] = 0.5 * (hi2 - lo2);
    }
    *n_hull = n_pred - 1;
    return FX_OK;
}

// self-test hook: atan / sin / cos of the device math kernels for n host values (synchronous)
int32_t fx_math_selftest(int32_t n, const double *x, double *atan_out, double *sin_out, double *cos_out) {
    if (n < 1 || !x || !atan_out || !sin_out || !cos_out) return set_err(FX_ERR_INVALID_ARGUMENT, "fx_math_selftest: bad argument");
    double *d = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(double) * 4 * n));
    HIP_TRY(hipMemcpy(d, x, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(fx_launch_math_test(n, d, d + n, d + 2 * n, d + 3 * n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(atan_out, d + n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sin_out, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cos_out, d + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return FX_OK;
}

// Road-boundary geometry prep (host only): split segments into pieces, bin them by reference knot.
int32_t fx_build_boundary_bins(int32_t M, const double *ref_x, const double *ref_y, int32_t n_seg, const double *seg,
                               double max_len, double reach, int32_t piece_cap, double *piece_out, int32_t *n_piece,
                               int32_t *bin_out, int32_t item_cap, int32_t *item_out, int32_t *n_item) {
    if (M < 1 || !ref_x || !ref_y || n_seg < 0 || (n_seg && !seg) || !(max_len > 0.0) || !(reach >= 0.0) || !n_piece || !n_item ||
        !bin_out)
        return set_err(FX_ERR_INVALID_ARGUMENT, "fx_build_boundary_bins: bad argument");
    int64_t np_ = 0;
    for (int i = 0; i < n_seg; i++) {
        const double *q = seg + 4 * (size_t)i;
        const double len = std::sqrt((q[2] - q[0]) * (q[2] - q[0]) + (q[3] - q[1]) * (q[3] - q[1]));
        const int n = std::max(1, (int)std::ceil(len / max_len));
        for (int k = 0; k < n; k++, np_++) {
            if (np_ >= piece_cap || !piece_out) continue;
            const double t0 = (double)k / n, t1 = (double)(k + 1) / n;
            const double ax = q[0] + t0 * (q[2] - q[0]), ay = q[1] + t0 * (q[3] - q[1]);
            const double bx = q[0] + t1 * (q[2] - q[0]), by = q[1] + t1 * (q[3] - q[1]);
            double *o = piece_out + 4 * (size_t)np_;
            o[0] = 0.5 * (ax + bx); o[1] = 0.5 * (ay + by); o[2] = 0.5 * (bx - ax); o[3] = 0.5 * (by - ay);
        }
    }
    *n_piece = (int32_t)np_;
    if (np_ > piece_cap || !piece_out) { *n_item = 0; return set_err(FX_ERR_CAPACITY, "boundary needs %lld pieces", (long long)np_); }
    int64_t ni = 0;
    bin_out[0] = 0;
    for (int k = 0; k < M; k++) {
        for (int64_t j = 0; j < np_; j++) {
            const double *o = piece_out + 4 * (size_t)j;
            const double dx = o[0] - ref_x[k], dy = o[1] - ref_y[k];
            if (std::sqrt(dx * dx + dy * dy) <= reach + std::sqrt(o[2] * o[2] + o[3] * o[3])) {
                if (ni < item_cap && item_out) item_out[ni] = (int32_t)j;
                ni++;
            }
        }
        bin_out[k + 1] = (int32_t)std::min<int64_t>(ni, INT32_MAX);
    }
    *n_item = (int32_t)std::min<int64_t>(ni, INT32_MAX);
    if (ni > item_cap || !item_out) return set_err(FX_ERR_CAPACITY, "boundary bins need %lld items", (long long)ni);
    return FX_OK;
}

}  // extern "C"
