/* fx_host_ext.c -- CPython extension `_fxhost`: the per-plan-step host work of the planner that is dict walking.
 *
 * ReactivePlanner.plan() hands the planner a predictions dict every step (planner.py:172-217 update_externals,
 * prediction_helpers.py:209-261: {obstacle id: {'pos_list' [n,2], 'cov_list' [n,2,2], 'orientation_list' [n], 'shape':
 * {'length', 'width'}}}).  Walking it from Python -- five obstacles, three arrays each, conversions, pointer extraction for
 * ctypes -- cost 16 of the 30 us the packing took; here the dict is walked in C through the buffer protocol and the arrays go
 * straight into libfxplan's fx_pack_predictions (covariance inverses with np.linalg.inv's arithmetic, OBB-sum hulls, padding
 * to the stride P).  The extension holds no arithmetic of its own: it calls the library through the function pointer Python
 * hands over (ctypes address of fx_pack_predictions), so there is nothing to link.
 *
 * pack_predictions(fn_addr, predictions, n_samples, max_obstacles) -> (K, P, out: bytearray, counts: bytearray) or None
 *   None: something in the dict is not a C-contiguous float64 buffer of the expected shape -- the caller takes the general
 *   Python path (which converts).  out = pos [K][P][2] | cov_inv [K][P][4] | hull [K][P-1][6] doubles, counts = npred [K] |
 *   nhull [K] int32.  Raises numpy.linalg.LinAlgError-compatible ValueError("singular") when a covariance is singular.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <stdint.h>
#include <string.h>

#include "../../include/fxplan.h"

#define FXH_MAX_OBSTACLES 256

typedef int32_t (*fx_pack_fn)(int32_t K, int32_t P, int32_t n_samples, const int32_t *n, const double *const *pos, const double *const *cov,
                              const double *const *yaw, const double *length, const double *width, double *pos_out, double *cov_inv_out,
                              int32_t *npred, double *hull, int32_t *nhull);

static PyObject *s_pos, *s_cov, *s_yaw, *s_shape, *s_length, *s_width;

/* a C-contiguous float64 buffer with `want` doubles per leading element; returns the number of leading elements or -1 */
static Py_ssize_t f64_rows(PyObject *obj, Py_buffer *view, Py_ssize_t want) {
    if (PyObject_GetBuffer(obj, view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
        PyErr_Clear();
        return -1;
    }
    if (view->itemsize != 8 || !view->format || !(view->format[0] == 'd' || (view->format[0] == '=' && view->format[1] == 'd') ||
                                                   (view->format[0] == '<' && view->format[1] == 'd'))) {
        PyBuffer_Release(view);
        return -1;
    }
    const Py_ssize_t n = view->len / 8;
    if (want <= 0 || n % want != 0) {
        PyBuffer_Release(view);
        return -1;
    }
    return n / want;
}

static PyObject *pack_predictions(PyObject *self, PyObject *args) {
    unsigned long long fn_addr;
    PyObject *preds;
    int n_samples, max_obstacles;
    if (!PyArg_ParseTuple(args, "KOii", &fn_addr, &preds, &n_samples, &max_obstacles)) return NULL;
    if (!PyDict_Check(preds) || fn_addr == 0) Py_RETURN_NONE;
    const Py_ssize_t K = PyDict_Size(preds);
    if (K <= 0 || K > FXH_MAX_OBSTACLES || K > max_obstacles) Py_RETURN_NONE;   /* (the Python path words the errors) */

    Py_buffer views[3 * FXH_MAX_OBSTACLES];
    int n_views = 0;
    int32_t n[FXH_MAX_OBSTACLES];
    const double *pos[FXH_MAX_OBSTACLES], *cov[FXH_MAX_OBSTACLES], *yaw[FXH_MAX_OBSTACLES];
    double length[FXH_MAX_OBSTACLES], width[FXH_MAX_OBSTACLES];
    int ok = 1;
    Py_ssize_t it = 0, longest = 0;
    PyObject *key, *pr;
    int k = 0;
    while (ok && PyDict_Next(preds, &it, &key, &pr)) {   /* dict order, as get_inv_mahalanobis_dist iterates (collision_probability.py:276) */
        if (!PyDict_Check(pr)) { ok = 0; break; }
        PyObject *o_pos = PyDict_GetItemWithError(pr, s_pos), *o_cov = PyDict_GetItemWithError(pr, s_cov);
        PyObject *o_yaw = PyDict_GetItemWithError(pr, s_yaw), *o_shape = PyDict_GetItemWithError(pr, s_shape);
        if (!o_pos || !o_cov) { ok = 0; break; }
        const Py_ssize_t np_ = f64_rows(o_pos, &views[n_views], 2);
        if (np_ < 0) { ok = 0; break; }
        pos[k] = (const double *)views[n_views++].buf;
        const Py_ssize_t nc = f64_rows(o_cov, &views[n_views], 4);
        if (nc < 0) { ok = 0; break; }
        cov[k] = (const double *)views[n_views++].buf;
        if (nc != np_) { ok = 0; break; }
        yaw[k] = NULL; length[k] = 0.0; width[k] = 0.0;
        if (o_yaw && o_shape) {   /* hulls need orientations and a shape */
            if (!PyDict_Check(o_shape)) { ok = 0; break; }
            const Py_ssize_t ny = f64_rows(o_yaw, &views[n_views], 1);
            if (ny < 0) { ok = 0; break; }
            yaw[k] = (const double *)views[n_views++].buf;
            if (ny != np_) { ok = 0; break; }
            PyObject *ol = PyDict_GetItemWithError(o_shape, s_length), *ow = PyDict_GetItemWithError(o_shape, s_width);
            if (!ol || !ow) { ok = 0; break; }
            length[k] = PyFloat_AsDouble(ol); width[k] = PyFloat_AsDouble(ow);
            if (PyErr_Occurred()) { PyErr_Clear(); ok = 0; break; }
        }
        if (np_ > 0x7fffffff) { ok = 0; break; }
        n[k] = (int32_t)np_;
        if (np_ > longest) longest = np_;
        k++;
    }
    if (PyErr_Occurred()) PyErr_Clear();
    PyObject *result = NULL;
    if (ok && k == (int)K) {
        /* only the first n_samples predictions are ever read: a long-horizon predictor does not enlarge the tables */
        Py_ssize_t P = longest < n_samples ? longest : n_samples;
        if (P < 2) P = 2;
        const Py_ssize_t n_pos = 2 * K * P, n_cov = 4 * K * P, n_hull = 6 * K * (P - 1);
        PyObject *out = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)sizeof(double) * (n_pos + n_cov + n_hull));
        PyObject *cnt = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)sizeof(int32_t) * 2 * K);
        if (out && cnt) {
            double *o = (double *)PyByteArray_AS_STRING(out);
            int32_t *c = (int32_t *)PyByteArray_AS_STRING(cnt);
            const int32_t rc = ((fx_pack_fn)(uintptr_t)fn_addr)((int32_t)K, (int32_t)P, n_samples, n, pos, cov, yaw, length, width, o, o + n_pos,
                                                                c, o + n_pos + n_cov, c + K);
            if (rc == 0) result = Py_BuildValue("(nnOO)", K, P, out, cnt);
            else PyErr_SetString(PyExc_ValueError, "fx_pack_predictions failed (singular covariance or bad argument)");
        }
        Py_XDECREF(out);
        Py_XDECREF(cnt);
    }
    for (int v = 0; v < n_views; v++) PyBuffer_Release(&views[v]);
    if (result || PyErr_Occurred()) return result;
    Py_RETURN_NONE;
}

/* address of a C-contiguous buffer of the given item format ('d' or 'i'), NULL for None; -1 on anything else */
static int buf_addr_len(PyObject *obj, char fmt, Py_ssize_t itemsize, const void **out, Py_ssize_t *n_items) {
    *out = NULL;
    *n_items = 0;
    if (obj == Py_None) return 0;
    Py_buffer view;
    if (PyObject_GetBuffer(obj, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return -1;
    const char *f = view.format ? view.format : "B";
    if (*f == '=' || *f == '<' || *f == '@') f++;
    const int ok = view.itemsize == itemsize && (f[0] == fmt || (fmt == 'i' && f[0] == 'l' && itemsize == 4)) && f[1] == 0;
    *out = view.buf;
    *n_items = view.len / itemsize;
    PyBuffer_Release(&view);   /* (the caller keeps the arrays alive; the address stays valid) */
    if (!ok) {
        PyErr_SetString(PyExc_TypeError, "expected a C-contiguous float64 / int32 array");
        return -1;
    }
    return 0;
}

static PyObject *state_update(PyObject *self, PyObject *args) {
    unsigned long long addr;
    PyObject *x0_lon, *x0_lat, *t, *v, *d, *pos, *cov, *npred, *hull, *nhull;
    double orient, v_des;
    int low_vel;
    if (!PyArg_ParseTuple(args, "KOOddiOOOOOOOO", &addr, &x0_lon, &x0_lat, &orient, &v_des, &low_vel, &t, &v, &d, &pos, &cov, &npred,
                          &hull, &nhull))
        return NULL;
    if (addr == 0) {
        PyErr_SetString(PyExc_ValueError, "state_update: NULL struct address");
        return NULL;
    }
    FxStateUpdate u;
    memset(&u, 0, sizeof(u));
    const void *p;
    Py_ssize_t n, n_pos, n_cov, n_np, n_hull, n_nh;
#define FXH_PTR(field, obj, fmt, size, type, count)                 \
    if (buf_addr_len(obj, fmt, size, &p, &count) != 0) return NULL; \
    u.field = (const type *)p
    FXH_PTR(x0_lon, x0_lon, 'd', 8, double, n);
    if (p && n < 3) goto too_short;
    FXH_PTR(x0_lat, x0_lat, 'd', 8, double, n);
    if (p && n < 3) goto too_short;
    FXH_PTR(t_samp, t, 'd', 8, double, n);
    u.nT = (int32_t)n;
    FXH_PTR(v_samp, v, 'd', 8, double, n);
    u.nV = (int32_t)n;
    FXH_PTR(d_samp, d, 'd', 8, double, n);
    u.nD = (int32_t)n;
    FXH_PTR(obs_pos, pos, 'd', 8, double, n_pos);
    FXH_PTR(obs_cov_inv, cov, 'd', 8, double, n_cov);
    FXH_PTR(obs_npred, npred, 'i', 4, int32_t, n_np);
    FXH_PTR(obs_hull, hull, 'd', 8, double, n_hull);
    FXH_PTR(obs_nhull, nhull, 'i', 4, int32_t, n_nh);
#undef FXH_PTR
    /* the obstacle arrays state their own (K, P): npred [K], pos [K][P][2]; the others must agree (fx_update_state then holds
     * (K, P) against the upload's) */
    if (u.obs_npred && u.obs_pos && n_np > 0 && n_pos % (2 * n_np) == 0) {
        const Py_ssize_t K = n_np, P = n_pos / (2 * n_np);
        if ((u.obs_cov_inv && n_cov != 4 * K * P) || (u.obs_hull && n_hull != 6 * K * (P - 1)) || (u.obs_nhull && n_nh != K)) goto too_short;
        u.K = (int32_t)K;
        u.P = (int32_t)P;
    } else if (u.obs_pos || u.obs_cov_inv || u.obs_npred || u.obs_hull || u.obs_nhull) {
        goto too_short;
    }
    if (0) {
    too_short:
        PyErr_SetString(PyExc_ValueError, "state_update: array shapes do not fit together");
        return NULL;
    }
    u.x0_orientation = orient;
    u.v_des = v_des;
    u.low_vel_mode = low_vel;
    memcpy((void *)(uintptr_t)addr, &u, sizeof(u));
    Py_RETURN_NONE;
}

/* ---- the plan steps of a batch of agents: one call from the planners' inputs to their results ----
 * plan_batch(fn_addr, ctx_addr, inputs, yaw_rates, blocks, pkg_addr, update) -> [result dict per agent]  |  int (library error code)
 *   fn_addr   address of fx_plan_batch_packaged, ctx_addr the FxContext*
 *   inputs    sequence of PlanInputs (attributes x0_lon, x0_lat, x0_orientation, v_des, low_vel_mode, t_samp, v_samp, d_samp,
 *             obstacles = packed predictions dict); read only when `update` is true: every agent's state is rewritten in place
 *             (fx_update_state) in front of the evaluation
 *   yaw_rates sequence of floats (row 0 of the packaged yaw-rate column), blocks one C-contiguous float64 buffer
 *             [n][FX_PKG_ROWS][S] the winners' blocks are written to, pkg_addr the address of an FxPackage[n]
 * What engine.plan_batch does with 2 n + 2 ctypes calls and n dict comprehensions (agent_batch.py:140-189's loop over planners). */
typedef int32_t (*fx_batch_fn)(FxContext *, int32_t, const FxStateUpdate *const *, const double *, FxResult *, FxPackage *, double *const *);

#define FXH_MAX_AGENTS 256
static PyObject *s_x0_lon, *s_x0_lat, *s_x0_orientation, *s_v_des, *s_low_vel_mode, *s_t_samp, *s_v_samp, *s_d_samp, *s_obstacles;
static PyObject *s_K, *s_kpos, *s_cov_inv, *s_npred, *s_hull, *s_nhull;
static PyObject *r_keys[10];

/* address of a contiguous buffer attribute / dict item (borrowed lifetime: the inputs outlive the call); *len_out = bytes */
static int addr_of(PyObject *obj, char fmt, Py_ssize_t itemsize, const void **out, Py_ssize_t *len_out) {
    Py_buffer view;
    *out = NULL;
    if (len_out) *len_out = 0;
    if (obj == NULL || obj == Py_None) return 0;
    if (PyObject_GetBuffer(obj, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return -1;
    const char *f = view.format ? view.format : "B";
    if (*f == '=' || *f == '<' || *f == '@') f++;
    const int ok = view.itemsize == itemsize && (f[0] == fmt || (fmt == 'i' && f[0] == 'l' && itemsize == 4)) && f[1] == 0;
    *out = view.buf;
    if (len_out) *len_out = view.len;
    PyBuffer_Release(&view);
    if (!ok) {
        PyErr_SetString(PyExc_TypeError, "expected a C-contiguous float64 / int32 array");
        return -1;
    }
    return 0;
}

static int fill_update(PyObject *inp, FxStateUpdate *u) {
    memset(u, 0, sizeof(*u));
    PyObject *a;
    const void *p;
    Py_ssize_t len;
#define FXH_ATTR(name, field, type, fmt, size, min_len)                                   \
    if (!(a = PyObject_GetAttr(inp, name))) return -1;                                    \
    if (addr_of(a, fmt, size, &p, &len) != 0 || len < (Py_ssize_t)(min_len)) {            \
        Py_DECREF(a);                                                                     \
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "plan_batch: array too short"); \
        return -1;                                                                        \
    }                                                                                     \
    Py_DECREF(a);                                                                         \
    u->field = (const type *)p
    FXH_ATTR(s_x0_lon, x0_lon, double, 'd', 8, 24);
    FXH_ATTR(s_x0_lat, x0_lat, double, 'd', 8, 24);
    FXH_ATTR(s_t_samp, t_samp, double, 'd', 8, 8);
    u->nT = (int32_t)(len / 8);   /* the shapes travel with the pointers: fx_update_state refuses what does not match the upload */
    FXH_ATTR(s_v_samp, v_samp, double, 'd', 8, 8);
    u->nV = (int32_t)(len / 8);
    FXH_ATTR(s_d_samp, d_samp, double, 'd', 8, 8);
    u->nD = (int32_t)(len / 8);
#undef FXH_ATTR
    if (!(a = PyObject_GetAttr(inp, s_x0_orientation))) return -1;
    u->x0_orientation = PyFloat_AsDouble(a);
    Py_DECREF(a);
    if (!(a = PyObject_GetAttr(inp, s_v_des))) return -1;
    u->v_des = PyFloat_AsDouble(a);
    Py_DECREF(a);
    if (PyErr_Occurred()) return -1;
    if (!(a = PyObject_GetAttr(inp, s_low_vel_mode))) return -1;
    const int lv = PyObject_IsTrue(a);
    Py_DECREF(a);
    if (lv < 0) return -1;
    u->low_vel_mode = lv;
    PyObject *o = PyObject_GetAttr(inp, s_obstacles);
    if (!o) return -1;
    int rc = 0;
    if (PyDict_Check(o)) {
        PyObject *k = PyDict_GetItemWithError(o, s_K);
        const long K = k ? PyLong_AsLong(k) : 0;
        if (K > 0) {
            Py_ssize_t l_pos = 0, l_cov = 0, l_np = 0, l_nh = 0;
            if (addr_of(PyDict_GetItemWithError(o, s_kpos), 'd', 8, &p, &l_pos)) rc = -1; else u->obs_pos = (const double *)p;
            if (!rc && addr_of(PyDict_GetItemWithError(o, s_cov_inv), 'd', 8, &p, &l_cov)) rc = -1; else u->obs_cov_inv = (const double *)p;
            if (!rc && addr_of(PyDict_GetItemWithError(o, s_npred), 'i', 4, &p, &l_np)) rc = -1; else u->obs_npred = (const int32_t *)p;
            if (!rc) {
                if (addr_of(PyDict_GetItemWithError(o, s_hull), 'd', 8, &p, &len)) rc = -1;
                else if (p && len > 0) {   /* (also when no hull is left: the old ones must go -- the counts say so) */
                    u->obs_hull = (const double *)p;
                    if (addr_of(PyDict_GetItemWithError(o, s_nhull), 'i', 4, &p, &l_nh)) rc = -1; else u->obs_nhull = (const int32_t *)p;
                }
            }
            if (!rc) {   /* (K, P) as the arrays have them: pos [K][P][2], cov_inv [K][P][4], npred [K], hull [K][P-1][6], nhull [K] */
                const Py_ssize_t P = l_pos / (16 * (Py_ssize_t)K);
                if (l_np != 4 * K || P < 1 || l_pos != 16 * K * P || l_cov != 32 * K * P ||
                    (u->obs_hull && (len != 48 * K * (P - 1) || l_nh != 4 * K))) {
                    PyErr_SetString(PyExc_ValueError, "plan_batch: the packed predictions' arrays do not have the shapes their K says");
                    rc = -1;
                } else {
                    u->K = (int32_t)K;
                    u->P = (int32_t)P;
                }
            }
        }
        if (PyErr_Occurred()) rc = -1;
    } else if (o != Py_None) {
        PyErr_SetString(PyExc_TypeError, "plan_batch: inputs.obstacles must be the packed predictions dict");
        rc = -1;
    }
    Py_DECREF(o);
    return rc;
}

static PyObject *result_dict(const FxResult *r) {
    PyObject *hist = PyList_New(FX_NUM_REASONS);
    if (!hist) return NULL;
    for (int k = 0; k < FX_NUM_REASONS; k++) PyList_SET_ITEM(hist, k, PyLong_FromLongLong(r->reason_hist[k]));
    PyObject *vals[10] = {PyLong_FromLongLong(r->n_candidates), PyLong_FromLongLong(r->best_index), PyFloat_FromDouble(r->best_cost),
                          PyLong_FromLongLong(r->n_returned),   PyLong_FromLongLong(r->n_feasible), PyLong_FromLongLong(r->n_infeasible),
                          PyLong_FromLongLong(r->n_collisions), PyFloat_FromDouble(r->feasible_percentage),
                          PyFloat_FromDouble(r->kernel_ms),     hist};
    PyObject *d = PyDict_New();
    int bad = d == NULL;
    for (int k = 0; k < 10; k++) {
        if (!vals[k] || (!bad && PyDict_SetItem(d, r_keys[k], vals[k]) != 0)) bad = 1;
        Py_XDECREF(vals[k]);
    }
    if (bad) {
        Py_XDECREF(d);
        return NULL;
    }
    return d;
}

/* the agents' state updates from their inputs (borrowed pointers into the inputs' arrays); 0 on success */
static int collect_updates(PyObject *seq, Py_ssize_t n, int update, FxStateUpdate *upd, const FxStateUpdate **updp) {
    for (Py_ssize_t a = 0; a < n; a++) {
        updp[a] = NULL;
        if (update) {
            if (fill_update(PySequence_Fast_GET_ITEM(seq, a), &upd[a]) != 0) return -1;
            updp[a] = &upd[a];
        }
    }
    return 0;
}

/* yaw rates and per-agent block pointers of a [n][FX_PKG_ROWS][S] buffer; *have_blocks = 0 for None; 0 on success */
static int collect_outputs(PyObject *yaws, PyObject *blocks, Py_ssize_t n, double *yaw, double **blk, int *have_blocks) {
    PyObject *yseq = PySequence_Fast(yaws, "yaw_rates must be a sequence");
    if (!yseq) return -1;
    int rc = -1;
    const void *bp;
    Py_ssize_t blen;
    if (PySequence_Fast_GET_SIZE(yseq) != n) {
        PyErr_SetString(PyExc_ValueError, "one yaw rate per agent");
        goto out;
    }
    if (addr_of(blocks, 'd', 8, &bp, &blen) != 0) goto out;
    if (bp && blen % (n * (Py_ssize_t)sizeof(double) * FX_PKG_ROWS) != 0) {
        PyErr_SetString(PyExc_ValueError, "blocks must be [n][FX_PKG_ROWS][S] doubles");
        goto out;
    }
    for (Py_ssize_t a = 0; a < n; a++) {
        yaw[a] = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(yseq, a));
        if (yaw[a] == -1.0 && PyErr_Occurred()) goto out;
        blk[a] = bp ? (double *)bp + a * (blen / (Py_ssize_t)sizeof(double) / n) : NULL;
    }
    *have_blocks = bp != NULL;
    rc = 0;
out:
    Py_DECREF(yseq);
    return rc;
}

static PyObject *results_list(const FxResult *res, Py_ssize_t n) {
    PyObject *out = PyList_New(n);
    for (Py_ssize_t a = 0; out && a < n; a++) {
        PyObject *d = result_dict(&res[a]);
        if (!d) {
            Py_CLEAR(out);
            break;
        }
        PyList_SET_ITEM(out, a, d);
    }
    return out;
}

static PyObject *plan_batch(PyObject *self, PyObject *args) {
    unsigned long long fn_addr, ctx_addr, pkg_addr;
    PyObject *inputs, *yaws, *blocks;
    int update;
    if (!PyArg_ParseTuple(args, "KKOOOKp", &fn_addr, &ctx_addr, &inputs, &yaws, &blocks, &pkg_addr, &update)) return NULL;
    if (!fn_addr || !ctx_addr || !pkg_addr) {
        PyErr_SetString(PyExc_ValueError, "plan_batch: NULL address");
        return NULL;
    }
    PyObject *seq = PySequence_Fast(inputs, "plan_batch: inputs must be a sequence"), *out = NULL;
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    FxStateUpdate upd[FXH_MAX_AGENTS];
    const FxStateUpdate *updp[FXH_MAX_AGENTS];
    double yaw[FXH_MAX_AGENTS], *blk[FXH_MAX_AGENTS];
    FxResult res[FXH_MAX_AGENTS];
    int have_blocks = 0;
    if (n < 1 || n > FXH_MAX_AGENTS) {
        PyErr_SetString(PyExc_ValueError, "plan_batch: 1 .. 256 agents");
        goto done;
    }
    if (collect_outputs(yaws, blocks, n, yaw, blk, &have_blocks) != 0 || collect_updates(seq, n, update, upd, updp) != 0) goto done;
    {
        int32_t rc;
        Py_BEGIN_ALLOW_THREADS
        rc = ((fx_batch_fn)(uintptr_t)fn_addr)((FxContext *)(uintptr_t)ctx_addr, (int32_t)n, update ? updp : NULL, yaw, res,
                                               (FxPackage *)(uintptr_t)pkg_addr, have_blocks ? blk : NULL);
        Py_END_ALLOW_THREADS
        out = rc != 0 ? PyLong_FromLong(rc) /* the caller words the error (fx_last_error) */ : results_list(res, n);
    }
done:
    Py_DECREF(seq);
    return out;
}

/* point_in_polygon(xi, yi, x, y) -> bool: ray casting over the closed outline (xi, yi), the arithmetic of
 * commonroad_xml.Lanelet.contains' array expression term by term (edge k runs from vertex k-1 to vertex k):
 * crossing <=> (yi > y) != (yj > y) and x < (xj - xi) * (y - yi) / (yj - yi) + xi; inside <=> an odd number of crossings.
 * (the goal test of the velocity planner, velocity_planner.py:104-109, once per agent and plan step) */
static PyObject *point_in_polygon(PyObject *self, PyObject *args) {
    PyObject *ox, *oy;
    double x, y;
    if (!PyArg_ParseTuple(args, "OOdd", &ox, &oy, &x, &y)) return NULL;
    const void *px, *py;
    Py_ssize_t lx, ly;
    if (addr_of(ox, 'd', 8, &px, &lx) != 0 || addr_of(oy, 'd', 8, &py, &ly) != 0) return NULL;
    if (!px || !py || lx != ly) {
        PyErr_SetString(PyExc_ValueError, "point_in_polygon: two float64 arrays of one length");
        return NULL;
    }
    const double *xi = (const double *)px, *yi = (const double *)py;
    const Py_ssize_t n = lx / 8;
    int odd = 0;
    for (Py_ssize_t k = 0; k < n; k++) {
        const Py_ssize_t j = k ? k - 1 : n - 1;
        if ((yi[k] > y) != (yi[j] > y)) {
            const double t = (xi[j] - xi[k]) * (y - yi[k]) / (yi[j] - yi[k]) + xi[k];
            if (x < t) odd ^= 1;
        }
    }
    return PyBool_FromLong(odd);
}

/* ---- the inputs of a planner's NEXT closed-loop step: problem.PlanInputs.next_step + sampling.dense_ranges in one call ----
 * next_inputs(prev, low_vel_mode, x_lon, x_lat, x0_orientation, v_des, v_lo, v_hi, d_cached, d0, obstacles) -> PlanInputs | NotImplemented
 *   prev        the planner's last PlanInputs (ranges, dense grid): everything a step does not change is carried over on a copy
 *               of its instance dict, the cached structure key included when the new arrays have the old lengths and the
 *               predictions the old (K, P)
 *   x_lon/x_lat sequences of three floats -> fresh float64 arrays
 *   v_lo, v_hi  the velocity sampling bounds: v = arange(n) * ((v_hi - v_lo) / (n - 1)) + v_lo with the end point set exactly
 *               (np.linspace's arithmetic, sampling.dense_ranges), n = len(prev.v_samp)
 *   d_cached    the grid's lateral set (read-only, shared); d0 is appended on a fresh array when it is not one of its values
 *   obstacles   the packed predictions dict of the step
 * NotImplemented: a case the Python path handles (n < 2, equal bounds, no predictions dict, foreign array types). */
static PyObject *s_empty_tuple;
static PyObject *s_v_samp_k, *s_d_samp_k, *s_t_samp_k, *s_x0_lon_k, *s_x0_lat_k, *s_lvm_k, *s_x0o_k, *s_vdes_k, *s_obst_k, *s_shard_k,
                *s_skey_k, *s_P_k;

static PyObject *array3(PyObject *seq) {
    npy_intp dim = 3;
    PyObject *a = PyArray_SimpleNew(1, &dim, NPY_DOUBLE);
    if (!a) return NULL;
    double *p = (double *)PyArray_DATA((PyArrayObject *)a);
    PyObject *fast = PySequence_Fast(seq, "expected a sequence of three floats");
    if (!fast || PySequence_Fast_GET_SIZE(fast) != 3) {
        Py_XDECREF(fast);
        Py_DECREF(a);
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "expected a sequence of three floats");
        return NULL;
    }
    for (int i = 0; i < 3; i++) {
        p[i] = PyFloat_AsDouble(PySequence_Fast_GET_ITEM(fast, i));
        if (p[i] == -1.0 && PyErr_Occurred()) {
            Py_DECREF(fast);
            Py_DECREF(a);
            return NULL;
        }
    }
    Py_DECREF(fast);
    return a;
}

static int is_f64_vector(PyObject *o) {
    return PyArray_Check(o) && PyArray_NDIM((PyArrayObject *)o) == 1 && PyArray_TYPE((PyArrayObject *)o) == NPY_DOUBLE &&
           PyArray_IS_C_CONTIGUOUS((PyArrayObject *)o);
}

static PyObject *next_inputs(PyObject *self, PyObject *args) {
    PyObject *prev, *lvm, *x_lon, *x_lat, *d_cached, *obstacles;
    double x0o, v_des, v_lo, v_hi, d0;
    if (!PyArg_ParseTuple(args, "OOOOddddOdO", &prev, &lvm, &x_lon, &x_lat, &x0o, &v_des, &v_lo, &v_hi, &d_cached, &d0, &obstacles))
        return NULL;
    PyObject *pd = PyObject_GenericGetDict(prev, NULL);   /* new reference */
    if (!pd) return NULL;
    PyObject *out = NULL, *nd = NULL, *v = NULL, *d = NULL, *lon = NULL, *lat = NULL, *tmp;
    PyObject *pv = PyDict_GetItemWithError(pd, s_v_samp_k), *pdd = PyDict_GetItemWithError(pd, s_d_samp_k);
    PyObject *pobs = PyDict_GetItemWithError(pd, s_obst_k);
    if (!pv || !pdd || !pobs || !is_f64_vector(pv) || !is_f64_vector(pdd) || !is_f64_vector(d_cached) || !PyDict_Check(obstacles) ||
        !PyDict_Check(pobs)) {
        if (!PyErr_Occurred()) { out = Py_NotImplemented; Py_INCREF(out); }
        goto done;
    }
    const npy_intp n_v = PyArray_DIM((PyArrayObject *)pv, 0);
    if (n_v < 2 || v_hi == v_lo || !(v_hi == v_hi) || !(v_lo == v_lo)) { out = Py_NotImplemented; Py_INCREF(out); goto done; }
    {
        npy_intp dim = n_v;
        if (!(v = PyArray_SimpleNew(1, &dim, NPY_DOUBLE))) goto done;
        double *p = (double *)PyArray_DATA((PyArrayObject *)v);
        const double step = (v_hi - v_lo) / (double)(n_v - 1);
        for (npy_intp i = 0; i < n_v; i++) p[i] = (double)i * step + v_lo;
        p[n_v - 1] = v_hi;
    }
    {
        const npy_intp n_d = PyArray_DIM((PyArrayObject *)d_cached, 0);
        const double *q = (const double *)PyArray_DATA((PyArrayObject *)d_cached);
        int found = 0;
        for (npy_intp i = 0; i < n_d; i++) found |= q[i] == d0;
        if (found) { d = d_cached; Py_INCREF(d); }
        else {
            npy_intp dim = n_d + 1;
            if (!(d = PyArray_SimpleNew(1, &dim, NPY_DOUBLE))) goto done;
            double *p = (double *)PyArray_DATA((PyArrayObject *)d);
            memcpy(p, q, sizeof(double) * (size_t)n_d);
            p[n_d] = d0;
        }
    }
    if (!(lon = array3(x_lon)) || !(lat = array3(x_lat))) goto done;
    /* the copy: type(prev).__new__, its dict = prev's with the step's fields replaced */
    out = PyBaseObject_Type.tp_new(Py_TYPE(prev), s_empty_tuple, NULL);
    if (!out) goto done;
    if (!(nd = PyObject_GenericGetDict(out, NULL)) || PyDict_Update(nd, pd) != 0) { Py_CLEAR(out); goto done; }
    {
        int bad = 0;
        bad |= PyDict_SetItem(nd, s_lvm_k, lvm);
        bad |= (tmp = PyFloat_FromDouble(x0o)) ? PyDict_SetItem(nd, s_x0o_k, tmp) : 1; Py_XDECREF(tmp);
        bad |= (tmp = PyFloat_FromDouble(v_des)) ? PyDict_SetItem(nd, s_vdes_k, tmp) : 1; Py_XDECREF(tmp);
        bad |= PyDict_SetItem(nd, s_obst_k, obstacles);
        bad |= PyDict_SetItem(nd, s_shard_k, Py_None);
        bad |= PyDict_SetItem(nd, s_x0_lon_k, lon);
        bad |= PyDict_SetItem(nd, s_x0_lat_k, lat);
        bad |= PyDict_SetItem(nd, s_v_samp_k, v);
        bad |= PyDict_SetItem(nd, s_d_samp_k, d);
        /* the structure key survives equal lengths and equal (K, P) */
        PyObject *k0 = PyDict_GetItemWithError(pobs, s_K), *k1 = PyDict_GetItemWithError(obstacles, s_K);
        PyObject *p0 = PyDict_GetItemWithError(pobs, s_P_k), *p1 = PyDict_GetItemWithError(obstacles, s_P_k);
        int same = k0 && k1 && p0 && p1 && PyObject_RichCompareBool(k0, k1, Py_EQ) == 1 && PyObject_RichCompareBool(p0, p1, Py_EQ) == 1 &&
                   PyArray_DIM((PyArrayObject *)d, 0) == PyArray_DIM((PyArrayObject *)pdd, 0);
        if (!same && PyDict_GetItemWithError(nd, s_skey_k)) bad |= PyDict_DelItem(nd, s_skey_k);
        if (bad || PyErr_Occurred()) Py_CLEAR(out);
    }
done:
    Py_XDECREF(nd);
    Py_XDECREF(pd);
    Py_XDECREF(v);
    Py_XDECREF(d);
    Py_XDECREF(lon);
    Py_XDECREF(lat);
    return out;
}

/* the values of a Python set in ITERATION order as a float64 vector (the reference iterates sets: the candidate index depends on
 * CPython's set order, reactive_planner.py:149-158) */
static PyObject *set_to_array(PyObject *set) {
    npy_intp dim = (npy_intp)PySet_GET_SIZE(set);
    PyObject *a = PyArray_SimpleNew(1, &dim, NPY_DOUBLE);
    if (!a) return NULL;
    double *p = (double *)PyArray_DATA((PyArrayObject *)a);
    PyObject *it = PyObject_GetIter(set), *item;
    npy_intp i = 0;
    if (!it) { Py_DECREF(a); return NULL; }
    while ((item = PyIter_Next(it))) {
        const double x = PyFloat_AsDouble(item);
        Py_DECREF(item);
        if ((x == -1.0 && PyErr_Occurred()) || i >= dim) { Py_DECREF(it); Py_DECREF(a); if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "set changed size"); return NULL; }
        p[i++] = x;
    }
    Py_DECREF(it);
    if (PyErr_Occurred() || i != dim) { Py_DECREF(a); if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "set changed size"); return NULL; }
    return a;
}

/* next_inputs_level(prev, low_vel_mode, x_lon, x_lat, x0_orientation, v_des, v_lo, v_hi, n_v, d_level_set, d0, obstacles) -> inputs
 * The closed loop's usual step at a SAMPLING LEVEL of the reference (sampling_matrix.py:156-182): the velocity samples are
 * set(np.linspace(v_lo, v_hi, n_v)) and the lateral ones d_level_set.union({d0}) IN THE SETS' ITERATION ORDER -- built here with the
 * same set operations (PySet_New over the linspace values, set.union), so the order is CPython's own -- on a copy of the last inputs
 * (see next_inputs).  NotImplemented: a case the Python path handles. */
static PyObject *s_union;
static PyObject *next_inputs_level(PyObject *self, PyObject *args) {
    PyObject *prev, *lvm, *x_lon, *x_lat, *d_set, *obstacles;
    double x0o, v_des, v_lo, v_hi, d0;
    Py_ssize_t n_v;
    if (!PyArg_ParseTuple(args, "OOOOddddnOdO", &prev, &lvm, &x_lon, &x_lat, &x0o, &v_des, &v_lo, &v_hi, &n_v, &d_set, &d0, &obstacles))
        return NULL;
    PyObject *pd = PyObject_GenericGetDict(prev, NULL);   /* new reference */
    if (!pd) return NULL;
    PyObject *out = NULL, *nd = NULL, *v = NULL, *d = NULL, *lon = NULL, *lat = NULL, *tmp, *lst = NULL, *vs = NULL, *one = NULL, *du = NULL;
    PyObject *pv = PyDict_GetItemWithError(pd, s_v_samp_k), *pdd = PyDict_GetItemWithError(pd, s_d_samp_k);
    PyObject *pobs = PyDict_GetItemWithError(pd, s_obst_k);
    if (!pv || !pdd || !pobs || !is_f64_vector(pv) || !is_f64_vector(pdd) || !PyAnySet_Check(d_set) || !PyDict_Check(obstacles) ||
        !PyDict_Check(pobs) || n_v < 2 || n_v > 4096 || v_hi == v_lo || !(v_hi == v_hi) || !(v_lo == v_lo)) {
        if (!PyErr_Occurred()) { out = Py_NotImplemented; Py_INCREF(out); }
        goto done;
    }
    {   /* set(np.linspace(v_lo, v_hi, n_v)): np.linspace's arithmetic, inserted in order */
        if (!(lst = PyList_New(n_v))) goto done;
        const double step = (v_hi - v_lo) / (double)(n_v - 1);
        for (Py_ssize_t i = 0; i < n_v; i++) {
            const double x = i == n_v - 1 ? v_hi : (double)i * step + v_lo;
            if (!(tmp = PyFloat_FromDouble(x))) goto done;
            PyList_SET_ITEM(lst, i, tmp);
        }
        if (!(vs = PySet_New(lst)) || !(v = set_to_array(vs))) goto done;
    }
    {   /* d_level_set.union({d0}) */
        if (!(one = PySet_New(NULL)) || !(tmp = PyFloat_FromDouble(d0))) goto done;
        const int rc = PySet_Add(one, tmp);
        Py_DECREF(tmp);
        if (rc != 0 || !(du = PyObject_CallMethodObjArgs(d_set, s_union, one, NULL)) || !PyAnySet_Check(du) || !(d = set_to_array(du))) goto done;
    }
    if (!(lon = array3(x_lon)) || !(lat = array3(x_lat))) goto done;
    out = PyBaseObject_Type.tp_new(Py_TYPE(prev), s_empty_tuple, NULL);
    if (!out) goto done;
    if (!(nd = PyObject_GenericGetDict(out, NULL)) || PyDict_Update(nd, pd) != 0) { Py_CLEAR(out); goto done; }
    {
        int bad = 0;
        bad |= PyDict_SetItem(nd, s_lvm_k, lvm);
        bad |= (tmp = PyFloat_FromDouble(x0o)) ? PyDict_SetItem(nd, s_x0o_k, tmp) : 1; Py_XDECREF(tmp);
        bad |= (tmp = PyFloat_FromDouble(v_des)) ? PyDict_SetItem(nd, s_vdes_k, tmp) : 1; Py_XDECREF(tmp);
        bad |= PyDict_SetItem(nd, s_obst_k, obstacles);
        bad |= PyDict_SetItem(nd, s_shard_k, Py_None);
        bad |= PyDict_SetItem(nd, s_x0_lon_k, lon);
        bad |= PyDict_SetItem(nd, s_x0_lat_k, lat);
        bad |= PyDict_SetItem(nd, s_v_samp_k, v);
        bad |= PyDict_SetItem(nd, s_d_samp_k, d);
        /* the structure key survives equal lengths and equal (K, P) */
        PyObject *k0 = PyDict_GetItemWithError(pobs, s_K), *k1 = PyDict_GetItemWithError(obstacles, s_K);
        PyObject *p0 = PyDict_GetItemWithError(pobs, s_P_k), *p1 = PyDict_GetItemWithError(obstacles, s_P_k);
        int same = k0 && k1 && p0 && p1 && PyObject_RichCompareBool(k0, k1, Py_EQ) == 1 && PyObject_RichCompareBool(p0, p1, Py_EQ) == 1 &&
                   PyArray_DIM((PyArrayObject *)d, 0) == PyArray_DIM((PyArrayObject *)pdd, 0) &&
                   PyArray_DIM((PyArrayObject *)v, 0) == PyArray_DIM((PyArrayObject *)pv, 0);
        if (!same && PyDict_GetItemWithError(nd, s_skey_k)) bad |= PyDict_DelItem(nd, s_skey_k);
        if (bad || PyErr_Occurred()) Py_CLEAR(out);
    }
done:
    Py_XDECREF(nd); Py_XDECREF(pd); Py_XDECREF(v); Py_XDECREF(d); Py_XDECREF(lon); Py_XDECREF(lat);
    Py_XDECREF(lst); Py_XDECREF(vs); Py_XDECREF(one); Py_XDECREF(du);
    return out;
}

/* plan_batch_begin(fn_addr, ctx_addr, inputs, update) -> None | int (library error code): the first half of plan_batch --
 * fx_plan_batch_begin: every agent's state rewritten from its inputs (update true) and the evaluation launched; returns without
 * waiting.  plan_batch_end(fn_addr, ctx_addr, n, yaw_rates, blocks, pkg_addr) -> [result dict per agent] | int: the second half
 * -- fx_plan_batch_end: the wait, the results, the packages.  A host with several contexts keeps one evaluating while it prepares
 * the next one's inputs and consumes the previous one's results (multiagent.AgentBatchHip, pipeline_groups). */
typedef int32_t (*fx_begin_fn)(FxContext *, int32_t, const FxStateUpdate *const *);
typedef int32_t (*fx_end_fn)(FxContext *, int32_t, const double *, FxResult *, FxPackage *, double *const *);

static PyObject *plan_batch_begin(PyObject *self, PyObject *args) {
    unsigned long long fn_addr, ctx_addr;
    PyObject *inputs;
    int update;
    if (!PyArg_ParseTuple(args, "KKOp", &fn_addr, &ctx_addr, &inputs, &update)) return NULL;
    if (!fn_addr || !ctx_addr) {
        PyErr_SetString(PyExc_ValueError, "plan_batch_begin: NULL address");
        return NULL;
    }
    PyObject *seq = PySequence_Fast(inputs, "plan_batch_begin: inputs must be a sequence"), *out = NULL;
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    FxStateUpdate upd[FXH_MAX_AGENTS];
    const FxStateUpdate *updp[FXH_MAX_AGENTS];
    if (n < 1 || n > FXH_MAX_AGENTS) {
        PyErr_SetString(PyExc_ValueError, "plan_batch_begin: 1 .. 256 agents");
        goto done;
    }
    if (collect_updates(seq, n, update, upd, updp) != 0) goto done;
    {
        int32_t rc;
        Py_BEGIN_ALLOW_THREADS
        rc = ((fx_begin_fn)(uintptr_t)fn_addr)((FxContext *)(uintptr_t)ctx_addr, (int32_t)n, update ? updp : NULL);
        Py_END_ALLOW_THREADS
        if (rc != 0) out = PyLong_FromLong(rc);
        else {
            out = Py_None;
            Py_INCREF(out);
        }
    }
done:
    Py_DECREF(seq);
    return out;
}

static PyObject *plan_batch_end(PyObject *self, PyObject *args) {
    unsigned long long fn_addr, ctx_addr, pkg_addr;
    PyObject *yaws, *blocks;
    int n_in;
    if (!PyArg_ParseTuple(args, "KKiOOK", &fn_addr, &ctx_addr, &n_in, &yaws, &blocks, &pkg_addr)) return NULL;
    if (!fn_addr || !ctx_addr || !pkg_addr) {
        PyErr_SetString(PyExc_ValueError, "plan_batch_end: NULL address");
        return NULL;
    }
    const Py_ssize_t n = n_in;
    double yaw[FXH_MAX_AGENTS], *blk[FXH_MAX_AGENTS];
    FxResult res[FXH_MAX_AGENTS];
    int have_blocks = 0;
    if (n < 1 || n > FXH_MAX_AGENTS) {
        PyErr_SetString(PyExc_ValueError, "plan_batch_end: 1 .. 256 agents");
        return NULL;
    }
    if (collect_outputs(yaws, blocks, n, yaw, blk, &have_blocks) != 0) return NULL;
    int32_t rc;
    Py_BEGIN_ALLOW_THREADS
    rc = ((fx_end_fn)(uintptr_t)fn_addr)((FxContext *)(uintptr_t)ctx_addr, (int32_t)n, yaw, res, (FxPackage *)(uintptr_t)pkg_addr,
                                         have_blocks ? blk : NULL);
    Py_END_ALLOW_THREADS
    return rc != 0 ? PyLong_FromLong(rc) : results_list(res, n);
}

static PyMethodDef methods[] = {
    {"next_inputs_level", next_inputs_level, METH_VARARGS,
     "next_inputs_level(prev, low_vel_mode, x_lon, x_lat, x0_orientation, v_des, v_lo, v_hi, n_v, d_level_set, d0, obstacles) -> inputs | NotImplemented"},
    {"next_inputs", next_inputs, METH_VARARGS,
     "next_inputs(prev, low_vel_mode, x_lon, x_lat, x0_orientation, v_des, v_lo, v_hi, d_cached, d0, obstacles) -> PlanInputs | NotImplemented"},
    {"plan_batch_begin", plan_batch_begin, METH_VARARGS, "plan_batch_begin(fn_addr, ctx_addr, inputs, update) -> None | error code"},
    {"plan_batch_end", plan_batch_end, METH_VARARGS,
     "plan_batch_end(fn_addr, ctx_addr, n, yaw_rates, blocks, pkg_addr) -> [result dict per agent] | error code"},
    {"point_in_polygon", point_in_polygon, METH_VARARGS, "point_in_polygon(xi, yi, x, y) -> bool (ray casting, closed outline)"},
    {"plan_batch", plan_batch, METH_VARARGS,
     "plan_batch(fn_addr, ctx_addr, inputs, yaw_rates, blocks, pkg_addr, update) -> [result dict per agent] | error code"},
    {"state_update", state_update, METH_VARARGS,
     "state_update(struct_addr, x0_lon, x0_lat, x0_orientation, v_des, low_vel_mode, t, v, d, pos, cov_inv, npred, hull, nhull): fill an FxStateUpdate"},
    {"pack_predictions", pack_predictions, METH_VARARGS,
     "pack_predictions(fn_addr, predictions, n_samples, max_obstacles) -> (K, P, out, counts) or None (general path)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fxhost", "host-side helpers of the planner (dict walking in C)", -1, methods};

PyMODINIT_FUNC PyInit__fxhost(void) {
    import_array();
    s_empty_tuple = PyTuple_New(0);
    s_v_samp_k = PyUnicode_InternFromString("v_samp");
    s_d_samp_k = PyUnicode_InternFromString("d_samp");
    s_t_samp_k = PyUnicode_InternFromString("t_samp");
    s_x0_lon_k = PyUnicode_InternFromString("x0_lon");
    s_x0_lat_k = PyUnicode_InternFromString("x0_lat");
    s_lvm_k = PyUnicode_InternFromString("low_vel_mode");
    s_x0o_k = PyUnicode_InternFromString("x0_orientation");
    s_vdes_k = PyUnicode_InternFromString("v_des");
    s_obst_k = PyUnicode_InternFromString("obstacles");
    s_shard_k = PyUnicode_InternFromString("shard");
    s_skey_k = PyUnicode_InternFromString("_skey");
    s_P_k = PyUnicode_InternFromString("P");
    s_union = PyUnicode_InternFromString("union");
    s_pos = PyUnicode_InternFromString("pos_list");
    s_cov = PyUnicode_InternFromString("cov_list");
    s_yaw = PyUnicode_InternFromString("orientation_list");
    s_shape = PyUnicode_InternFromString("shape");
    s_length = PyUnicode_InternFromString("length");
    s_width = PyUnicode_InternFromString("width");
    s_x0_lon = PyUnicode_InternFromString("x0_lon");
    s_x0_lat = PyUnicode_InternFromString("x0_lat");
    s_x0_orientation = PyUnicode_InternFromString("x0_orientation");
    s_v_des = PyUnicode_InternFromString("v_des");
    s_low_vel_mode = PyUnicode_InternFromString("low_vel_mode");
    s_t_samp = PyUnicode_InternFromString("t_samp");
    s_v_samp = PyUnicode_InternFromString("v_samp");
    s_d_samp = PyUnicode_InternFromString("d_samp");
    s_obstacles = PyUnicode_InternFromString("obstacles");
    s_K = PyUnicode_InternFromString("K");
    s_kpos = PyUnicode_InternFromString("pos");
    s_cov_inv = PyUnicode_InternFromString("cov_inv");
    s_npred = PyUnicode_InternFromString("npred");
    s_hull = PyUnicode_InternFromString("hull");
    s_nhull = PyUnicode_InternFromString("nhull");
    {
        static const char *names[10] = {"n_candidates", "best_index", "best_cost", "n_returned", "n_feasible", "n_infeasible",
                                        "n_collisions", "feasible_percentage", "kernel_ms", "reason_hist"};
        for (int k = 0; k < 10; k++) r_keys[k] = PyUnicode_InternFromString(names[k]);
    }
    return PyModule_Create(&moduledef);
}
