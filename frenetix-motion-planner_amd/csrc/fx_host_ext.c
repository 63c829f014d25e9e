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
static int buf_addr(PyObject *obj, char fmt, Py_ssize_t itemsize, const void **out) {
    *out = NULL;
    if (obj == Py_None) return 0;
    Py_buffer view;
    if (PyObject_GetBuffer(obj, &view, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return -1;
    const char *f = view.format ? view.format : "B";
    if (*f == '=' || *f == '<' || *f == '@') f++;
    const int ok = view.itemsize == itemsize && (f[0] == fmt || (fmt == 'i' && f[0] == 'l' && itemsize == 4)) && f[1] == 0;
    *out = view.buf;
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
#define FXH_PTR(field, obj, fmt, size, type)                 \
    if (buf_addr(obj, fmt, size, &p) != 0) return NULL; \
    u.field = (const type *)p
    FXH_PTR(x0_lon, x0_lon, 'd', 8, double);
    FXH_PTR(x0_lat, x0_lat, 'd', 8, double);
    FXH_PTR(t_samp, t, 'd', 8, double);
    FXH_PTR(v_samp, v, 'd', 8, double);
    FXH_PTR(d_samp, d, 'd', 8, double);
    FXH_PTR(obs_pos, pos, 'd', 8, double);
    FXH_PTR(obs_cov_inv, cov, 'd', 8, double);
    FXH_PTR(obs_npred, npred, 'i', 4, int32_t);
    FXH_PTR(obs_hull, hull, 'd', 8, double);
    FXH_PTR(obs_nhull, nhull, 'i', 4, int32_t);
#undef FXH_PTR
    u.x0_orientation = orient;
    u.v_des = v_des;
    u.low_vel_mode = low_vel;
    memcpy((void *)(uintptr_t)addr, &u, sizeof(u));
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"state_update", state_update, METH_VARARGS,
     "state_update(struct_addr, x0_lon, x0_lat, x0_orientation, v_des, low_vel_mode, t, v, d, pos, cov_inv, npred, hull, nhull): fill an FxStateUpdate"},
    {"pack_predictions", pack_predictions, METH_VARARGS,
     "pack_predictions(fn_addr, predictions, n_samples, max_obstacles) -> (K, P, out, counts) or None (general path)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fxhost", "host-side helpers of the planner (dict walking in C)", -1, methods};

PyMODINIT_FUNC PyInit__fxhost(void) {
    s_pos = PyUnicode_InternFromString("pos_list");
    s_cov = PyUnicode_InternFromString("cov_list");
    s_yaw = PyUnicode_InternFromString("orientation_list");
    s_shape = PyUnicode_InternFromString("shape");
    s_length = PyUnicode_InternFromString("length");
    s_width = PyUnicode_InternFromString("width");
    return PyModule_Create(&moduledef);
}
