"""ctypes front-end of oracle/libfxoracle.so (the CPU restatement in fx_oracle.c).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
checker / reported baseline.  The product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from frenetix_motion_planner_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libfxoracle.so")
    src = os.path.join(_HERE, "fx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libfxoracle.so"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        pd, pi32, pi64, pu32 = (C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                C.POINTER(C.c_uint32))
        L.fxo_plan_step.argtypes = [C.POINTER(_abi.FxProblem), pd, pd, pi32, pd, pu32, pd, pd, pi64, pd,
                                    C.POINTER(_abi.FxResult)]
        L.fxo_plan_step.restype = C.c_int32
        L.fxo_plan_step_b.argtypes = [C.POINTER(_abi.FxProblem), pd, pd, pi32, pd, pu32, pd, pd, pi64, pd, pi32,
                                      C.POINTER(_abi.FxResult)]
        L.fxo_plan_step_b.restype = C.c_int32
        L.fxo_plan_step_c.argtypes = [C.POINTER(_abi.FxProblem), pd, pd, pi32, pd, pu32, pd, pd, pi64, pd, pi32, pu32,
                                      C.POINTER(_abi.FxResult)]
        L.fxo_plan_step_c.restype = C.c_int32
        L.fxo_plan_step_d.argtypes = [C.POINTER(_abi.FxProblem), pd, pd, pi32, pd, pu32, pd, pd, pi64, pd, pi32, pu32, pd,
                                      C.POINTER(_abi.FxResult)]
        L.fxo_plan_step_d.restype = C.c_int32
        L.fxo_eval_forced.argtypes = [C.POINTER(_abi.FxProblem), C.c_int64, C.c_uint32, C.c_uint32, pd, pu32, pd, pd, pi32, pu32]
        L.fxo_eval_forced.restype = C.c_int32
        L.fxo_plan_range.argtypes = [C.POINTER(_abi.FxProblem), C.c_int64, C.c_int64, pu32, pd, pi64, pd]
        L.fxo_plan_range.restype = C.c_int32
        L.fxo_plan_range_mt.argtypes = [C.POINTER(_abi.FxProblem), C.c_int64, C.c_int64, C.c_int32, C.c_int32, pu32, pd, pi64, pd]
        L.fxo_plan_range_mt.restype = C.c_int32
        L.fxo_num_candidates.argtypes = [C.POINTER(_abi.FxProblem)]
        L.fxo_num_candidates.restype = C.c_int64
        L.fxo_build_obstacle_hulls.argtypes = [C.c_int32, pd, pd, C.c_double, C.c_double, pd, pi32]
        L.fxo_build_obstacle_hulls.restype = C.c_int32
        L.fxo_project.argtypes = [C.POINTER(_abi.FxProblem), C.c_double, C.c_double, pd]
        L.fxo_project.restype = C.c_int32
        L.fxo_obb_overlap.argtypes = [pd, pd]
        L.fxo_obb_overlap.restype = C.c_int32
        L.fxo_obb_hull.argtypes = [pd, pd, pd, pd, C.c_double, C.c_double, pd]
        L.fxo_np_sum.argtypes = [pd, C.c_int32]
        L.fxo_np_sum.restype = C.c_double
        L.fxo_simpson.argtypes = [pd, C.c_int32, C.c_double, pd]
        L.fxo_simpson.restype = C.c_double
        L.fxo_quartic.argtypes = [C.c_double] * 6 + [pd]
        L.fxo_quintic.argtypes = [C.c_double] * 7 + [pd]
        _LIB = L
    return _LIB


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def build_obstacle_hulls(n_pred, pos, yaw, length, width):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    yaw = np.ascontiguousarray(yaw, dtype=np.float64)
    out = np.zeros((max(n_pred - 1, 0), 6))
    n = C.c_int32(0)
    lib().fxo_build_obstacle_hulls(n_pred, _p(pos), _p(yaw), length, width, _p(out) if out.size else None, C.byref(n))
    return out[:n.value]


def plan_step(inputs, want_planes=True):
    """Run the oracle on a PlanInputs; returns a dict of numpy outputs (candidate-major)."""
    prob = inputs.as_struct()
    Cn = inputs.n_candidates
    S = inputs.n_samples
    nc = len(inputs.cost_names)
    out = dict(
        coeff_lon=np.zeros((Cn, 6)), coeff_lat=np.zeros((Cn, 6)), traj_len=np.zeros(Cn, np.int32),
        planes=np.zeros((Cn, _abi.FX_NUM_PLANES, S)) if want_planes else None,
        flags=np.zeros(Cn, np.uint32), cost=np.zeros(Cn), costmap=np.zeros((Cn, max(nc, 1))),
        order=np.zeros(Cn, np.int64), margin=np.zeros(Cn), boundary_step=np.full(Cn, -1, np.int32),
        frag_sites=np.zeros(Cn, np.uint32), tau_lat=np.zeros(Cn))
    res = _abi.FxResult()
    rc = lib().fxo_plan_step_d(C.byref(prob), _p(out["coeff_lon"]), _p(out["coeff_lat"]), _p(out["traj_len"], C.c_int32),
                               _p(out["planes"]) if want_planes else None, _p(out["flags"], C.c_uint32),
                               _p(out["cost"]), _p(out["costmap"]), _p(out["order"], C.c_int64), _p(out["margin"]),
                               _p(out["boundary_step"], C.c_int32), _p(out["frag_sites"], C.c_uint32), _p(out["tau_lat"]),
                               C.byref(res))
    if rc != 0:
        raise ValueError(f"fxo_plan_step failed: {rc}")
    out["costmap"] = out["costmap"][:, :nc]
    out["result"] = res.as_dict()
    f = out["flags"]
    for name, bit in (("valid", _abi.FX_FLAG_VALID), ("feasible", _abi.FX_FLAG_FEASIBLE),
                      ("collision", _abi.FX_FLAG_COLLISION), ("returned", _abi.FX_FLAG_RETURNED),
                      ("costed", _abi.FX_FLAG_COSTED), ("selectable", _abi.FX_FLAG_SELECTABLE),
                      ("boundary", _abi.FX_FLAG_BOUNDARY)):
        out[name] = (f & bit) != 0
    out["reasons"] = (f >> _abi.FX_REASON_SHIFT) & 0x7FF
    return out


FRAGILE = 1e-9  # FXO_FRAGILE of fx_oracle.c: decisions closer than this to their threshold are taken by rounding noise
SITES = ("lon_goal", "neg", "clamp", "acc_pre", "moving", "v_neg", "kappa", "yaw", "kappa_rate", "a_lo", "a_hi", "dom_lo",
         "dom_hi", "collision", "bound_reach", "bound_hit")


def eval_forced(inputs, g, force_mask=0, force_vals=0):
    """One candidate (index within the evaluated shard) with every fragile decision of the sites in `force_mask` taken the way
    `force_vals` says.  Returns dict(planes [14, S], flags, cost, raw [n_cost], boundary_step, frag_sites)."""
    prob = inputs.as_struct()
    S, nc = inputs.n_samples, len(inputs.cost_names)
    planes = np.zeros((_abi.FX_NUM_PLANES, S))
    raw = np.zeros(max(nc, 1))
    flags, frag, bstep, cost = C.c_uint32(0), C.c_uint32(0), C.c_int32(-1), C.c_double(0.0)
    rc = lib().fxo_eval_forced(C.byref(prob), int(g) + inputs.shard_begin, int(force_mask), int(force_vals), _p(planes),
                               C.byref(flags), C.byref(cost), _p(raw), C.byref(bstep), C.byref(frag))
    if rc != 0:
        raise ValueError(f"fxo_eval_forced failed: {rc}")
    return dict(planes=planes, flags=flags.value, cost=cost.value, raw=raw[:nc], boundary_step=bstep.value, frag_sites=frag.value)


def admissible_outcomes(inputs, g, frag_sites, max_sites=4):
    """Every outcome the reference's arithmetic admits for a candidate whose decisions at the sites `frag_sites` are taken
    by the last ulp: all 2^n assignments of those sites (n capped; beyond the cap the unforced evaluation alone)."""
    sites = [k for k in range(len(SITES)) if (int(frag_sites) >> k) & 1]
    if not sites or len(sites) > max_sites:
        return [eval_forced(inputs, g)]
    mask = sum(1 << k for k in sites)
    outs = []
    for combo in range(1 << len(sites)):
        vals = sum(((combo >> j) & 1) << k for j, k in enumerate(sites))
        outs.append(eval_forced(inputs, g, mask, vals))
    return outs


def plan_range(inputs, g0, g1, n_threads=1, reps=1):
    """Timed leg of bench.py's cpu_baseline: evaluate candidates [g0, g1) without keeping planes (n_threads > 1:
    chunks handed to that many host threads, `reps` whole passes by the same team of threads)."""
    prob = inputs.as_struct()
    n = g1 - g0
    flags = np.zeros(n, np.uint32)
    cost = np.zeros(n)
    best = C.c_int64(-1)
    bc = C.c_double(0)
    if n_threads > 1:
        rc = lib().fxo_plan_range_mt(C.byref(prob), g0, g1, int(n_threads), int(reps), _p(flags, C.c_uint32), _p(cost),
                                     C.byref(best), C.byref(bc))
    else:
        rc = lib().fxo_plan_range(C.byref(prob), g0, g1, _p(flags, C.c_uint32), _p(cost), C.byref(best), C.byref(bc))
    if rc != 0:
        raise ValueError(f"fxo_plan_range failed: {rc}")
    return flags, cost, best.value, bc.value
