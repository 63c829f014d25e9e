"""Loader of csrc/libfxplan.so (the HIP engine) and its ctypes prototypes (include/fxplan.h).

The product path has NO CPU fallback: if the shared library is missing or cannot be loaded, or no GPU is
visible when an engine is created, this raises -- it never routes to the oracle.
"""
import ctypes as C
import os
import subprocess

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("FXPLAN_SO") or os.path.join(CSRC, "libfxplan.so")  # FXPLAN_SO: developer override (A/B builds)
_LIB = None


class FxError(RuntimeError):
    pass


class FxTimeoutError(FxError):
    """FX_ERR_TIMEOUT: no answer from the device within the context's time bound (a peer that never joined a collective, a
    faulted kernel).  The context is unusable; a multi-rank job ends the process (distributed.exit_on_timeout)."""


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    # the Makefile lists every source and header and is a no-op when the library is up to date
    subprocess.run(["make", "-C", CSRC, "-s"] + (["-B"] if force else []), check=True)
    return SO_PATH


def exported_symbols():
    """Entry points include/fxplan.h declares (checked against the .so by the CPU test-suite)."""
    return [
        "fx_abi_version", "fx_last_error", "fx_device_count", "fx_create", "fx_create_batch", "fx_destroy",
        "fx_set_stream", "fx_set_tuning", "fx_set_block_size", "fx_set_store_mode", "fx_set_winner_buffer", "fx_publish", "fx_wait_published", "fx_set_part_mapping", "fx_build_boundary_bins", "fx_read_boundary_steps", "fx_read_boundary_steps_agent", "fx_set_timing", "fx_set_timing_interval", "fx_read_kernel_times", "fx_set_fused_selection", "fx_set_step_kernel", "fx_step_info", "fx_step_info_ex", "fx_set_obstacle_stage", "fx_last_obstacle_kernel_ms", "fx_read_obstacle_kernel_times", "fx_math_selftest", "fx_upload", "fx_upload_batch", "fx_update_state", "fx_update_step", "fx_comm_unique_id", "fx_comm_check", "fx_comm_set_agents", "fx_comm_info", "fx_set_exchange_mode", "fx_step_exchange_topk", "fx_set_timeout_ms", "fx_wait_word", "fx_comm_init", "fx_comm_destroy", "fx_step_exchange", "fx_set_package", "fx_read_package", "fx_plan_and_package", "fx_plan_batch_packaged", "fx_plan_batch_begin", "fx_plan_batch_end", "fx_cs_to_curvilinear", "fx_cs_to_curvilinear_ex", "fx_build_obstacle_hulls_batch", "fx_invert_cov2", "fx_pack_predictions", "fx_evaluate", "fx_finish", "fx_finish_batch", "fx_plan_step", "fx_step",
        "fx_read_costs", "fx_read_costs_agent", "fx_read_costmap", "fx_read_costmap_agent", "fx_read_coeffs",
        "fx_read_coeffs_agent", "fx_read_lat_tau_agent", "fx_read_sample", "fx_read_sample_agent", "fx_read_candidate_agent", "fx_read_plane", "fx_read_plane_agent",
        "fx_read_topk", "fx_read_topk_batch", "fx_topk_to_device", "fx_build_obstacle_hulls", "fx_device_bytes",
        "fx_last_kernel_ms", "fx_last_eval_kernel_ms", "fx_device_views",
    ]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise FxError(f"{SO_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()); "
                      "the engine has no CPU fallback")
    try:
        L = C.CDLL(SO_PATH)
    except OSError as e:
        raise FxError(f"cannot load {SO_PATH}: {e}") from e
    pd, pi32, pi64, pu32 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint32)
    vp = C.c_void_p
    PP, PR = C.POINTER(_abi.FxProblem), C.POINTER(_abi.FxResult)
    sig = {
        "fx_abi_version": ([], C.c_int32),
        "fx_last_error": ([], C.c_char_p),
        "fx_device_count": ([pi32], C.c_int32),
        "fx_create": ([C.POINTER(vp), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32], C.c_int32),
        "fx_create_batch": ([C.POINTER(vp), C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32],
                            C.c_int32),
        "fx_destroy": ([vp], C.c_int32),
        "fx_set_stream": ([vp, vp], C.c_int32),
        "fx_set_tuning": ([vp, C.c_int32, C.c_int32, C.c_int32], C.c_int32),
        "fx_math_selftest": ([C.c_int32, pd, pd, pd, pd], C.c_int32),
        "fx_set_store_mode": ([vp, C.c_int32], C.c_int32),
        "fx_set_block_size": ([vp, C.c_int32], C.c_int32),
        "fx_set_timing": ([vp, C.c_int32], C.c_int32),
        "fx_set_fused_selection": ([vp, C.c_int32], C.c_int32),
        "fx_set_step_kernel": ([vp, C.c_int32, C.c_int32], C.c_int32),
        "fx_step_info": ([vp, vp], C.c_int32),
        "fx_step_info_ex": ([vp, vp], C.c_int32),
        "fx_set_obstacle_stage": ([vp, C.c_int32, C.c_int32], C.c_int32),
        "fx_last_obstacle_kernel_ms": ([vp], C.c_double),
        "fx_read_obstacle_kernel_times": ([vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)], C.c_int32),
        "fx_set_timing_interval": ([vp, C.c_int32], C.c_int32),
        "fx_read_boundary_steps": ([vp, C.POINTER(C.c_int32)], C.c_int32),
        "fx_read_boundary_steps_agent": ([vp, C.c_int32, C.POINTER(C.c_int32)], C.c_int32),
        "fx_build_boundary_bins": ([C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double),
                                    C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)], C.c_int32),
        "fx_read_kernel_times": ([vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)], C.c_int32),
        "fx_set_part_mapping": ([vp, C.c_int32], C.c_int32),
        "fx_set_winner_buffer": ([vp, vp], C.c_int32),
        "fx_publish": ([vp, vp, C.c_int32], C.c_int32),
        "fx_wait_published": ([vp, pd], C.c_int32),
        "fx_upload": ([vp, PP], C.c_int32),
        "fx_upload_batch": ([vp, C.c_int32, PP], C.c_int32),
        "fx_update_state": ([vp, C.c_int32, C.POINTER(_abi.FxStateUpdate)], C.c_int32),
        "fx_update_step": ([vp, C.POINTER(_abi.FxStateUpdate), PR], C.c_int32),
        "fx_comm_unique_id": ([vp], C.c_int32),
        "fx_comm_init": ([vp, vp, C.c_int32, C.c_int32], C.c_int32),
        "fx_comm_check": ([vp, C.c_int32], C.c_int32),
        "fx_comm_set_agents": ([vp, C.c_int32], C.c_int32),
        "fx_comm_info": ([vp, pi32], C.c_int32),
        "fx_set_exchange_mode": ([vp, C.c_int32], C.c_int32),
        "fx_step_exchange_topk": ([vp, C.c_int32, PR, vp, vp], C.c_int32),
        "fx_set_timeout_ms": ([vp, C.c_int32], C.c_int32),
        "fx_wait_word": ([vp, C.c_uint64, C.c_int32], C.c_int32),
        "fx_comm_destroy": ([vp], C.c_int32),
        "fx_step_exchange": ([vp, PR, vp, vp], C.c_int32),
        "fx_set_package": ([vp, C.c_int32], C.c_int32),
        "fx_read_package": ([vp, C.c_int32, C.c_double, C.POINTER(_abi.FxPackage), pd], C.c_int32),
        "fx_plan_and_package": ([vp, C.POINTER(_abi.FxStateUpdate), C.c_double, PR, C.POINTER(_abi.FxPackage), vp], C.c_int32),   # (block: a plain address)
        "fx_plan_batch_packaged": ([vp, C.c_int32, vp, vp, PR, C.POINTER(_abi.FxPackage), vp], C.c_int32),   # (arrays: plain addresses)
        "fx_plan_batch_begin": ([vp, C.c_int32, vp], C.c_int32),
        "fx_plan_batch_end": ([vp, C.c_int32, vp, PR, C.POINTER(_abi.FxPackage), vp], C.c_int32),
        # host geometry called once per plan step: addresses as integers (array.ctypes.data), no pointer objects on the way
        "fx_cs_to_curvilinear": ([C.c_int32, vp, vp, vp, C.c_double, C.c_double, vp], C.c_int32),
        "fx_cs_to_curvilinear_ex": ([C.c_int32, vp, vp, vp, C.c_double, C.c_double, C.c_int32, vp], C.c_int32),
        "fx_invert_cov2": ([C.c_int32, vp, vp], C.c_int32),
        "fx_pack_predictions": ([C.c_int32, C.c_int32, C.c_int32] + [vp] * 11, C.c_int32),
        "fx_build_obstacle_hulls_batch": ([C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp], C.c_int32),
        "fx_evaluate": ([vp], C.c_int32),
        "fx_finish": ([vp, PR], C.c_int32),
        "fx_finish_batch": ([vp, PR], C.c_int32),
        "fx_step": ([vp, PR], C.c_int32),
        "fx_plan_step": ([vp, PP, PR], C.c_int32),
        "fx_read_costs": ([vp, pd, pu32], C.c_int32),
        "fx_read_costs_agent": ([vp, C.c_int32, pd, pu32], C.c_int32),
        "fx_read_costmap": ([vp, pd], C.c_int32),
        "fx_read_costmap_agent": ([vp, C.c_int32, pd], C.c_int32),
        "fx_read_coeffs": ([vp, C.c_int64, pd, pd, pi32], C.c_int32),
        "fx_read_coeffs_agent": ([vp, C.c_int32, C.c_int64, pd, pd, pi32], C.c_int32),
        "fx_read_lat_tau_agent": ([vp, C.c_int32, C.c_int64, pd], C.c_int32),
        "fx_read_sample": ([vp, C.c_int64, pd], C.c_int32),
        "fx_read_sample_agent": ([vp, C.c_int32, C.c_int64, pd], C.c_int32),
        "fx_read_candidate_agent": ([vp, C.c_int32, C.c_int64, pd, pd, pi32, pd, pd, pu32], C.c_int32),
        "fx_read_plane": ([vp, C.c_int32, pd], C.c_int32),
        "fx_read_plane_agent": ([vp, C.c_int32, C.c_int32, pd], C.c_int32),
        "fx_read_topk": ([vp, C.c_int32, pd, pi64, pi32], C.c_int32),
        "fx_read_topk_batch": ([vp, C.c_int32, pd, pi64], C.c_int32),
        "fx_topk_to_device": ([vp, C.c_int32, vp, vp], C.c_int32),
        "fx_build_obstacle_hulls": ([C.c_int32, pd, pd, C.c_double, C.c_double, pd, pi32], C.c_int32),
        "fx_device_bytes": ([vp], C.c_int64),
        "fx_last_kernel_ms": ([vp], C.c_double),
        "fx_last_eval_kernel_ms": ([vp], C.c_double),
        "fx_device_views": ([vp, C.c_int32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), pi64], C.c_int32),
    }
    for name, (args, ret) in sig.items():
        fn = getattr(L, name)
        fn.argtypes, fn.restype = args, ret
    if L.fx_abi_version() != _abi.FX_ABI_VERSION:
        raise FxError(f"ABI mismatch: library {L.fx_abi_version()} vs python {_abi.FX_ABI_VERSION}")
    _LIB = L
    return L


def check(rc: int):
    """Map a status code to the reference's exception convention: <0 ValueError, >0 RuntimeError."""
    if rc == 0:
        return
    msg = lib().fx_last_error().decode(errors="replace")
    if rc < 0:
        raise ValueError(f"fxplan: {msg} (status {rc})")
    if rc == _abi.FX_ERR_TIMEOUT:
        raise FxTimeoutError(f"fxplan: {msg} (status {rc})")
    raise FxError(f"fxplan: {msg} (status {rc})")
