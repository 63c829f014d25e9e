"""Runs the host-side tests of the product with the AddressSanitizer / UBSan builds of its host C (csrc/asan/, `make -C csrc asan`):
  * `_fxhost` (csrc/fx_host_ext.c: raw buffer walking behind pack_predictions / state_update / plan_batch / next_inputs / ...),
  * the host half of libfxplan (fx_pack_predictions, fx_invert_cov2, fx_build_obstacle_hulls*, fx_cs_to_curvilinear*, fx_build_boundary_bins,
    fx_wait_word, argument validation) with stub kernel launchers.
Started by tools/asan_host.sh with the sanitizer runtime preloaded.  No GPU involved."""
import os
import sys


def main():
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ASAN = os.path.join(ROOT, "frenetix-motion-planner_amd", "csrc", "asan")
    os.environ["FXPLAN_SO"] = os.path.join(ASAN, "libfxplan_host_asan.so")
    sys.path.insert(0, ROOT)
    import frenetix_motion_planner_amd as pkg   # noqa: E402

    pkg.__path__.insert(0, ASAN)                 # `from . import _fxhost` finds the sanitizer build first
    from frenetix_motion_planner_amd import _fxhost, _lib   # noqa: E402

    assert os.path.dirname(_fxhost.__file__) == ASAN, _fxhost.__file__
    assert _lib.lib()._name == os.environ["FXPLAN_SO"], _lib.lib()._name

    import numpy as np   # noqa: E402
    import pytest        # noqa: E402

    # ---- malformed arguments straight at the extension and the C-ABI's host helpers (each must be refused, none may read out of bounds) ----
    import ctypes as C   # noqa: E402
    L = _lib.lib()
    bad = 0
    for fn, args in (
            (_fxhost.pack_predictions, ({1: dict(pos_list=np.zeros((3, 2)), cov_list=np.zeros((2, 2, 2)), orientation_list=np.zeros(3),
                                                 shape=dict(length=4.0, width=2.0))}, 31, None)),      # covariances shorter than the positions
            (_fxhost.pack_predictions, ({1: dict(pos_list=np.zeros((3, 3)))}, 31, None)),
            (_fxhost.pack_predictions, ("not a dict", 31, None)),
            (_fxhost.point_in_polygon, (np.zeros((2, 3)), 0.0, 0.0)),
            (_fxhost.state_update, (object(),)),
            (_fxhost.plan_batch, (0, 0, [], [], 0)),
            (_fxhost.next_inputs, (None,)),
    ):
        try:
            fn(*args)
        except Exception:   # noqa: BLE001  (refused, as it should be)
            bad += 1
    out = (C.c_double * 4)()
    for m in ([0.0] * 4, [1.0, 2.0, 2.0, 4.0], [float("nan")] * 4, [1e300, 0, 0, 1e-300]):
        L.fx_invert_cov2(1, (C.c_double * 4)(*m), out)
    print(f"malformed-argument calls refused: {bad} of 7", flush=True)

    rc = pytest.main(["-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "-k", "not gloo",   # (spawned ranks test torch.distributed, not host C)
                      os.path.join(ROOT, "tests", "test_planner_host.py"), os.path.join(ROOT, "tests", "test_host_logic.py"),
                      os.path.join(ROOT, "tests", "test_host_golden.py"), os.path.join(ROOT, "tests", "test_multiagent.py"),
                      os.path.join(ROOT, "tests", "test_commonroad_xml.py"), os.path.join(ROOT, "tests", "test_road_boundary.py"),
                      os.path.join(ROOT, "tests", "test_abi.py")])
    print("asan/ubsan host run:", "clean" if rc == 0 else f"pytest exit code {rc}", flush=True)
    return int(rc)


if __name__ == "__main__":
    sys.exit(main())
