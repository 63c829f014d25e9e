"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/fxplan.h declares, and its
host-only helpers agree with the oracle (no compute call needs a GPU here)."""
import ctypes as C
import os
import re

import numpy as np

from frenetix_motion_planner_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "fxplan.h")).read()
    declared = set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in fxplan.h but not exported"
    assert set(_lib.exported_symbols()) <= declared
    assert L.fx_abi_version() == _abi.FX_ABI_VERSION


def test_struct_layout_matches_header():
    # sizes the C compiler gives the same declarations
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "fxplan.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(FxVehicle), sizeof(FxProblem), sizeof(FxResult));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", os.path.join(d, "s")], check=True)
        out = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(_abi.FxVehicle), C.sizeof(_abi.FxProblem), C.sizeof(_abi.FxResult)]


def test_create_without_gpu_fails_loudly():
    from frenetix_motion_planner_amd.engine import FrenetEngine, device_count
    if device_count() > 0:
        return
    try:
        FrenetEngine(max_candidates=1000)
    except RuntimeError as e:
        assert "HIP device" in str(e) or "device" in str(e)
    else:
        raise AssertionError("engine creation must fail without a GPU (no CPU fallback)")


def test_obstacle_hull_helper_matches_oracle():
    from frenetix_motion_planner_amd.engine import build_obstacle_hulls
    from oracle import oracle
    rng = np.random.default_rng(5)
    for n in (0, 2, 3, 7, 30):
        pos = np.cumsum(rng.normal(size=(max(n, 1), 2)), axis=0)[:n]
        yaw = rng.uniform(-4, 4, size=n)
        a = build_obstacle_hulls(n, pos, yaw, 4.8, 2.0)
        b = oracle.build_obstacle_hulls(n, pos, yaw, 4.8, 2.0)
        assert a.shape == b.shape
        assert np.allclose(a, b, rtol=0, atol=1e-12)
