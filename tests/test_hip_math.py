"""Device elementary functions (csrc/fx_math.h) against NumPy on a real MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ulp_err(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_atan_sin_cos_accuracy():
    from frenetix_motion_planner_amd.engine import math_selftest
    rng = np.random.default_rng(11)
    x = np.concatenate([np.array([0.0, -0.0, 0.4375, 0.6875, 1.1875, 2.4375, 1e-300, 1e300, -1e300, np.pi / 2, np.pi]),
                        rng.uniform(-64, 64, 200_000), rng.normal(size=100_000) * 1e-3, rng.normal(size=50_000) * 1e3,
                        np.linspace(-7, 7, 20_001)])
    at, sn, cs = math_selftest(x)
    assert ulp_err(at, np.arctan(x)).max() <= 1.0
    small = np.abs(x) <= 64
    # sin/cos: absolute error relative to 1 (near zeros of the function the relative error is set by the reduction)
    assert np.abs(sn[small] - np.sin(x[small])).max() < 4e-16
    assert np.abs(cs[small] - np.cos(x[small])).max() < 4e-16
    mid = small & (np.abs(np.sin(x)) > 0.1) & (np.abs(np.cos(x)) > 0.1)
    assert ulp_err(sn[mid], np.sin(x[mid])).max() <= 2.0
    assert ulp_err(cs[mid], np.cos(x[mid])).max() <= 2.0
    assert at[1] == 0 and np.signbit(at[1])  # atan(-0.0) = -0.0
