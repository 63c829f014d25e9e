import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
import oracle.oracle as om
# point the wrapper at the sanitizer build
om._LIB = None
om.build = lambda force=False: os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', 'libfxoracle_asan.so')
from frenetix_motion_planner_amd import synthetic

from tests.test_hip_parity import _random_case
n = n_forced = 0
for case in range(3000, 3120):
    kw = _random_case(np.random.default_rng([20241008, case]))
    inp = synthetic.make_inputs(hull_builder=om.build_obstacle_hulls, **kw)
    out = om.plan_step(inp)
    f, c, b, bc = om.plan_range(inp, 0, inp.n_candidates, n_threads=3, reps=2)
    assert np.array_equal(f, out["flags"])
    # forced-decision evaluation (fxo_eval_forced): every fragile candidate both ways, and a few robust ones with forced sites
    frag = np.nonzero(out["margin"] < om.FRAGILE)[0]
    for g in list(frag[:6]) + [0, inp.n_candidates - 1]:
        sites = int(out["frag_sites"][g]) or 0x5
        for o in om.admissible_outcomes(inp, int(g), sites):
            assert o["planes"].shape[0] == 14
        n_forced += 1
    n += inp.n_candidates
print("asan/ubsan run ok:", n, "candidates,", n_forced, "forced-decision evaluations")
