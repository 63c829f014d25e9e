"""Times the REFERENCE's own Python hot path (/root/reference, imported through ref_harness.py) on BASELINE config 1 -- the
ZAM_Tjunction ego's default plan step, 630 candidates x 31 samples, 5 predicted obstacles, production flag set -- on ONE core of
the build container -- and ONE step of BASELINE config 3's own inputs (the bench default) --, and writes
tests/golden/reference_python_timing.json (data only).  bench.py copies that record into its
line as cpu_baseline.reference_python: the only figure on the line that is the reference's own (it cannot be measured on the GPU
box, where /root/reference does not exist).  Run in the build container only:   python tests/golden/time_reference_python.py

What is timed: _create_trajectory_bundle -> check_feasibility -> TrajectoryBundle.sort (feasible pool), i.e. rows a6 - a16 of
SURVEY 8 -- the same span as one evaluation of the HIP engine -- with multiproc = False.  The (s, d) -> (x, y) projection inside it
is the harness's NumPy restatement (the reference calls commonroad_dc's C++ there, which is not installed): the figure is a little
pessimistic for the reference on that one call."""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
import gen_golden  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ref_harness.install()
    name = "zam_tjunction_ego_l2_prod"
    kw, _ = gen_golden.SCENARIOS[name]
    inp = gen_golden.scenario_inputs(kw)
    prob = gen_golden.to_reference_problem(inp, kw)
    from frenetix_motion_planner.trajectories import TrajectoryBundle
    os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})   # one core

    def step():
        rp = ref_harness.make_planner(prob)
        bundle = rp._create_trajectory_bundle(rp.x_cl[0], rp.x_cl[1], rp.cost_function, samp_level=rp._sampling_min)
        trajs = list(bundle.trajectories)
        returned = rp.check_feasibility(trajs, None, None)
        feas = [o for o in returned if o.valid is True and o.feasible is True]
        b2 = TrajectoryBundle(feas, cost_function=rp.cost_function, multiproc=False, num_workers=1)
        b2.sort()
        return len(trajs), len(feas), b2.trajectories[0].uniqueId

    n, n_feas, best = step()   # warm-up (imports, caches)
    fx = np.load(os.path.join(HERE, name + ".npz"))
    assert n == len(fx["valid"]) and best == int(fx["walk_ids"][0]), "not the step of the committed golden"
    ts = []
    t_end = time.perf_counter() + 20.0
    while time.perf_counter() < t_end or len(ts) < 5:
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    p50 = float(np.median(ts))
    rec = dict(value=n / p50, unit="trajectories/s", cores=1, cpu=cpu_model(), measured_in="build container (not the GPU box)",
               plan_step_p50_ms=p50 * 1e3, plan_step_min_ms=float(min(ts)) * 1e3, steps_timed=len(ts), candidates=n, feasible=n_feas,
               workload="BASELINE config 1: ZAM_Tjunction-1_42_T-1 ego, sampling level 2 (630 candidates x 31 samples), 5 predicted obstacles, "
                        "production flags, multiproc=False",
               method="tests/golden/time_reference_python.py: the reference's _create_trajectory_bundle -> check_feasibility -> "
                      "TrajectoryBundle.sort imported from /root/reference, one pinned core, median of whole plan steps over 20 s",
               python=platform.python_version(), numpy=np.__version__)
    # the bench's DEFAULT workload as well -- BASELINE config 3's own inputs (50 388 candidates x 31 samples, 20 predicted obstacles)
    # through the same three calls, ONE step (two minutes): golden config3_grid_prod_obs20 holds what it returns
    name3 = "config3_grid_prod_obs20"
    kw3, _ = gen_golden.SCENARIOS[name3]
    from frenetix_motion_planner_amd import synthetic
    prob3 = gen_golden.to_reference_problem(synthetic.make_inputs(**kw3), kw3)

    def step3():
        rp = ref_harness.make_planner(prob3)
        bundle = rp._create_trajectory_bundle(rp.x_cl[0], rp.x_cl[1], rp.cost_function, samp_level=rp._sampling_min)
        trajs = list(bundle.trajectories)
        returned = rp.check_feasibility(trajs, None, None)
        feas = [o for o in returned if o.valid is True and o.feasible is True]
        b2 = TrajectoryBundle(feas, cost_function=rp.cost_function, multiproc=False, num_workers=1)
        b2.sort()
        return len(trajs), len(feas), b2.trajectories[0].uniqueId

    t0 = time.perf_counter()
    n3, n3_feas, best3 = step3()
    t3 = time.perf_counter() - t0
    fx3 = np.load(os.path.join(HERE, name3 + ".npz"))
    assert n3 == len(fx3["valid"]) and best3 == int(fx3["walk_ids"][0]), "not the step of the committed config-3 golden"
    rec["config3"] = dict(value=n3 / t3, unit="trajectories/s", plan_step_s=t3, steps_timed=1, candidates=n3, feasible=n3_feas,
                          workload="BASELINE config 3 (the bench default): 19 x 51 x 52 grid = 50 388 candidates x 31 samples, 20 predicted "
                                   "obstacles, production flags, multiproc=False -- the inputs of golden config3_grid_prod_obs20")
    with open(os.path.join(HERE, "reference_python_timing.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
