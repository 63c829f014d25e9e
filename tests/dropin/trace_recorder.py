"""Record / replay the calls a `frenetix` user makes (TEST CODE).

`Recorder.wrap(top_module)` swaps every class and free function of an installed `frenetix`-shaped module tree for a
recording twin: constructors, handler methods and free functions append an event (names, keyword arguments, array payloads)
to a list.  `save()` writes the events as JSON plus one .npz with the arrays -- data only.  `replay()` re-issues the events
against a module tree (frenetix_compat on the real engine) and returns what the recorded run captured at the same points, so
a GPU test can check the reference adapter's exact call sequence against the HIP engine without the reference being there."""
import json
import logging

import numpy as np

CLASSES = {
    "frenetix": ("TrajectoryHandler", "CoordinateSystemWrapper", "PoseWithCovariance", "PredictedObject", "CartesianPlannerState",
                 "CurvilinearPlannerState", "PlannerState", "SamplingConfiguration"),
    "frenetix.trajectory_functions": ("FillCoordinates",),
    "frenetix.trajectory_functions.feasability_functions": ("CheckYawRateConstraint", "CheckAccelerationConstraint",
                                                            "CheckCurvatureConstraint", "CheckCurvatureRateConstraint"),
    "frenetix.trajectory_functions.cost_functions": (
        "CalculateAccelerationCost", "CalculateJerkCost", "CalculateLateralJerkCost", "CalculateLongitudinalJerkCost",
        "CalculateOrientationOffsetCost", "CalculateDistanceToReferencePathCost", "CalculateCollisionProbabilityFast",
        "CalculateDistanceToObstacleCost", "CalculateVelocityOffsetCost"),
}
HANDLER_METHODS = ("add_feasability_function", "add_cost_function", "add_function", "reset_Trajectories", "generate_trajectories",
                   "generate_stopping_trajectories", "evaluate_all_current_functions", "evaluate_all_current_functions_concurrent",
                   "get_sorted_trajectories")
SORTED_HEAD = 64


def summarise_sorted(trajs):
    """what the adapter reads from get_sorted_trajectories() (reactive_planner_cpp.py:353-358)"""
    return dict(n=len(trajs), ids=[int(t.uniqueId) for t in trajs[:SORTED_HEAD]], costs=[float(t.cost) for t in trajs[:SORTED_HEAD]],
                feasible=[bool(t.feasible) for t in trajs[:SORTED_HEAD]], n_feasible=int(sum(bool(t.feasible) for t in trajs)))


class Recorder:
    def __init__(self):
        self.events, self.arrays, self.ids = [], {}, {}

    # -- encoding of call arguments --
    def enc(self, v):
        if isinstance(v, (bool, int, float, str)) or v is None:
            return v
        if isinstance(v, (np.floating, np.integer, np.bool_)):
            return v.item()
        if isinstance(v, np.ndarray):
            key = f"a{len(self.arrays)}"
            self.arrays[key] = np.array(v)
            return {"@array": key}
        if id(v) in self.ids:
            return {"@obj": self.ids[id(v)]}
        if isinstance(v, (list, tuple)):
            return [self.enc(x) for x in v]
        if isinstance(v, dict):
            return {"@dict": [[self.enc(k), self.enc(x)] for k, x in v.items()]}
        if isinstance(v, logging.Logger):
            return {"@logger": v.name}
        raise TypeError(f"trace recorder: cannot encode {type(v)}")

    def _recording_class(self, cls, name):
        rec = self

        def init(this, *a, **k):
            # arguments first: objects built for this call were registered when THEY were constructed
            ev = {"op": "new", "cls": name, "args": rec.enc(a), "kwargs": {kk: rec.enc(x) for kk, x in k.items()}}
            rec.ids[id(this)] = ev["id"] = len(rec.ids)
            rec._keep.append(this)
            rec.events.append(ev)
            cls.__init__(this, *a, **k)

        body = {"__init__": init}
        if name == "TrajectoryHandler":
            for m in HANDLER_METHODS:
                body[m] = self._recording_method(cls, m)
        return type(cls.__name__, (cls,), body)

    def _recording_method(self, cls, m):
        rec = self

        def call(this, *a, **k):
            ev = {"op": "call", "obj": rec.ids[id(this)], "method": m, "args": rec.enc(a), "kwargs": {kk: rec.enc(x) for kk, x in k.items()}}
            rec.events.append(ev)
            out = getattr(cls, m)(this, *a, **k)
            if m == "get_sorted_trajectories":
                ev["sorted"] = summarise_sorted(list(out))
            return out
        return call

    def wrap(self, modules):
        """modules: dict name -> module object of the installed tree (sys.modules entries)"""
        self._keep = []
        for mod_name, names in CLASSES.items():
            mod = modules[mod_name]
            for n in names:
                setattr(mod, n, self._recording_class(getattr(mod, n), n))
        top = modules["frenetix"]
        inner = top.compute_initial_state
        rec = self

        def compute_initial_state(**k):
            ev = {"op": "func", "name": "compute_initial_state", "kwargs": {kk: rec.enc(x) for kk, x in k.items()}}
            rec.events.append(ev)
            out = inner(**k)
            ev["result"] = {"x0_lon": [float(x) for x in out.x0_lon], "x0_lat": [float(x) for x in out.x0_lat]}
            return out
        top.compute_initial_state = compute_initial_state

    def save(self, path_json, expected):
        np.savez_compressed(path_json.replace(".json", ".npz"), **self.arrays)
        json.dump({"events": self.events, "expected": expected}, open(path_json, "w"), indent=0)


def replay(path_json, namespace, on_new=None):
    """Re-issue a recorded trace.  namespace: class / function name -> callable (frenetix_compat); on_new(obj) is called for
    every object built (tests use it to put a stand-in engine behind the handler).  Returns (objects by id, list of (event,
    result) for the calls that captured a result, the recorded run's outputs)."""
    doc = json.load(open(path_json))
    arrays = np.load(path_json.replace(".json", ".npz"))
    objs, captured = {}, []

    def dec(v):
        if isinstance(v, dict):
            if "@array" in v:
                return np.array(arrays[v["@array"]])
            if "@obj" in v:
                return objs[v["@obj"]]
            if "@dict" in v:
                return {dec(k): dec(x) for k, x in v["@dict"]}
            if "@logger" in v:
                return logging.getLogger(v["@logger"])
        if isinstance(v, list):
            return [dec(x) for x in v]
        return v

    for ev in doc["events"]:
        a = [dec(x) for x in ev.get("args", [])]
        k = {kk: dec(x) for kk, x in ev.get("kwargs", {}).items()}
        if ev["op"] == "new":
            objs[ev["id"]] = namespace[ev["cls"]](*a, **k)
            if on_new is not None:
                on_new(objs[ev["id"]])
        elif ev["op"] == "call":
            out = getattr(objs[ev["obj"]], ev["method"])(*a, **k)
            if ev["method"] == "get_sorted_trajectories":
                captured.append((ev, summarise_sorted(list(out))))
        elif ev["op"] == "func":
            out = namespace[ev["name"]](**k)
            captured.append((ev, {"x0_lon": [float(x) for x in out.x0_lon], "x0_lat": [float(x) for x in out.x0_lat]}))
    return objs, captured, doc["expected"]
