"""MI355X-native Frenet sampling-and-evaluation engine (drop-in for the hot path of
TUM-AVS/Frenetix-Motion-Planner: ReactivePlanner.plan() / frenetix.TrajectoryHandler).

Sub-modules are imported lazily; the HIP library (csrc/libfxplan.so) is only loaded when an engine
is created, and that fails loudly -- there is no CPU fallback in the product path.
"""
__version__ = "0.1.0"

from . import _abi  # noqa: F401
from .sampling import SamplingHandler, generate_sampling_matrix  # noqa: F401
from .coordinate_system import CoordinateSystem  # noqa: F401
from .problem import PlanInputs, VehicleParams, DEFAULT_COST_WEIGHTS, pack_predictions  # noqa: F401


def __getattr__(name):
    import importlib
    lazy = {"FrenetEngine": ".engine", "ReactivePlannerHip": ".reactive_planner",
            "TrajectorySample": ".trajectories", "frenetix": ".frenetix_compat",
            "FrenetPlannerInterfaceHip": ".frenet_interface", "AgentBatchHip": ".multiagent",
            "MultiAgentSimulation": ".multiagent", "DataLoggingCosts": ".logging_formats", "SqlLogger": ".logging_formats",
            "ShardedEvaluator": ".distributed"}
    if name in lazy:
        mod = importlib.import_module(lazy[name], __name__)
        return mod if name == "frenetix" else getattr(mod, name)
    raise AttributeError(name)
