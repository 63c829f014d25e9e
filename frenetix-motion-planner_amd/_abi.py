"""ctypes mirror of include/fxplan.h (struct layouts, enums, flag bits).

Kept free of any library loading so that both the product loader (_lib.py) and the test-side oracle
wrapper (oracle/oracle.py) can share the struct definitions.
"""
import ctypes as C

FX_ABI_VERSION = 9
FX_LON_VELOCITY_KEEPING, FX_LON_STOP_POINT = 0, 1

FX_OK = 0
FX_ERR_INVALID_ARGUMENT = -1
FX_ERR_NOT_READY = -2
FX_ERR_CAPACITY = -3
FX_ERR_HIP = 1
FX_ERR_NO_DEVICE = 2
FX_ERR_TIMEOUT = 3

PLANE_NAMES = ("x", "y", "theta", "v", "a", "kappa", "kappa_dot",
               "s", "d", "theta_cl", "s_dot", "s_ddot", "d_dot", "d_ddot")
FX_NUM_PLANES = len(PLANE_NAMES)
PLANE_INDEX = {n: i for i, n in enumerate(PLANE_NAMES)}

# alphabetical == evaluation order of cost_function.py:55-60
COST_NAMES = ("acceleration", "distance_to_obstacles", "distance_to_reference_path", "jerk", "lane_center_offset", "lateral_jerk",
              "longitudinal_jerk", "orientation_offset", "path_length", "prediction", "velocity_offset")
FX_NUM_COSTS = len(COST_NAMES)
COST_ID = {n: i for i, n in enumerate(COST_NAMES)}
# reference cost names that exist upstream but need scenario / lanelet / reach-set objects (out of scope)
UNSUPPORTED_COSTS = ("velocity", "responsibility", "steering_angle", "steering_rate", "yaw",
                     "time", "inverse_duration", "longitudinal_velocity_offset")

FX_FLAG_VALID = 1 << 0
FX_FLAG_FEASIBLE = 1 << 1
FX_FLAG_COLLISION = 1 << 2
FX_FLAG_RETURNED = 1 << 3
FX_FLAG_COSTED = 1 << 4
FX_FLAG_SELECTABLE = 1 << 5
FX_FLAG_BOUNDARY = 1 << 6
FX_REASON_SHIFT = 8
FX_NUM_REASONS = 11

FX_MODE_DRAW_TRAJ_SET = 1 << 0
FX_MODE_KINEMATIC_DEBUG = 1 << 1
FX_MODE_WRITE_BUNDLE = 1 << 2
FX_MODE_WRITE_COSTMAP = 1 << 3
FX_MODE_COLLISION = 1 << 4
FX_MODE_ROAD_BOUNDARY = 1 << 5
FX_MODE_PROJ_PSEUDO_NORMAL = 1 << 6

_pd = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)


class FxVehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("a_max", "v_switch", "delta_max", "wheelbase", "length", "width", "wb_rear_axle", "kappa_max")]


class FxProblem(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("dt", C.c_double), ("mode", C.c_uint32), ("low_vel_mode", C.c_int32), ("lon_mode", C.c_int32),
        ("x0_lon", C.c_double * 3), ("x0_lat", C.c_double * 3), ("x0_orientation", C.c_double),
        ("v_des", C.c_double), ("veh", FxVehicle),
        ("tpow", _pd),
        ("nT", C.c_int32), ("nV", C.c_int32), ("nD", C.c_int32),
        ("t_samp", _pd), ("v_samp", _pd), ("d_samp", _pd),
        ("sampling_matrix", _pd), ("n_rows", C.c_int64),
        ("shard_begin", C.c_int64), ("shard_count", C.c_int64),
        ("M", C.c_int32),
        ("ref_x", _pd), ("ref_y", _pd), ("ref_nx", _pd), ("ref_ny", _pd),
        ("ref_pos", _pd), ("ref_theta", _pd), ("ref_curv", _pd), ("ref_curv_d", _pd),
        ("n_cost", C.c_int32), ("cost_id", _pi32), ("cost_w", _pd), ("simpson_corr", C.c_double * 3),
        ("K", C.c_int32), ("P", C.c_int32),
        ("obs_pos", _pd), ("obs_cov_inv", _pd), ("obs_npred", _pi32),
        ("obs_hull", _pd), ("obs_nhull", _pi32),
        ("n_dto", C.c_int32), ("dto_pos", _pd),
        ("n_bound", C.c_int32), ("bound_piece", _pd), ("bound_bin", C.POINTER(C.c_int32)), ("bound_item", C.POINTER(C.c_int32)),
        ("bound_d_reach", C.c_double),
        ("n_lane", C.c_int32), ("lane_bbox", _pd), ("lane_poly_off", _pi32), ("lane_poly", _pd), ("lane_ctr_off", _pi32),
        ("lane_ctr", _pd),
    ]


class FxStateUpdate(C.Structure):
    """what changes between two plan steps of a planner (fx_update_state); NULL / NaN / negative = keep"""
    # the pointers as plain addresses (array.ctypes.data): this struct is filled once per plan step
    _fields_ = [
        ("x0_lon", C.c_void_p), ("x0_lat", C.c_void_p), ("x0_orientation", C.c_double), ("v_des", C.c_double), ("low_vel_mode", C.c_int32),
        ("t_samp", C.c_void_p), ("v_samp", C.c_void_p), ("d_samp", C.c_void_p),
        ("obs_pos", C.c_void_p), ("obs_cov_inv", C.c_void_p), ("obs_npred", C.c_void_p), ("obs_hull", C.c_void_p), ("obs_nhull", C.c_void_p),
        # the shapes of the caller's arrays (0 = not stated): a mismatch with the upload is refused instead of read out of bounds
        ("nT", C.c_int32), ("nV", C.c_int32), ("nD", C.c_int32), ("K", C.c_int32), ("P", C.c_int32),
    ]


FX_PKG_ROWS = FX_NUM_PLANES + 3
PKG_ROW_YAW_RATE, PKG_ROW_STEERING, PKG_ROW_ORIENTATION = FX_NUM_PLANES, FX_NUM_PLANES + 1, FX_NUM_PLANES + 2


class FxPackage(C.Structure):
    """the chosen trajectory of a plan step (fx_read_package / fx_plan_and_package)"""
    _fields_ = [
        ("found", C.c_int32), ("S", C.c_int32), ("traj_len", C.c_int32), ("flags", C.c_uint32),
        ("index", C.c_int64), ("cost", C.c_double),
        ("coeff_lon", C.c_double * 6), ("coeff_lat", C.c_double * 6),
        ("n_cost", C.c_int32), ("reserved", C.c_int32),
        ("raw_costs", C.c_double * FX_NUM_COSTS), ("tau_lat", C.c_double),
    ]


class FxResult(C.Structure):
    _fields_ = [
        ("n_candidates", C.c_int64), ("best_index", C.c_int64), ("best_cost", C.c_double),
        ("n_returned", C.c_int64), ("n_feasible", C.c_int64), ("n_infeasible", C.c_int64),
        ("n_collisions", C.c_int64), ("reason_hist", C.c_int64 * FX_NUM_REASONS),
        ("feasible_percentage", C.c_double), ("kernel_ms", C.c_double),
    ]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "reason_hist"}
        d["reason_hist"] = list(self.reason_hist)
        return d
