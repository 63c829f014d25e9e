"""Sampling sets of the Frenet planner (host side, hot-path row a1/a2 of SURVEY.md section 8).

Mirrors the *behaviour* of the reference's frenetix_motion_planner/sampling_matrix.py:
  * density level l of v / d / s uses n = 3, 5, 9, 17, ... points: set(np.linspace(lo, hi, n))
    (sampling_matrix.py:156-182),
  * density level l of t uses set(np.round(np.arange(t_min, horizon + dt, int((1/(l+1))/dt)*dt), 2))
    (sampling_matrix.py:190-195),
  * `to_range(level)` hands out Python *sets*; the planner iterates them, so the candidate index
    g = (it*nV + iv)*nD + id depends on CPython's set iteration order (reactive_planner.py:149-158).
The engine wants ordered arrays, so `ordered()` materialises that iteration order once, on the
host, with the same set construction the reference performs.
"""
import itertools

import numpy as np


class Sampling:
    """Base: a list of sets, one per density level (sampling_matrix.py:124-149)."""

    def __init__(self, minimum: float, maximum: float, max_density: int):
        if maximum < minimum:
            raise AssertionError("maximum < minimum")
        if not isinstance(max_density, int) or max_density <= 0:
            raise AssertionError("max_density must be a positive int")
        self.minimum, self.maximum, self.max_density = minimum, maximum, max_density
        # the level sets are built when they are asked for: a planner renews its velocity sampling every step
        # (set_desired_velocity) and reads one level of it -- or none, with a dense grid
        self._sampling_vec = [None] * max_density

    def _level(self, level: int) -> set:
        raise NotImplementedError

    def to_range(self, sampling_stage: int = 0) -> set:
        if not 0 <= sampling_stage < self.max_density:
            raise AssertionError(f"<Sampling/to_range>: stage {sampling_stage} out of range")
        s = self._sampling_vec[sampling_stage]
        if s is None:
            s = self._sampling_vec[sampling_stage] = self._level(sampling_stage)
        return s

    def ordered(self, sampling_stage: int, extra=None) -> np.ndarray:
        """Iteration order of to_range(stage) (optionally .union({extra})) as an f64 array."""
        s = self.to_range(sampling_stage)
        if extra is not None:
            s = s.union({extra})
        return np.array(list(s), dtype=np.float64)


class _LinspaceSampling(Sampling):
    def _level(self, level: int) -> set:
        n = 2 ** (level + 1) + 1  # 3, 5, 9, 17, ...  (n <- 2n-1)
        return set(np.linspace(self.minimum, self.maximum, n))


class VelocitySampling(_LinspaceSampling):
    pass


class LateralPositionSampling(_LinspaceSampling):
    pass


class LongitudinalPositionSampling(_LinspaceSampling):
    pass


class TimeSampling(Sampling):
    def __init__(self, minimum: float, maximum: float, density: int, dT: float):
        self.dT = dT
        super().__init__(minimum, maximum, density)

    def _level(self, level: int) -> set:
        step = int((1 / (level + 1)) / self.dT)
        return set(np.round(np.arange(self.minimum, self.maximum + self.dT, step * self.dT), 2))


class SamplingHandler:
    """Same constructor and setters as sampling_matrix.py:17-82."""

    def __init__(self, dt: float, max_sampling_number: int, t_min: float, horizon: float, delta_d_min: float,
                 delta_d_max: float, d_ego_pos: bool):
        self.dt = dt
        self.max_sampling_number = max_sampling_number
        self.s_sampling_mode = False
        self.d_ego_pos = d_ego_pos
        self.t_min, self.horizon = t_min, horizon
        self.delta_d_min, self.delta_d_max = delta_d_min, delta_d_max
        self.t_sampling = self.d_sampling = self.v_sampling = self.s_sampling = None
        self.set_t_sampling()
        if not self.d_ego_pos:
            self.set_d_sampling()

    def update_static_params(self, t_min: float, horizon: float, delta_d_min: float, delta_d_max: float):
        assert t_min > 0, "t_min cant be <= 0"
        self.t_min, self.horizon = t_min, horizon
        self.delta_d_min, self.delta_d_max = delta_d_min, delta_d_max
        self.set_t_sampling()
        self.set_d_sampling()

    def change_max_sampling_level(self, max_samp_lvl):
        self.max_sampling_number = max_samp_lvl

    def set_t_sampling(self):
        self.t_sampling = TimeSampling(self.t_min, self.horizon, self.max_sampling_number, self.dt)

    def set_d_sampling(self, lat_pos=None):
        off = 0.0 if not self.d_ego_pos else lat_pos
        self.d_sampling = LateralPositionSampling(off + self.delta_d_min, off + self.delta_d_max,
                                                  self.max_sampling_number)

    def set_v_sampling(self, v_min, v_max):
        self.v_sampling = VelocitySampling(v_min, v_max, self.max_sampling_number)

    def set_s_sampling(self, delta_s_min, delta_s_max):
        self.s_sampling = LongitudinalPositionSampling(delta_s_min, delta_s_max, self.max_sampling_number)

    # ---- engine-facing: ordered arrays for one sampling level ----
    def ordered_ranges(self, level: int, d0: float, *, cpp_style: bool = False, ss0: float = None,
                       t_full: float = None):
        """(t, v, d) in reference iteration order.

        Python back-end: T x V x (D u {d0})                      (reactive_planner.py:149-158)
        C++-handler style: (T u {N dT}) x (V u {ss0}) x (D u {d0}) (reactive_planner_cpp.py:235-237)
        """
        t = self.t_sampling.ordered(level, t_full if cpp_style else None)
        v = self.v_sampling.ordered(level, ss0 if cpp_style else None)
        d = self.d_sampling.ordered(level, d0)
        return t, v, d


    def max_candidates(self, levels=None, cpp_style: bool = False) -> int:
        """Largest |T| x |V| x (|D| + 1) over the sampling levels (|T| + 1, |V| + 1 with the C++ handler's extra values):
        what an engine must hold so that every level of the escalation fits.  The time set is not bounded by 16:
        horizon 5 s at level 3 has 20 values."""
        levels = range(self.max_sampling_number) if levels is None else levels
        extra = 1 if cpp_style else 0
        best = 0
        for lvl in levels:
            n = 2 ** (lvl + 1) + 1
            best = max(best, (len(self.t_sampling.to_range(lvl)) + extra) * (n + extra) * (n + 1))
        return best


def dense_ranges(n_t: int, n_v: int, n_d: int, v_lo: float, v_hi: float, horizon: float, dt: float, d0: float,
                 t_min: float = 1.1, d_min: float = -3.0, d_max: float = 3.0):
    """Dense grid in natural (ascending) order: T = t_min .. horizon step dt (first n_t), V = linspace(v_lo, v_hi, n_v),
    D = linspace(d_min, d_max, n_d) plus d0 appended if absent (BASELINE configs 2 - 5)."""
    t, d, d_set, ramp = dense_cached(n_t, n_v, n_d, horizon, dt, t_min, d_min, d_max)
    # np.linspace(v_lo, v_hi, n_v) with its arithmetic (function_base.py: arange * step + start, the end point set exactly)
    # on the cached ramp -- a third of the time of the call
    if n_v > 1 and v_hi != v_lo:
        v = ramp * ((v_hi - v_lo) / (n_v - 1)) + v_lo
        v[-1] = v_hi
    else:
        v = np.linspace(v_lo, v_hi, n_v)
    if d0 not in d_set:
        d1 = np.empty(len(d) + 1)
        d1[:-1] = d
        d1[-1] = d0
        d = d1
    return t, v, d


_DENSE_CACHE: dict = {}


def dense_cached(n_t: int, n_v: int, n_d: int, horizon: float, dt: float, t_min: float = 1.1, d_min: float = -3.0, d_max: float = 3.0):
    """(T, D, frozenset(D), arange(n_v)) of a dense grid: the time and lateral sets of a planner never change -- built once per
    key, handed out by reference and therefore read-only (a caller that sorts or edits "its" range in place gets an error instead
    of silently changing the sampling sets of every planner with the same key)."""
    key = (n_t, n_v, n_d, horizon, dt, t_min, d_min, d_max)
    c = _DENSE_CACHE.get(key)
    if c is None:
        t = np.round(t_min + dt * np.arange(n_t), 2)
        d = np.linspace(d_min, d_max, n_d)
        t = t[t <= horizon + 1e-9]
        t.setflags(write=False)
        d.setflags(write=False)
        c = _DENSE_CACHE[key] = (t, d, frozenset(d.tolist()), np.arange(0, n_v, dtype=np.float64))
    return c


def generate_sampling_matrix(*, t0_range, t1_range, s0_range, ss0_range, sss0_range, ss1_range, sss1_range,
                             d0_range, dd0_range, ddd0_range, d1_range, dd1_range, ddd1_range):
    """Cartesian product of the 13 ranges, row-major, rows in itertools.product order
    (sampling_matrix.py:85-121).  Built with broadcasting instead of a Python product loop."""
    ranges = [np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in (
        t0_range, t1_range, s0_range, ss0_range, sss0_range, ss1_range, sss1_range, d0_range, dd0_range,
        ddd0_range, d1_range, dd1_range, ddd1_range)]
    grids = np.meshgrid(*ranges, indexing="ij")
    return np.stack([g.reshape(-1) for g in grids], axis=1)


def v_sampling_bounds(current_speed: float, a_max: float, horizon: float, v_max: float, v_limit: float = 36.0):
    """planner.py:304-306 (set_desired_velocity)."""
    min_v = max(0.001, current_speed - a_max * horizon)
    max_v = min(min(current_speed + (a_max / 6.0) * horizon, v_limit), v_max)
    return min_v, max_v
