"""Host-side mirror of the reference's public interface for the hot path, on top of the C ABI.

Names, argument meaning and error behaviour follow include/icp/icp.h, include/icp/driver.h and
include/icp/config.h of cornellev/icp (the C++ twin of this file is include/icp/*.h in this
repository).  All compute happens in the HIP library; this module only marshals pointers.

Clouds are (n, dim) float64 arrays — numpy (host) or torch CUDA tensors (device, zero-copy into
the C ABI) — whose memory is the reference's column-major Dim x n PointCloud (geo.h:26-29).
Transforms are (dim+1, dim+1) numpy arrays in math layout.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


class Config:
    """icp::Config (include/icp/config.h:9-33): string -> int | double | string."""

    def __init__(self):
        self._params = {}

    def set(self, key, value):
        if not isinstance(value, (int, float, str)) or isinstance(value, bool):
            raise TypeError("Config values are int, double or string")
        self._params[key] = value

    def get(self, key, otherwise):
        if key not in self._params:
            return otherwise
        v = self._params[key]
        # std::get<T> on the wrong alternative throws std::bad_variant_access (config.h:30)
        if type(v) is not type(otherwise):
            raise TypeError(f"bad_variant_access: '{key}' holds {type(v).__name__}, "
                            f"asked for {type(otherwise).__name__}")
        return v


def _ptr_of(x, dim, what):
    """(pointer, n, keepalive) of a cloud given as numpy array or torch tensor."""
    if x is None:
        raise ValueError(f"{what} is None")
    if hasattr(x, "data_ptr"):  # torch tensor (host or device)
        import torch
        if x.dtype != torch.float64:
            x = x.to(torch.float64)
        if x.dim() != 2 or x.shape[1] != dim:
            raise ValueError(f"{what}: expected shape (n, {dim}), got {tuple(x.shape)}")
        x = x.contiguous()
        return x.data_ptr() or None, x.shape[0], x
    a = np.ascontiguousarray(x, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != dim:
        raise ValueError(f"{what}: expected shape (n, {dim}), got {a.shape}")
    return a.ctypes.data or None, a.shape[0], a


def _guess(guess, dim):
    g = np.eye(dim + 1) if guess is None else np.asarray(guess, dtype=np.float64)
    if g.shape != (dim + 1, dim + 1):
        raise ValueError(f"guess must be {(dim + 1, dim + 1)}")
    return np.asfortranarray(g)


@dataclass
class Match:
    """ICP::Match (icp.h:61-70)"""
    point: int
    pair: int
    cost: float


class ICP:
    """icp::ICP<Dim> (include/icp/icp.h:49-170) backed by an icpx handle."""

    dim = 0
    _methods = {}

    def __init__(self, method: int, config: Config | None = None, *, device: int = -1,
                 stream: int | None = None, nn_mode: int = L.NN_AUTO):
        config = config or Config()
        lib = L.load()
        p = L.Params()
        lib.icpx_default_params(C.byref(p))
        # Trimmed(const Config&): trimmed.cpp:28; Trimmed3d(const Config&): trimmed_3d.cpp:27
        p.overlap_rate = config.get("overlap_rate", 0.9)
        p.max_distance = config.get("max_distance", 1.0)
        # FeatureAware(const Config&): feature_aware.cpp:38-41
        p.feature_weight = config.get("feature_weight", 0.7)
        p.symmetric_neighbors = config.get("symmetric_neighbors", 10)
        # optional, non-reference keys (ints 0/1): opt-in corrections of reference quirks Q1 / Q4 / Q5
        p.flags = ((L.FLAG_MATCH_TRANSFORMED if config.get("match_transformed", 0) else 0) |
                   (L.FLAG_FILL_MATCHES if config.get("fill_matches", 0) else 0) |
                   (L.FLAG_APPLY_GUESS if config.get("apply_guess", 0) else 0))
        # optional, non-reference key: "nn" = "auto" | "brute" | "grid"
        names = {"auto": L.NN_AUTO, "brute": L.NN_BRUTE, "grid": L.NN_GRID, "bvh": L.NN_BVH, "grid_f32": L.NN_GRID_F32}
        nn_name = config.get("nn", "")
        if nn_name and nn_name not in names:
            raise ValueError(f"Config 'nn' must be one of {sorted(names)}")
        p.nn_mode = names[nn_name] if nn_name else nn_mode
        p.device = config.get("device", device)
        p.stream = stream
        self._h = L._vp()
        L.check(lib.icpx_create(self.dim, method, C.byref(p), C.byref(self._h)))
        self._n = 0
        self._lib = lib

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.icpx_destroy(h)
            self._h = None

    # --- registry (icp.h:95-114; lib/icp/icp.cpp:17-27) ---
    @classmethod
    def from_method(cls, name: str, config: Config | None = None, **kw):
        """Returns None for an unknown name (the reference returns an empty optional)."""
        if name not in cls._methods:
            return None
        return cls(cls._methods[name], config, **kw)

    @classmethod
    def is_method_registered(cls, name: str) -> bool:
        return name in cls._methods

    @classmethod
    def registered_methods(cls):
        return list(cls._methods.keys())

    # --- icp.h:120-169 ---
    def begin(self, source, target, guess=None):
        sp, n, _s = _ptr_of(source, self.dim, "source")
        tp, m, _t = _ptr_of(target, self.dim, "target")
        g = _guess(guess, self.dim)
        L.check(self._lib.icpx_begin(self._h, sp, n, tp, m, g.ctypes.data))
        self._n = n

    def iterate(self):
        L.check(self._lib.icpx_iterate(self._h))

    def calculate_cost(self) -> float:
        out = C.c_double()
        L.check(self._lib.icpx_cost(self._h, C.byref(out)))
        return out.value

    def current_transform(self) -> np.ndarray:
        d1 = self.dim + 1
        out = np.zeros((d1, d1), order="F")
        L.check(self._lib.icpx_transform(self._h, out.ctypes.data))
        return np.array(out)

    def get_matches_arrays(self):
        p = np.zeros(self._n, np.int64)
        q = np.zeros(self._n, np.int64)
        c = np.zeros(self._n, np.float64)
        L.check(self._lib.icpx_matches(self._h, p.ctypes.data, q.ctypes.data, c.ctypes.data))
        return p, q, c

    def get_matches(self):
        p, q, c = self.get_matches_arrays()
        return [Match(int(a), int(b), float(d)) for a, b, d in zip(p, q, c)]

    def debug_neighbors(self):
        idx = np.zeros(self._n, np.int64)
        dist = np.zeros(self._n, np.float64)
        L.check(self._lib.icpx_debug_neighbors(self._h, idx.ctypes.data, dist.ctypes.data))
        return idx, dist


class ICP2(ICP):
    dim = 2
    _methods = {"vanilla": L.VANILLA, "trimmed": L.TRIMMED, "feature_aware": L.FEATURE_AWARE}


class ICP3(ICP):
    dim = 3
    _methods = {"vanilla": L.VANILLA, "trimmed": L.TRIMMED}


@dataclass
class ConvergenceState:
    """ICPDriver::ConvergenceState (driver.h:24-33)"""
    cost: float = 0.0
    iteration_count: int = 0
    transform: np.ndarray = None


def make_criteria(min_iterations=None, max_iterations=None, stop_cost=None, relative=None,
                  absolute=None, transform_tol=None, time_limit=None) -> L.Criteria:
    c = L.Criteria()
    has = 0
    if min_iterations is not None:
        has |= L.HAS_MIN_ITER
        c.min_iterations = int(min_iterations)
    if max_iterations is not None:
        has |= L.HAS_MAX_ITER
        c.max_iterations = int(max_iterations)
    if stop_cost is not None:
        has |= L.HAS_STOP_COST
        c.stop_cost = float(stop_cost)
    if relative is not None:
        has |= L.HAS_REL_TOL
        c.relative_cost_tolerance = float(relative)
    if absolute is not None:
        has |= L.HAS_ABS_TOL
        c.absolute_cost_tolerance = float(absolute)
    if transform_tol is not None:
        has |= L.HAS_TRANSFORM_TOL
        c.angle_tolerance_rad = float(transform_tol[0])
        c.translation_tolerance = float(transform_tol[1])
    if time_limit is not None:
        has |= L.HAS_TIME_LIMIT
        c.time_limit_s = float(time_limit)
    c.has = has
    return c


class ICPDriver:
    """icp::ICPDriver<Dim> (include/icp/driver.h:20-236).  converge() is ONE C-ABI call: the
    loop and should_terminate run on the device."""

    def __init__(self, icp: ICP):
        self.icp_ = icp
        self._min = self._max = self._stop = self._rel = self._abs = None
        self._tf = None
        self._time = None

    def set_min_iterations(self, v):
        assert self._max is None or v <= self._max  # driver.h:81
        self._min = int(v)

    def set_max_iterations(self, v):
        assert self._min is None or v >= self._min  # driver.h:91
        self._max = int(v)

    def set_stop_cost(self, v):
        self._stop = float(v)

    def set_relative_cost_tolerance(self, v):
        self._rel = float(v)

    def set_absolute_cost_tolerance(self, v):
        self._abs = float(v)

    def set_transform_tolerance(self, angle_tolerance, translation_tolerance):
        self._tf = (float(angle_tolerance), float(translation_tolerance))

    def set_time_limit(self, seconds):
        self._time = float(seconds)

    def criteria(self) -> L.Criteria:
        return make_criteria(self._min, self._max, self._stop, self._rel, self._abs, self._tf,
                             self._time)

    def converge(self, source, target, guess=None) -> ConvergenceState:
        icp = self.icp_
        sp, n, _s = _ptr_of(source, icp.dim, "source")
        tp, m, _t = _ptr_of(target, icp.dim, "target")
        g = _guess(guess, icp.dim)
        crit = self.criteria()
        st = L.State()
        L.check(icp._lib.icpx_converge(icp._h, sp, n, tp, m, g.ctypes.data, C.byref(crit),
                                       C.byref(st)))
        icp._n = n
        d1 = icp.dim + 1
        T = np.array(st.transform[: d1 * d1]).reshape(d1, d1, order="F")
        return ConvergenceState(st.cost, int(st.iteration_count), T)


def converge_batch(dim, method, sources, source_offsets, targets, target_offsets, criteria,
                   config: Config | None = None, guesses=None, out_states=None, device=-1,
                   stream=None, lib=None):
    """icpx_converge_batch.  sources/targets: concatenated (N, dim) clouds (numpy or torch CUDA);
    offsets: int64 numpy arrays (n_pairs+1, in points).  Returns (cost, iterations, transforms)
    numpy arrays unless `out_states` (a torch CUDA uint8/float64 buffer of n_pairs*144 bytes) is
    given, in which case results stay on the device and None is returned.  `lib`: another bound build of the
    library (L.load_variant), default the release one."""
    lib = lib or L.load()
    config = config or Config()
    p = L.Params()
    lib.icpx_default_params(C.byref(p))
    p.overlap_rate = config.get("overlap_rate", 0.9)
    p.max_distance = config.get("max_distance", 1.0)
    p.device = device
    p.stream = stream
    method = {"vanilla": L.VANILLA, "trimmed": L.TRIMMED, "feature_aware": L.FEATURE_AWARE}.get(method, method)
    so = np.ascontiguousarray(source_offsets, dtype=np.int64)
    to = np.ascontiguousarray(target_offsets, dtype=np.int64)
    n_pairs = len(so) - 1
    sp, _, _s = _ptr_of(sources, dim, "sources")
    tp, _, _t = _ptr_of(targets, dim, "targets")
    gp = None
    if guesses is not None:
        gs = np.ascontiguousarray(np.asarray(guesses, np.float64).transpose(0, 2, 1))
        gp = gs.ctypes.data
    if out_states is not None:
        L.check(lib.icpx_converge_batch(dim, method, C.byref(p), n_pairs, sp, so.ctypes.data, tp,
                                        to.ctypes.data, gp, C.byref(criteria),
                                        out_states.data_ptr()))
        return None
    states = (L.State * n_pairs)()
    L.check(lib.icpx_converge_batch(dim, method, C.byref(p), n_pairs, sp, so.ctypes.data, tp,
                                    to.ctypes.data, gp, C.byref(criteria), C.addressof(states)))
    raw = np.frombuffer(states, dtype=np.dtype([("cost", "f8"), ("it", "u8"), ("T", "f8", 16)]))
    d1 = dim + 1
    T = raw["T"][:, : d1 * d1].reshape(n_pairs, d1, d1).transpose(0, 2, 1).copy()
    return raw["cost"].copy(), raw["it"].astype(np.int64), T


def nearest_neighbors(queries, targets, mode=L.NN_AUTO, device=-1, stream=None, out_idx=None,
                      out_d2=None):
    """icpx_nn: exact nearest target of every query under the (d2, index) rule."""
    lib = L.load()
    if hasattr(queries, "data_ptr"):
        dim = queries.shape[1]
    else:
        queries = np.ascontiguousarray(queries, np.float64)
        dim = queries.shape[1]
    qp, n, _q = _ptr_of(queries, dim, "queries")
    tp, m, _t = _ptr_of(targets, dim, "targets")
    if out_idx is not None or out_d2 is not None:
        L.check(lib.icpx_nn(dim, qp, n, tp, m, out_idx.data_ptr() if out_idx is not None else None,
                            out_d2.data_ptr() if out_d2 is not None else None, mode, device, stream))
        return None
    idx = np.zeros(n, np.int64)
    d2 = np.zeros(n, np.float64)
    L.check(lib.icpx_nn(dim, qp, n, tp, m, idx.ctypes.data, d2.ctypes.data, mode, device, stream))
    return idx, d2


def debug_kd_tree(targets, device=-1):
    """icpx_debug_kd_tree (test hook): the in-order node array of the kd rule's tree, as the device builds it."""
    lib = L.load()
    targets = np.ascontiguousarray(targets, np.float64)
    node = np.zeros(len(targets), np.uint32)
    L.check(lib.icpx_debug_kd_tree(targets.shape[1], targets.ctypes.data, len(targets), node.ctypes.data, device))
    return node


def nearest_neighbors_kdtree(queries, targets, mode=L.NN_AUTO, device=-1, stream=None):
    """icpx_nn_kdtree: KdTree::search (kdtree.h:79-89) for every query — the target the reference's tree walk
    returns and its Euclidean distance."""
    lib = L.load()
    if not hasattr(queries, "data_ptr"):
        queries = np.ascontiguousarray(queries, np.float64)
    dim = queries.shape[1]
    qp, n, _q = _ptr_of(queries, dim, "queries")
    tp, m, _t = _ptr_of(targets, dim, "targets")
    idx = np.zeros(n, np.int64)
    dist = np.zeros(n, np.float64)
    L.check(lib.icpx_nn_kdtree(dim, qp, n, tp, m, idx.ctypes.data, dist.ctypes.data, mode, device, stream))
    return idx, dist
