"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) and of oracle/_ref/libref_kdtree.so.

TEST INFRASTRUCTURE.  Import this only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (icp_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

VANILLA, TRIMMED, FEATURE_AWARE = 0, 1, 2
FLAG_MATCH_TRANSFORMED, FLAG_FILL_MATCHES, FLAG_APPLY_GUESS = 1, 2, 4
NN_KDTREE, NN_BRUTE, NN_KDCANON = 0, 1, 2
SUM_SEQUENTIAL, SUM_TREE64 = 0, 1

HAS_MIN_ITER, HAS_MAX_ITER, HAS_STOP_COST, HAS_REL_TOL = 1, 2, 4, 8
HAS_ABS_TOL, HAS_TRANSFORM_TOL, HAS_TIME_LIMIT = 16, 32, 64


class Params(C.Structure):
    _fields_ = [("overlap_rate", C.c_double), ("max_distance", C.c_double),
                ("nn_mode", C.c_int32), ("sum_mode", C.c_int32),
                ("feature_weight", C.c_double), ("symmetric_neighbors", C.c_int32), ("flags", C.c_int32)]


class Criteria(C.Structure):
    _fields_ = [("has", C.c_uint32), ("_pad", C.c_uint32),
                ("min_iterations", C.c_uint64), ("max_iterations", C.c_uint64),
                ("stop_cost", C.c_double), ("relative_cost_tolerance", C.c_double),
                ("absolute_cost_tolerance", C.c_double), ("angle_tolerance_rad", C.c_double),
                ("translation_tolerance", C.c_double), ("time_limit_s", C.c_double)]


class State(C.Structure):
    _fields_ = [("cost", C.c_double), ("iteration_count", C.c_uint64),
                ("transform", C.c_double * 16)]


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref when /root/reference is present). Building the checker is
    not using it."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "icp_oracle.cpp")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_missing = os.path.isdir("/root/reference") and not os.path.exists(
        os.path.join(_HERE, "_ref", "libref_kdtree.so"))
    if force or stale or ref_missing:
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


_lib = None
_ref = None
_lib_ref_tree = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def lib_ref_tree():
    """oracle/_ref/liboracle_ref.so: the oracle with NN_KDTREE served by the reference's own kdtree.h (compiled from
    where it lies).  None when not built."""
    global _lib_ref_tree
    if _lib_ref_tree is None:
        path = os.path.join(_HERE, "_ref", "liboracle_ref.so")
        if not os.path.exists(path):
            return None
        _lib_ref_tree = _bind(C.CDLL(path))
    return _lib_ref_tree


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _lib = _bind(C.CDLL(path))
    return _lib


def _bind(L):
    L.orc_nn_brute.argtypes = [C.c_int, _dp, C.c_int64, _dp, C.c_int64, _ip, _dp]
    L.orc_nn_kdtree.argtypes = [C.c_int, _dp, C.c_int64, _dp, C.c_int64, _ip, _dp]
    L.orc_nn_kdcanon.argtypes = [C.c_int, _dp, C.c_int64, _dp, C.c_int64, _ip, _dp, C.c_int, C.POINTER(C.c_int32)]
    L.orc_jacobi_svd.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
    L.orc_kabsch.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
    L.orc_sum.argtypes = [_dp, C.c_int64, C.c_int]
    L.orc_sum.restype = C.c_double
    L.orc_create.argtypes = [C.c_int, C.c_int, C.POINTER(Params)]
    L.orc_create.restype = C.c_void_p
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_begin.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, C.c_int64, _dp]
    L.orc_iterate.argtypes = [C.c_void_p]
    L.orc_cost.argtypes = [C.c_void_p]
    L.orc_cost.restype = C.c_double
    L.orc_transform.argtypes = [C.c_void_p, _dp]
    L.orc_matches.argtypes = [C.c_void_p, _ip, _ip, _dp]
    L.orc_neighbors.argtypes = [C.c_void_p, _ip, _dp]
    L.orc_current_cloud.argtypes = [C.c_void_p, _dp]
    L.orc_converge.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, C.c_int64, _dp,
                               C.POINTER(Criteria), C.POINTER(State), C.c_int64, _dp, _dp]
    return L


def ref_lib():
    """oracle/_ref/libref_kdtree.so: the reference's real kdtree.h. None when not built."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libref_kdtree.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_kdtree_nn.argtypes = [C.c_int, _dp, C.c_int64, _dp, C.c_int64, _ip, _dp]
        L.ref_kdtree_build.argtypes = [C.c_int, _dp, C.c_int64]
        L.ref_kdtree_build.restype = C.c_void_p
        L.ref_kdtree_query.argtypes = [C.c_int, C.c_void_p, _dp, C.c_int64, _ip, _dp]
        L.ref_kdtree_free.argtypes = [C.c_int, C.c_void_p]
        _ref = L
    return _ref


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _cloud(x, dim):
    """Accept (n, dim) row arrays; memory is then x0 y0 (z0) x1 ... = Eigen's Dim x n col-major."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.ndim == 2 and x.shape[1] == dim, x.shape
    return x


def nn_brute(q, t):
    dim = q.shape[1]
    q, t = _cloud(q, dim), _cloud(t, dim)
    idx = np.zeros(len(q), np.int64)
    d2 = np.zeros(len(q), np.float64)
    lib().orc_nn_brute(dim, _d(q), len(q), _d(t), len(t), _i(idx), _d(d2))
    return idx, d2


def nn_kdtree(q, t):
    dim = q.shape[1]
    q, t = _cloud(q, dim), _cloud(t, dim)
    idx = np.zeros(len(q), np.int64)
    dist = np.zeros(len(q), np.float64)
    lib().orc_nn_kdtree(dim, _d(q), len(q), _d(t), len(t), _i(idx), _d(dist))
    return idx, dist


def nn_kdcanon(q, t, by_key=False, want_nodes=False):
    """The kd-tree rule in canonical form (ORC_NN_KDCANON); dist is Euclidean."""
    dim = q.shape[1]
    q, t = _cloud(q, dim), _cloud(t, dim)
    idx = np.zeros(len(q), np.int64)
    dist = np.zeros(len(q), np.float64)
    nodes = np.zeros(max(len(t), 1), np.int32)
    lib().orc_nn_kdcanon(dim, _d(q), len(q), _d(t), len(t), _i(idx), _d(dist), 1 if by_key else 0,
                         nodes.ctypes.data_as(C.POINTER(C.c_int32)))
    return (idx, dist, nodes[:len(t)]) if want_nodes else (idx, dist)


def ref_nn_kdtree(q, t):
    dim = q.shape[1]
    q, t = _cloud(q, dim), _cloud(t, dim)
    idx = np.zeros(len(q), np.int64)
    dist = np.zeros(len(q), np.float64)
    ref_lib().ref_kdtree_nn(dim, _d(q), len(q), _d(t), len(t), _i(idx), _d(dist))
    return idx, dist


def jacobi_svd(a):
    """a: (dim, dim) ndarray (math layout). Returns U, S, V in math layout."""
    dim = a.shape[0]
    acm = np.asfortranarray(a, dtype=np.float64)
    u = np.zeros((dim, dim), order="F")
    v = np.zeros((dim, dim), order="F")
    s = np.zeros(dim)
    lib().orc_jacobi_svd(dim, acm.ctypes.data_as(_dp), u.ctypes.data_as(_dp), _d(s),
                         v.ctypes.data_as(_dp))
    return np.array(u), s, np.array(v)


def kabsch(N, ca, cb):
    dim = N.shape[0]
    ncm = np.asfortranarray(N, dtype=np.float64)
    ca = np.ascontiguousarray(ca, np.float64)
    cb = np.ascontiguousarray(cb, np.float64)
    step = np.zeros((dim + 1, dim + 1), order="F")
    lib().orc_kabsch(dim, ncm.ctypes.data_as(_dp), _d(ca), _d(cb), step.ctypes.data_as(_dp))
    return np.array(step)


def osum(v, mode):
    v = np.ascontiguousarray(v, np.float64)
    return lib().orc_sum(_d(v), len(v), mode)


def make_criteria(min_iterations=None, max_iterations=None, stop_cost=None, relative=None,
                  absolute=None, transform_tol=None, time_limit=None, cls=Criteria):
    c = cls()
    has = 0
    if min_iterations is not None:
        has |= HAS_MIN_ITER
        c.min_iterations = int(min_iterations)
    if max_iterations is not None:
        has |= HAS_MAX_ITER
        c.max_iterations = int(max_iterations)
    if stop_cost is not None:
        has |= HAS_STOP_COST
        c.stop_cost = float(stop_cost)
    if relative is not None:
        has |= HAS_REL_TOL
        c.relative_cost_tolerance = float(relative)
    if absolute is not None:
        has |= HAS_ABS_TOL
        c.absolute_cost_tolerance = float(absolute)
    if transform_tol is not None:
        has |= HAS_TRANSFORM_TOL
        c.angle_tolerance_rad = float(transform_tol[0])
        c.translation_tolerance = float(transform_tol[1])
    if time_limit is not None:
        has |= HAS_TIME_LIMIT
        c.time_limit_s = float(time_limit)
    c.has = has
    return c


@dataclass
class Result:
    cost: float
    iteration_count: int
    transform: np.ndarray
    trace_cost: np.ndarray = field(default=None)
    trace_transform: np.ndarray = field(default=None)


class OracleICP:
    """Mirror of icp::ICP<Dim> (icp.h) + ICPDriver::converge (driver.h) on the CPU oracle."""

    def __init__(self, dim, method, overlap_rate=0.9, max_distance=1.0, nn_mode=NN_KDTREE,
                 sum_mode=SUM_SEQUENTIAL, feature_weight=0.7, symmetric_neighbors=10, flags=0, reference_tree=False):
        # reference_tree: NN_KDTREE through the reference's own kdtree.h (oracle/_ref/liboracle_ref.so)
        self._L = lib_ref_tree() if reference_tree else lib()
        assert self._L is not None, "oracle/_ref/liboracle_ref.so is not built"
        self.dim = dim
        self.method = {"vanilla": VANILLA, "trimmed": TRIMMED, "feature_aware": FEATURE_AWARE}.get(method, method)
        p = Params(overlap_rate, max_distance, nn_mode, sum_mode, feature_weight, symmetric_neighbors, flags)
        self._h = self._L.orc_create(dim, self.method, C.byref(p))
        assert self._h
        self.n = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    def _guess(self, guess):
        d1 = self.dim + 1
        g = np.eye(d1) if guess is None else np.asarray(guess, np.float64)
        return np.asfortranarray(g)

    def begin(self, src, dst, guess=None):
        src, dst = _cloud(src, self.dim), _cloud(dst, self.dim)
        g = self._guess(guess)
        self.n = len(src)
        self._L.orc_begin(self._h, _d(src), len(src), _d(dst), len(dst), g.ctypes.data_as(_dp))

    def iterate(self):
        self._L.orc_iterate(self._h)

    def calculate_cost(self):
        return self._L.orc_cost(self._h)

    def current_transform(self):
        d1 = self.dim + 1
        out = np.zeros((d1, d1), order="F")
        self._L.orc_transform(self._h, out.ctypes.data_as(_dp))
        return np.array(out)

    def get_matches(self):
        p = np.zeros(self.n, np.int64)
        q = np.zeros(self.n, np.int64)
        c = np.zeros(self.n, np.float64)
        self._L.orc_matches(self._h, _i(p), _i(q), _d(c))
        return p, q, c

    def neighbors(self):
        idx = np.zeros(self.n, np.int64)
        dist = np.zeros(self.n, np.float64)
        self._L.orc_neighbors(self._h, _i(idx), _d(dist))
        return idx, dist

    def current_cloud(self):
        out = np.zeros((self.n, self.dim))
        self._L.orc_current_cloud(self._h, _d(out))
        return out

    def converge(self, src, dst, guess=None, criteria=None, trace_cap=0):
        src, dst = _cloud(src, self.dim), _cloud(dst, self.dim)
        g = self._guess(guess)
        self.n = len(src)
        d1 = self.dim + 1
        st = State()
        tc = np.zeros(max(trace_cap, 1))
        tt = np.zeros((max(trace_cap, 1), d1 * d1))
        self._L.orc_converge(self._h, _d(src), len(src), _d(dst), len(dst), g.ctypes.data_as(_dp),
                           C.byref(criteria), C.byref(st), trace_cap, _d(tc), _d(tt))
        T = np.array(st.transform[: d1 * d1]).reshape(d1, d1, order="F")
        k = min(int(st.iteration_count), trace_cap)
        return Result(st.cost, int(st.iteration_count), T, tc[:k].copy(),
                      tt[:k].reshape(k, d1, d1).transpose(0, 2, 1).copy())
