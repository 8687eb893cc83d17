"""ctypes loader for icp_amd/lib/libicpx.so (C ABI: include/icpx/icpx.h).

There is no fallback: if the HIP library is missing or no gfx950 device is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libicpx.so")

OK, ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_UNSUPPORTED, ERR_NO_DEVICE = range(7)
VANILLA, TRIMMED, FEATURE_AWARE = 0, 1, 2
FLAG_MATCH_TRANSFORMED, FLAG_FILL_MATCHES, FLAG_APPLY_GUESS = 1, 2, 4
NN_AUTO, NN_BRUTE, NN_GRID, NN_BVH, NN_GRID_F32 = 0, 1, 2, 3, 4
HAS_MIN_ITER, HAS_MAX_ITER, HAS_STOP_COST, HAS_REL_TOL = 1, 2, 4, 8
HAS_ABS_TOL, HAS_TRANSFORM_TOL, HAS_TIME_LIMIT = 16, 32, 64


class Params(C.Structure):
    _fields_ = [("overlap_rate", C.c_double), ("max_distance", C.c_double),
                ("nn_mode", C.c_int32), ("device", C.c_int32), ("stream", C.c_void_p),
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


class IcpxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"icpx error {code}: {msg}")
        self.code = code


# every symbol include/icpx/icpx.h declares
SYMBOLS = [
    "icpx_init", "icpx_version", "icpx_last_error", "icpx_default_params", "icpx_create",
    "icpx_destroy", "icpx_begin", "icpx_iterate", "icpx_cost", "icpx_transform", "icpx_matches",
    "icpx_sizes", "icpx_debug_neighbors", "icpx_converge", "icpx_converge_batch", "icpx_nn",
    "icpx_nn_grid_profile", "icpx_converge_batch_multi", "icpx_debug_knob", "icpx_debug_kd_tree",
    "icpx_converge_batch_noalloc", "icpx_nn_kdtree",
]

_lib = None
_vp = C.c_void_p


def load():
    """Load libicpx.so. Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        # development: ICPX_LIB names another build (make -C icp_amd/csrc stats)
        _lib = _bind(os.environ.get("ICPX_LIB", LIB_PATH))
    return _lib


def load_variant(libdir: str):
    """A second, separately bound build of the library (icp_amd/<libdir>/libicpx.so), or None when it has not
    been built: bench.py counts executed work with the -DICPX_STATS build beside the release library."""
    path = os.path.join(_HERE, libdir, "libicpx.so")
    return _bind(path) if os.path.exists(path) else None


def _bind(path):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP library first (make -C icp_amd/csrc). "
            "icp_amd has no CPU fallback.")
    L = C.CDLL(path)
    if hasattr(L, "icpx_debug_stats"):
        L.icpx_debug_stats.argtypes = [_vp, C.c_int]
    L.icpx_init.argtypes = [C.c_int]
    L.icpx_version.restype = C.c_int
    L.icpx_last_error.restype = C.c_char_p
    L.icpx_default_params.argtypes = [C.POINTER(Params)]
    L.icpx_default_params.restype = None
    L.icpx_create.argtypes = [C.c_int, C.c_int, C.POINTER(Params), C.POINTER(_vp)]
    L.icpx_destroy.argtypes = [_vp]
    L.icpx_begin.argtypes = [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp]
    L.icpx_iterate.argtypes = [_vp]
    L.icpx_cost.argtypes = [_vp, C.POINTER(C.c_double)]
    L.icpx_transform.argtypes = [_vp, _vp]
    L.icpx_matches.argtypes = [_vp, _vp, _vp, _vp]
    L.icpx_sizes.argtypes = [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.icpx_debug_neighbors.argtypes = [_vp, _vp, _vp]
    L.icpx_converge.argtypes = [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp, C.POINTER(Criteria),
                                C.POINTER(State)]
    L.icpx_nn_grid_profile.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int64, C.c_int, _vp, C.POINTER(C.c_double)]
    L.icpx_converge_batch.argtypes = [C.c_int, C.c_int, C.POINTER(Params), C.c_int64, _vp, _vp,
                                      _vp, _vp, _vp, C.POINTER(Criteria), _vp]
    L.icpx_converge_batch_multi.argtypes = [C.c_int, C.c_int, C.POINTER(Params), C.c_int, _vp, C.c_int64, _vp, _vp,
                                            _vp, _vp, _vp, C.POINTER(Criteria), _vp]
    L.icpx_debug_knob.argtypes = [C.c_char_p, C.c_char_p]
    L.icpx_debug_kd_tree.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    if hasattr(L, "icpx_converge_batch_noalloc"):
        L.icpx_converge_batch_noalloc.argtypes = L.icpx_converge_batch.argtypes
    L.icpx_nn.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_int, C.c_int, _vp]
    L.icpx_nn_kdtree.argtypes = [C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_int, C.c_int, _vp]
    return L


def check(rc):
    if rc != OK:
        raise IcpxError(rc, load().icpx_last_error().decode("utf-8", "replace"))
