import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The torch wheel of this image bundles its own HIP runtime (torch/lib/libamdhip64.so, ROCm 7.0) while
    # libicpx.so links the system one (/opt/rocm, 7.2).  Both can live in one process when torch's comes up
    # first; the other way round torch reports "No HIP GPUs are available".  Tests that use torch tensors
    # import it lazily, so bring it up here, whatever subset of the tests runs.
    try:
        import torch
        torch.cuda.is_available()
    except Exception:  # no torch: the tests that need it fail on their own import
        pass


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def scans():
    from icp_amd import io as iio
    return {k: (iio.parse_lidar_scan(f"{GOLDEN}/ex_data/scan{k}/first.csv"),
                iio.parse_lidar_scan(f"{GOLDEN}/ex_data/scan{k}/second.csv")) for k in (1, 2, 3)}


@pytest.fixture(scope="session")
def queen():
    """(source, target) as tests/test_ply.cpp uses them: queen_transformed -> queen."""
    from icp_amd import io as iio
    return (iio.read_ply_vertices(f"{GOLDEN}/ex_data/ply/queen_transformed.ply"),
            iio.read_ply_vertices(f"{GOLDEN}/ex_data/ply/queen.ply"))


@pytest.fixture(scope="session")
def traces():
    with open(os.path.join(GOLDEN, "traces.json")) as f:
        raw = json.load(f)
    out = {}
    for k, v in raw.items():
        if k.startswith("_"):
            continue
        d1 = int(round(len(v["transform"]) ** 0.5))
        out[k] = {
            "iterations": v["iterations"],
            "cost": float.fromhex(v["cost"]),
            "transform": np.array([float.fromhex(x) for x in v["transform"]]).reshape(d1, d1),
            "trace_cost": np.array([float.fromhex(x) for x in v["trace_cost"]]),
            "trace_transform": np.array([[float.fromhex(x) for x in t] for t in v["trace_transform"]]).reshape(-1, d1, d1),
        }
    return out


class _Knobs:
    """Development knobs of libicpx (icpx_debug_knob): the library reads the ICPX_* environment once, so
    tests that switch kernels inside one process set the knob instead, and get the defaults back afterwards."""

    def __init__(self):
        from icp_amd import _lib
        self._lib, self._check, self._touched = _lib.load(), _lib.check, set()

    def set(self, name, value):
        self._touched.add(name)
        self._check(self._lib.icpx_debug_knob(name.encode(), str(value).encode()))

    def reset(self, name):
        self._check(self._lib.icpx_debug_knob(name.encode(), None))

    def restore(self):
        for name in self._touched:
            self.reset(name)


@pytest.fixture
def knob():
    k = _Knobs()
    yield k
    k.restore()


BENCH_CRIT = dict(max_iterations=50, transform_tol=(0.1 * np.pi / 180, 0.01))  # bench/bench.cpp:109-113
PLY_CRIT = dict(max_iterations=100, transform_tol=(1e-4, 1e-4))               # tests/test_ply.cpp:22-23
TEST_CRIT = dict(max_iterations=100, transform_tol=(0.1 * np.pi / 180, 0.1))  # tests/test.cpp:87-88


def rot2(deg):
    a = np.radians(deg)
    return np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])


def rot3_zyx(dx, dy, dz):
    """(rot_z * rot_y * rot_x).toRotationMatrix(), tests/test3d.cpp:109"""
    ax, ay, az = np.radians([dx, dy, dz])
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def reference_cases_2d():
    """The six cases of the reference's test_icp_generic (tests/test.cpp:83-216) as
    (name, a, b, expected_R(2x2), expected_t, trans_eps, rad_eps, max_iterations or None)."""
    TE, RE = 0.5, 0.01
    cases = []
    cases.append(("single_point", np.array([[0., 0.]]), np.array([[100., 0.]]), rot2(0), [100, 0], TE, RE, 10))
    a = np.array([[0., 0.], [100., 100.]])
    cases.append(("identity", a, a.copy(), rot2(0), [0, 0], TE, RE, None))
    for deg in range(10):
        a = np.array([[-100., -100.], [100., 100.]])
        cases.append((f"rot3a_{deg}", a, a @ rot2(deg).T, rot2(deg), [0, 0], TE, RE, None))
    for deg in range(10):
        a = np.array([[0., 0.], [1., 0.]])
        cases.append((f"rot3b_{deg}", a, a @ rot2(deg).T, rot2(deg), [0, 0], TE, RE, None))
    cases.append(("translation", np.array([[0., 0.], [0., 100.]]), np.array([[100., 0.], [100., 100.]]),
                  rot2(0), [100, 0], TE, RE, None))
    a = np.array([[0., 0.], [100., 0.]])
    cases.append(("rot_trans", a, a @ rot2(45).T + [5, 5], rot2(45), [5, 5], TE, RE, None))
    a = np.array([[0., 0.], [100., 0.], [50., 50.], [0., 50.]])
    noise = np.fromfile(os.path.join(GOLDEN, "noise2d_f64.bin")).reshape(4, 2)
    cases.append(("noise", a, a @ rot2(30).T + [20, 10] + noise, rot2(30), [20, 10], TE * 3, RE * 5, None))
    return cases


def reference_cases_3d():
    """The five cases of tests/test3d.cpp:36-202 as (name, a, b, R, t, trans_eps, rot_eps, max_it)."""
    TE, RE = 0.5, 0.01
    cases = []
    cases.append(("single_point", np.array([[0., 0., 0.]]), np.array([[100., 0., 0.]]), np.eye(3), [100, 0, 0], TE, RE, 10))
    a = np.eye(3)
    c = a.mean(axis=0)
    for deg in range(10):
        R = rot3_zyx(deg, 0, 0)
        t = c - R @ c
        cases.append((f"rotx_{deg}", a, a @ R.T + t, R, t, TE, RE, None))
    R = rot3_zyx(30, 30, 30)
    t = c - R @ c
    cases.append(("rot_multi", a, a @ R.T + t, R, t, TE, RE, None))
    a = 100.0 * np.eye(3)
    R = rot3_zyx(10, 10, 10)
    cases.append(("rot_trans", a, a @ R.T + [50, 50, 50], R, [50, 50, 50], TE, RE, None))
    noise = np.fromfile(os.path.join(GOLDEN, "noise3d_f64.bin")).reshape(3, 3)
    cases.append(("noise", a, a @ R.T + [20, 10, 30] + noise, R, [20, 10, 30], TE + 1.0, 0.1, None))
    return cases


def smallest_angle(R):
    return np.arctan2(R[1, 0], R[0, 0])


def is_approx(a, b, eps):
    """Eigen isApprox: ||a-b|| <= eps * min(||a||, ||b||) (Frobenius)"""
    return np.linalg.norm(a - b) <= eps * min(np.linalg.norm(a), np.linalg.norm(b))


def check_case_2d(T, iters, case):
    name, a, b, R, t, te, re_, max_it = case
    assert np.all(np.abs(T[:2, 2] - np.asarray(t, float)) < te), (name, T)
    assert abs(smallest_angle(T[:2, :2]) - smallest_angle(R)) < re_, (name, T)
    if max_it is not None:
        assert iters <= max_it, name


def check_case_3d(T, iters, case):
    name, a, b, R, t, te, re_, max_it = case
    assert np.all(np.abs(T[:3, 3] - np.asarray(t, float)) < te), (name, T)
    assert is_approx(T[:3, :3], R, re_), (name, T)
    if max_it is not None:
        assert iters <= max_it, name



def _seq_d2(x):
    """sum of squares along the last axis in axis order, as the kernels and kdtree.h:124 accumulate it"""
    x2 = x * x
    s = x2[..., 0] + x2[..., 1]
    return s + x2[..., 2] if x.shape[-1] == 3 else s


import functools  # noqa: E402


@functools.lru_cache(maxsize=None)
def _tie_clouds(dim, n_random, m_random, planted, seed):
    rng = np.random.default_rng(seed)
    q = rng.uniform(-10, 10, (n_random, dim))
    t = rng.uniform(-10, 10, (m_random, dim))
    lat = 2.0 ** -20
    pq, pt = [], []
    for k in range(planted):
        c = np.round(rng.uniform(-9, 9, dim) / lat) * lat
        r = 1e-3 * (1 + rng.random())
        if k % 2 == 0:
            while True:
                v = np.round(rng.normal(size=dim) / lat * r) * lat
                v2 = v[rng.permutation(dim)] * rng.choice([-1.0, 1.0], dim)
                if np.all(v2 != v) and np.all(v != 0):
                    break
            pq.append(c)
            pt += [c + v, c + v2]
        else:
            v = rng.normal(size=dim)
            t1 = c + v * (r / np.linalg.norm(v))
            d1 = _seq_d2(t1 - c)
            t2 = None
            for _ in range(8):
                w = rng.normal(size=(2048, dim))
                cand = c + w * (r / np.linalg.norm(w, axis=1))[:, None]
                for _ in range(3):  # pull the candidates' squared distances to within a few ulps of d1
                    cand = c + (cand - c) * np.sqrt(d1 / _seq_d2(cand - c))[:, None]
                d2 = _seq_d2(cand - c)
                ok = np.nonzero((d2 != d1) & (np.sqrt(d2) == np.sqrt(d1)))[0]
                if len(ok):
                    t2 = cand[ok[0]]
                    break
            if t2 is None:  # (not seen in practice) fall back to an exact tie
                t2 = c - (t1 - c)
            pq.append(c)
            pt += [t1, t2]
    pq, pt = np.array(pq).reshape(-1, dim), np.array(pt).reshape(-1, dim)
    # keep random targets away from the planted queries
    if len(pq):
        far = np.ones(len(t), bool)
        for c in pq:
            far &= ((t - c) ** 2).sum(axis=1) > (0.05) ** 2
        t = t[far]
    order_t = rng.permutation(len(t) + len(pt))
    order_q = rng.permutation(len(q) + len(pq))
    tt = np.concatenate([t, pt])[order_t]
    qq = np.concatenate([q, pq])[order_q]
    planted_idx = np.nonzero(order_q >= len(q))[0]
    return np.ascontiguousarray(qq), np.ascontiguousarray(tt), planted_idx


def tie_clouds(dim, n_random, m_random, planted, seed):
    """(queries, targets, planted query indices) with PLANTED kd-rule ties and, with overwhelming probability,
    pairwise distinct coordinates on every axis — the data on which the reference's kd-tree is uniquely defined
    and its first-visited pick differs from the (d2, index) rule about half of the time.  Two families per
    planted query q (nothing else is nearer than its partners):
      exact   two targets at q + v and q + v' with v' a signed permutation of v on a 2^-20 lattice: all the
              arithmetic is exact, so the two d2 are bit-equal;
      sqrt    two targets whose d2 differ by a few ulps and round to ONE square root (found by trial):
              the case of ex_data/scan1, source 109."""
    qq, tt, pi = _tie_clouds(dim, n_random, m_random, planted, seed)
    return qq.copy(), tt.copy(), pi.copy()
