"""CPU tests of the ORACLE itself: pinned against the reference's real kd-tree (oracle/_ref), the
reference's own test cases and tolerances, numpy's SVD, and the committed golden traces."""
import os

import numpy as np
import pytest

from conftest import (BENCH_CRIT, GOLDEN, PLY_CRIT, TEST_CRIT, check_case_2d, check_case_3d,
                      reference_cases_2d, reference_cases_3d, tie_clouds)


def kat_points():
    return np.fromfile(os.path.join(GOLDEN, "kdtree_kat_points_f64.bin")).reshape(1000, 3)


def test_kdtree_kat_reference_setup(ob):
    """tests/test.cpp:39-81: every point queries itself; kd-tree index == brute-force argmin."""
    pts = kat_points()
    bi, bd2 = ob.nn_brute(pts, pts)
    ki, kd = ob.nn_kdtree(pts, pts)
    assert np.array_equal(bi, np.arange(1000))
    assert np.array_equal(ki, bi)
    assert np.array_equal(kd, np.sqrt(bd2))


def test_oracle_kdtree_equals_real_reference_kdtree(ob, scans, queen):
    """The restated kd-tree against the reference's own include/algo/kdtree.h (oracle/_ref):
    same libstdc++ nth_element => identical trees, indices and distances, ties included."""
    if ob.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    pts = kat_points()
    cases = [(pts, pts)]
    for k in (1, 2, 3):
        cases.append(scans[k])
    cases.append(queen)  # 1 956 duplicated vertices: exact ties
    rng = np.random.default_rng(7)
    cases.append((rng.normal(size=(500, 2)), rng.normal(size=(700, 2))))
    cases.append((rng.integers(0, 4, (300, 3)).astype(float), rng.integers(0, 4, (400, 3)).astype(float)))
    for q, t in cases:
        oi, od = ob.nn_kdtree(q, t)
        ri, rd = ob.ref_nn_kdtree(q, t)
        assert np.array_equal(oi, ri)
        assert np.array_equal(od, rd)


def test_kdtree_vs_brute_differ_only_on_sqrt_ties(ob, scans, queen):
    """Documented tie-break: the kd-tree compares sqrt(d2) and keeps the first node
    it visits, the canonical rule is the lexicographic min of (d2, index).  The two may name
    different targets ONLY where sqrt(d2) is bit-identical (duplicate targets, or two distinct d2
    that round to one sqrt)."""
    counts = {}
    for name, (q, t) in [("scan1", scans[1]), ("scan2", scans[2]), ("scan3", scans[3]), ("queen", queen)]:
        bi, bd2 = ob.nn_brute(q, t)
        ki, kd = ob.nn_kdtree(q, t)
        assert np.array_equal(kd, np.sqrt(bd2)), name  # the distance itself always agrees
        diff = np.nonzero(bi != ki)[0]
        for i in diff:
            d2_k = ((t[ki[i]] - q[i]) ** 2).sum()
            assert np.sqrt(d2_k) == np.sqrt(bd2[i]), (name, i)
            assert bi[i] < ki[i] or d2_k > bd2[i], (name, i)
        counts[name] = len(diff)
    # scan1 holds ONE sqrt collision: source point 109, targets 152 / 158, d2 one ulp apart
    assert counts["scan1"] == 1 and counts["scan2"] == 0 and counts["scan3"] == 0


def test_canonical_kd_rule_equals_real_reference_kdtree(ob, scans, queen):
    """ORC_NN_KDCANON — the rule the HIP kernels implement for the kd-tree methods — against the reference's own
    include/algo/kdtree.h (oracle/_ref): identical indices and distances wherever the reference tree is uniquely
    defined (distinct coordinates): scans 1-3 (scan1's sqrt collision at source 109 included), the reference's own
    kd-tree test cloud, random clouds, and clouds with PLANTED ties (exact d2 ties between distinct targets and
    pairs of d2 that round to one square root), on which the (d2, index) rule differs from the tree dozens of times."""
    if ob.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    pts = kat_points()
    cases = [("kat", pts, pts)] + [(f"scan{k}", *scans[k]) for k in (1, 2, 3)]
    rng = np.random.default_rng(17)
    cases.append(("rand2", rng.normal(size=(900, 2)), rng.normal(size=(1200, 2))))
    cases.append(("rand3", rng.uniform(-3, 3, (1500, 3)), rng.uniform(-3, 3, (2500, 3))))
    brute_differs = 0
    for dim in (2, 3):
        q, t, planted = tie_clouds(dim, 700, 1300, 150, 100 + dim)
        for a in range(dim):
            assert len(np.unique(t[:, a])) == len(t)  # the reference tree is uniquely defined
        cases.append((f"planted{dim}", q, t))
        brute_differs += int((ob.nn_brute(q, t)[0] != ob.ref_nn_kdtree(q, t)[0]).sum())
    assert brute_differs > 40  # the planted ties do tell the two rules apart
    for name, q, t in cases:
        ri, rd = ob.ref_nn_kdtree(q, t)
        ci, cd = ob.nn_kdcanon(q, t)
        assert np.array_equal(ci, ri), (name, np.nonzero(ci != ri)[0][:5])
        assert np.array_equal(cd, rd), name
    # queen.ply: 1 956 duplicated vertices.  Which of coincident targets the reference returns is its nth_element's
    # choice; the canonical rule returns the lowest index among them: same coordinates, same distances
    q, t = queen
    ri, rd = ob.ref_nn_kdtree(q, t)
    ci, cd = ob.nn_kdcanon(q, t)
    assert np.array_equal(cd, rd) and np.array_equal(t[ci], t[ri])


def test_canonical_kd_rule_two_formulations(ob, scans, queen):
    """The tree walk (kdtree.h:132-154 on the canonical tree) and the brute-force formulation of the same rule (minimum,
    tied set by sqrt(d2), smallest pre-order key, lowest coincident index — what the kernels' tie resolution
    computes): identical answers, also where ties abound (lattices, duplicated vertices, non-finite points)."""
    rng = np.random.default_rng(5)
    cases = [scans[1], (queen[0][:1500], queen[1])]
    cases.append((rng.integers(0, 6, (400, 2)).astype(float), rng.integers(0, 6, (900, 2)).astype(float)))
    cases.append((rng.integers(0, 4, (300, 3)).astype(float) + 0.5, rng.integers(0, 4, (1100, 3)).astype(float)))
    for dim in (2, 3):
        q, t, _ = tie_clouds(dim, 300, 700, 80, 200 + dim)
        cases.append((q, t))
    cases.append((np.array([[1.0, 2.0, 3.0]]), np.array([[4.0, 5.0, 6.0]])))
    for q, t in cases:
        wi, wd = ob.nn_kdcanon(q, t)
        ki, kd = ob.nn_kdcanon(q, t, by_key=True)
        assert np.array_equal(wi, ki) and np.array_equal(wd, kd)
        bi, bd2 = ob.nn_brute(q, t)
        assert np.array_equal(wd, np.sqrt(bd2))  # the distance is the minimum's under either rule


def test_brute_lowest_index_on_duplicates(ob):
    t = np.array([[1., 1.], [0., 0.], [0., 0.], [2., 2.]])
    q = np.array([[0.1, 0.], [5., 5.]])
    idx, d2 = ob.nn_brute(q, t)
    assert idx.tolist() == [1, 3]
    assert d2[0] == 0.1 * 0.1 + 0.0


def test_jacobi_svd_against_numpy(ob):
    rng = np.random.default_rng(0)
    for dim in (2, 3):
        for trial in range(200):
            A = rng.normal(size=(dim, dim)) * 10.0 ** rng.integers(-3, 4)
            if trial % 10 == 0:
                A[:, 0] = A[:, 1]  # rank deficient
            U, S, V = ob.jacobi_svd(A)
            assert np.allclose(U @ np.diag(S) @ V.T, A, atol=1e-12 * max(1.0, np.abs(A).max()))
            assert np.allclose(U.T @ U, np.eye(dim), atol=1e-13)
            assert np.allclose(V.T @ V, np.eye(dim), atol=1e-13)
            assert np.all(S >= 0) and np.all(np.diff(S) <= 0)
            assert np.allclose(S, np.linalg.svd(A, compute_uv=False), rtol=1e-12, atol=1e-14 * np.abs(A).max())


def test_jacobi_svd_zero_matrix_gives_identity(ob):
    """Q12: N = 0 (single point) => U = V = I, so R = I (tests/test.cpp:91-102)."""
    for dim in (2, 3):
        U, S, V = ob.jacobi_svd(np.zeros((dim, dim)))
        assert np.array_equal(U, np.eye(dim)) and np.array_equal(V, np.eye(dim))
        assert np.all(S == 0)


def test_kabsch_recovers_rotation_and_fixes_reflection(ob):
    rng = np.random.default_rng(1)
    for dim in (2, 3):
        A = rng.normal(size=(50, dim))
        Q, _ = np.linalg.qr(rng.normal(size=(dim, dim)))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        t = rng.normal(size=dim)
        B = A @ Q.T + t
        ca, cb = A.mean(0), B.mean(0)
        N = (A - ca).T @ (B - cb)
        step = ob.kabsch(N, ca, cb)
        assert np.allclose(step[:dim, :dim], Q, atol=1e-12)
        assert np.allclose(step[:dim, dim], t, atol=1e-12)
        # a mirrored cloud must still give a proper rotation (vanilla.cpp:56-59)
        M = np.eye(dim)
        M[0, 0] = -1
        Bm = A @ M.T
        Nm = (A - ca).T @ (Bm - Bm.mean(0))
        stepm = ob.kabsch(Nm, ca, Bm.mean(0))
        assert np.linalg.det(stepm[:dim, :dim]) > 0.999


def test_tree64_sum_matches_sequential_closely(ob):
    rng = np.random.default_rng(2)
    for n in (0, 1, 63, 64, 65, 4095, 4096, 4097, 100000):
        v = rng.normal(size=n) * 1e3
        s0, s1 = ob.osum(v, ob.SUM_SEQUENTIAL), ob.osum(v, ob.SUM_TREE64)
        assert abs(s0 - s1) <= 1e-9 * max(1.0, np.abs(v).sum())
    # exact on integers
    v = np.arange(1000, dtype=float)
    assert ob.osum(v, ob.SUM_TREE64) == 499500.0


@pytest.mark.parametrize("method,overlap", [("vanilla", 0.9), ("trimmed", 1.0)])
@pytest.mark.parametrize("modes", ["ref", "dev"])
def test_reference_2d_cases_on_oracle(ob, method, overlap, modes):
    nn, sm = (ob.NN_KDTREE, ob.SUM_SEQUENTIAL) if modes == "ref" else (ob.NN_KDCANON, ob.SUM_TREE64)
    for case in reference_cases_2d():
        o = ob.OracleICP(2, method, overlap_rate=overlap, nn_mode=nn, sum_mode=sm)
        r = o.converge(case[1], case[2], None, ob.make_criteria(**TEST_CRIT))
        check_case_2d(r.transform, r.iteration_count, case)


@pytest.mark.parametrize("modes", ["ref", "dev"])
def test_reference_2d_cases_feature_aware_on_oracle(ob, modes):
    """tests/test.cpp:228-232: the same six cases through feature_aware with overlap_rate 1.0,
    feature_weight 0.7, symmetric_neighbors 1."""
    sm = ob.SUM_SEQUENTIAL if modes == "ref" else ob.SUM_TREE64
    for case in reference_cases_2d():
        o = ob.OracleICP(2, "feature_aware", overlap_rate=1.0, feature_weight=0.7, symmetric_neighbors=1,
                         nn_mode=ob.NN_BRUTE, sum_mode=sm)
        r = o.converge(case[1], case[2], None, ob.make_criteria(**TEST_CRIT))
        check_case_2d(r.transform, r.iteration_count, case)


def test_feature_aware_features_against_numpy(ob, scans):
    """compute_features / compute_matches (feature_aware.cpp:129-177) restated independently in numpy:
    the first correspondence step of feature_aware on scan 3 (indices equal, costs to 1e-12)."""
    a, b = scans[3]
    sn, fw = 10, 0.7

    def feats(p):
        c = p.mean(axis=0)
        rad = np.linalg.norm(p - c, axis=1)
        f = np.zeros((len(p), 2 * sn))
        for i in range(len(p)):
            lo = max(0, i - sn)
            f[i, : i - lo] = rad[lo:i] - rad[i]
            hi = min(len(p) - 1, i + sn)
            f[i, sn: sn + hi - i] = rad[i + 1: hi + 1] - rad[i]
        return f

    fa, fb = feats(a), feats(b)
    nfd = np.sqrt(((fa[:, None, :] - fb[None, :, :]) ** 2).sum(axis=2))
    nfd /= nfd.max()
    nd = np.sqrt(((a[:, None, :] - b[None, :, :]) ** 2).sum(axis=2))
    nd /= nd.max()
    cost = (1 - fw) * nd + fw * nfd
    o = ob.OracleICP(2, "feature_aware")
    o.begin(a, b)
    point, pair, c = o.get_matches()
    assert np.array_equal(point, np.arange(len(a)))
    assert np.array_equal(pair, cost.argmin(axis=1))
    assert np.allclose(c, cost.min(axis=1), rtol=0, atol=1e-12)
    assert abs(o.calculate_cost() - np.sqrt(cost.min(axis=1).sum() / len(a))) < 1e-12


@pytest.mark.parametrize("modes", ["ref", "dev"])
def test_reference_3d_cases_on_oracle(ob, modes):
    nn, sm = (ob.NN_KDTREE, ob.SUM_SEQUENTIAL) if modes == "ref" else (ob.NN_KDCANON, ob.SUM_TREE64)
    for case in reference_cases_3d():
        o = ob.OracleICP(3, "vanilla", nn_mode=nn, sum_mode=sm)
        r = o.converge(case[1], case[2], None, ob.make_criteria(**TEST_CRIT))
        check_case_3d(r.transform, r.iteration_count, case)


def test_golden_traces_reproduce(ob, scans, queen, traces):
    """The committed traces (tests/tools/make_golden.py) pin the oracle; SURVEY §8c sanity anchors too."""
    for key, g in traces.items():
        data, method, mode = key.split("/")
        nn, sm = (ob.NN_KDTREE, ob.SUM_SEQUENTIAL) if mode == "ref" else (ob.NN_KDCANON, ob.SUM_TREE64)
        if data == "queen":
            src, dst = queen
            dim, crit = 3, PLY_CRIT
        else:
            src, dst = scans[int(data[-1])]
            dim, crit = 2, BENCH_CRIT
        o = ob.OracleICP(dim, method, nn_mode=nn, sum_mode=sm)
        r = o.converge(src, dst, None, ob.make_criteria(**crit), trace_cap=128)
        assert r.iteration_count == g["iterations"], key
        assert np.allclose(r.transform, g["transform"], rtol=0, atol=1e-12), key
        assert np.allclose(r.trace_cost, g["trace_cost"], rtol=1e-12, atol=0, equal_nan=True), key
    # anchors from the survey's independent numpy probe
    its = [traces[f"scan{k}/trimmed/ref"]["iterations"] for k in (1, 2, 3)]
    assert its == [22, 3, 12]
    assert all(traces[f"scan{k}/vanilla/ref"]["iterations"] == 2 for k in (1, 2, 3))
    ang = [np.degrees(np.arctan2(traces[f"scan{k}/vanilla/ref"]["transform"][1, 0],
                                 traces[f"scan{k}/vanilla/ref"]["transform"][0, 0])) for k in (1, 2, 3)]
    assert np.allclose(ang, [-4.468, 2.955, -0.059], atol=2e-3)
    assert traces["queen/vanilla/ref"]["iterations"] == 12


def test_reference_order_and_device_order_agree_to_1e6(ob, scans, queen, traces):
    """north_star tolerance: final transforms within 1e-6 rad / 1e-6 m between the reference-order restatement
    (real kd-tree walk, sequential sums) and the device order (the kd rule in canonical form, TREE64 sums) — for
    every data set and method, scan1 / vanilla with its sqrt collision included."""
    for key, g in traces.items():
        if not key.endswith("/ref"):
            continue
        d = traces[key[:-3] + "dev"]
        assert g["iterations"] == d["iterations"], key
        assert np.abs(g["transform"] - d["transform"]).max() < 1e-6, key


def test_quirks(ob, scans):
    a, b = scans[3]
    # Q1: vanilla 2D matches never change
    o = ob.OracleICP(2, "vanilla")
    o.begin(a, b)
    p0 = o.get_matches()[1].copy()
    o.iterate()
    o.iterate()
    assert np.array_equal(o.get_matches()[1], p0)
    # Q7: trimmed leaves matches sorted by cost
    o = ob.OracleICP(2, "trimmed")
    o.begin(a, b)
    o.iterate()
    c = o.get_matches()[2]
    assert np.all(np.diff(c) >= 0)
    # Q4: 3D never fills matches -> cost 0
    rng = np.random.default_rng(3)
    x = rng.normal(size=(20, 3))
    o = ob.OracleICP(3, "vanilla")
    o.begin(x, x + 0.1)
    o.iterate()
    assert o.calculate_cost() == 0.0
    assert np.all(o.get_matches()[2] == 0)
    # Q5: the 3D running cloud ignores the guess
    g = np.eye(4)
    g[:3, 3] = [5, 5, 5]
    o.begin(x, x, g)
    assert np.array_equal(o.current_cloud(), x)
    o.iterate()
    assert np.allclose(o.current_transform(), g, atol=1e-12)


def test_driver_criteria_on_oracle(ob, scans):
    a, b = scans[1]
    o = ob.OracleICP(2, "trimmed")
    r = o.converge(a, b, None, ob.make_criteria(max_iterations=5))
    assert r.iteration_count == 5
    r = o.converge(a, b, None, ob.make_criteria(max_iterations=0))
    assert r.iteration_count == 0
    r = o.converge(a, b, None, ob.make_criteria(min_iterations=7, max_iterations=50, stop_cost=1e9))
    assert r.iteration_count == 7
    r = o.converge(a, b, None, ob.make_criteria(max_iterations=50, absolute=1e-3))
    r2 = o.converge(a, b, None, ob.make_criteria(max_iterations=50, relative=1e-2))
    assert 1 <= r.iteration_count < 50 and 1 <= r2.iteration_count < 50
