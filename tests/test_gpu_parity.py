"""GPU parity tests: the HIP path, called through the C ABI (icp_amd._lib / icp_amd.api), against
the CPU oracle on the same inputs.

Bars (tier ③): correspondence indices and squared distances BIT-EXACT against the canonical
(d2, index) rule; whole converge() runs bit-exact against the oracle in device summation order
(TREE64) — transforms, costs, iteration counts, matches; and within 1e-6 rad / 1e-6 m (north_star)
against the oracle in the reference's summation order.
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import (BENCH_CRIT, GOLDEN, PLY_CRIT, TEST_CRIT, check_case_2d, check_case_3d,
                      reference_cases_2d, reference_cases_3d, tie_clouds)

pytestmark = pytest.mark.gpu

TOL = 1e-6  # north_star: 1e-6 rad and 1e-6 m


@pytest.fixture(scope="module")
def icp():
    import icp_amd
    icp_amd._lib.check(icp_amd._lib.load().icpx_init(0))
    return icp_amd


def dev_oracle(ob, dim, method, **kw):
    return ob.OracleICP(dim, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64, **kw)


def driver(icp, cls, method, crit, config=None, **kw):
    d = icp.ICPDriver(cls.from_method(method, config, **kw))
    if "max_iterations" in crit:
        d.set_max_iterations(crit["max_iterations"])
    if "min_iterations" in crit:
        d.set_min_iterations(crit["min_iterations"])
    if "transform_tol" in crit:
        d.set_transform_tolerance(*crit["transform_tol"])
    if "stop_cost" in crit:
        d.set_stop_cost(crit["stop_cost"])
    if "absolute" in crit:
        d.set_absolute_cost_tolerance(crit["absolute"])
    if "relative" in crit:
        d.set_relative_cost_tolerance(crit["relative"])
    return d


# ------------------------------------------------------------------------------------------
# per-call NN parity
# ------------------------------------------------------------------------------------------

def nn_cases(scans, queen):
    rng = np.random.default_rng(11)
    cases = [("scan1", *scans[1]), ("scan2", *scans[2]), ("scan3", *scans[3]), ("queen", *queen)]
    cases.append(("rand2d", rng.uniform(-10, 10, (3000, 2)), rng.uniform(-10, 10, (2500, 2))))
    cases.append(("rand3d", rng.uniform(-10, 10, (5000, 3)), rng.uniform(-10, 10, (7001, 3))))
    # exact ties: integer lattice with many duplicates
    cases.append(("lattice2d", rng.integers(0, 6, (700, 2)).astype(float), rng.integers(0, 6, (900, 2)).astype(float)))
    cases.append(("lattice3d", rng.integers(0, 4, (600, 3)).astype(float), rng.integers(0, 4, (1100, 3)).astype(float)))
    # ragged / tiny
    cases.append(("one_one", np.array([[1., 2., 3.]]), np.array([[4., 5., 6.]])))
    cases.append(("one_target", rng.normal(size=(130, 2)), np.array([[0.5, 0.5]])))
    cases.append(("one_query", np.array([[0.1, 0.2]]), rng.normal(size=(1025, 2))))
    # queries far outside the target's bounding box, clustered targets, degenerate extents
    cases.append(("outside", rng.uniform(50, 60, (400, 3)), rng.uniform(-1, 1, (3000, 3))))
    clus = np.concatenate([rng.normal(c, 0.01, (800, 3)) for c in ([0, 0, 0], [5, 5, 5], [-7, 3, 1])])
    cases.append(("clustered", rng.uniform(-8, 8, (1500, 3)), clus))
    line = np.stack([np.linspace(-3, 3, 2000), np.zeros(2000), np.full(2000, 2.0)], axis=1)
    cases.append(("collinear", rng.uniform(-4, 4, (900, 3)), line))
    plane = np.concatenate([rng.uniform(-5, 5, (4000, 2)), np.zeros((4000, 1))], axis=1)
    cases.append(("planar", rng.uniform(-5, 5, (1000, 3)), plane))
    cases.append(("same_point", rng.normal(size=(64, 2)), np.tile([[1.0, -1.0]], (500, 1))))
    return cases


@pytest.mark.parametrize("mode", ["brute", "grid"])
def test_nn_bit_exact(icp, ob, scans, queen, mode):
    m = icp.NN_BRUTE if mode == "brute" else icp.NN_GRID
    for name, q, t in nn_cases(scans, queen):
        idx, d2 = icp.nearest_neighbors(q, t, m)
        oi, od2 = ob.nn_brute(q, t)
        assert np.array_equal(idx, oi), (name, mode, np.nonzero(idx != oi)[0][:5])
        assert np.array_equal(d2, od2), (name, mode)


def test_nn_empty_and_nonfinite(icp, ob):
    q = np.array([[0.0, 0.0], [np.nan, 1.0], [np.inf, 0.0], [1.0, 1.0]])
    t = np.array([[1.0, 1.0], [np.nan, 0.0], [0.25, 0.0], [np.inf, np.inf]])
    for m in (icp.NN_BRUTE, icp.NN_GRID):
        idx, d2 = icp.nearest_neighbors(q, t, m)
        oi, od2 = ob.nn_brute(q, t)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od2)
        idx, d2 = icp.nearest_neighbors(q, np.zeros((0, 2)), m)  # no targets: (0, inf)
        assert np.all(idx == 0) and np.all(np.isinf(d2))
        idx, d2 = icp.nearest_neighbors(np.zeros((0, 2)), t, m)
        assert len(idx) == 0


@pytest.mark.parametrize("mode", ["auto", "brute", "grid", "bvh"])
def test_nn_against_reference_kdtree_semantics(icp, ob, scans, queen, mode):
    """icpx_nn_kdtree against the REAL reference kd-tree (oracle/_ref when present, else its restatement): ZERO
    differing indices and bit-equal distances on scans 1-3 (source 109 of scan1 — two targets whose d2 are one ulp
    apart and share a square root — included), on random clouds and on clouds with planted ties, all with distinct
    coordinates (where the reference's tree is uniquely defined); through every search engine."""
    m = {"auto": icp.NN_AUTO, "brute": icp.NN_BRUTE, "grid": icp.NN_GRID, "bvh": icp.NN_BVH}[mode]
    rng = np.random.default_rng(23)
    cases = [(f"scan{k}", *scans[k]) for k in (1, 2, 3)]
    cases.append(("rand2", rng.uniform(-10, 10, (3000, 2)), rng.uniform(-10, 10, (2500, 2))))
    cases.append(("rand3", rng.uniform(-10, 10, (5000, 3)), rng.uniform(-10, 10, (7001, 3))))
    for dim in (2, 3):
        q, t, _ = tie_clouds(dim, 2500, 4000, 300, 300 + dim)
        cases.append((f"planted{dim}", q, t))
    for name, q, t in cases:
        ki, kd = ob.ref_nn_kdtree(q, t) if ob.ref_lib() is not None else ob.nn_kdtree(q, t)
        idx, dist = icp.nearest_neighbors_kdtree(q, t, m)
        assert np.array_equal(dist, kd), (name, mode)
        assert np.array_equal(idx, ki), (name, mode, np.nonzero(idx != ki)[0][:8])
    # scan1, source 109: the tree's pick (158), not the (d2, index) minimum (152)
    idx, _ = icp.nearest_neighbors_kdtree(*scans[1], m)
    bi, _ = icp.nearest_neighbors(*scans[1], m)
    assert idx[109] == 158 and bi[109] == 152
    # coincident targets and lattices (the reference's tree is its nth_element's choice there): the canonical form
    q, t = queen
    ci, cd = ob.nn_kdcanon(q, t)
    idx, dist = icp.nearest_neighbors_kdtree(q, t, m)
    assert np.array_equal(idx, ci) and np.array_equal(dist, cd), mode
    lat = rng.integers(0, 5, (900, 3)).astype(float)
    ci, cd = ob.nn_kdcanon(lat[:400] + 0.5, lat)
    idx, dist = icp.nearest_neighbors_kdtree(lat[:400] + 0.5, lat, m)
    assert np.array_equal(idx, ci) and np.array_equal(dist, cd), mode
    # no targets / non-finite points: the (d2, index) conventions
    idx, dist = icp.nearest_neighbors_kdtree(np.zeros((3, 2)), np.zeros((0, 2)), m)
    assert np.all(idx == 0) and np.all(np.isinf(dist))


@pytest.mark.parametrize("path", ["fused", "large_auto", "large_grid", "large_bvh", "large_grid_sorted"])
def test_kd_tie_rule_in_converge(icp, ob, path, knob):
    """The kd-tree methods follow the reference tree's tie rule inside begin / iterate / converge: clouds with planted
    ties (distinct coordinates), every iteration's neighbours, matches and transform against the oracle in canonical
    form — whose correspondences equal the real kdtree.h's on such data (test_oracle).  fused: one workgroup per
    pair (2D vanilla resolves in place; the 3D kernels stop at a tie and the take-over kernel goes on);
    large_*: the multi-kernel path (tree built on the device beside the first step, resolve kernel) with each
    correspondence engine; large_grid_sorted: enough points for the 3D grid search to run in its sorted domain and a
    target cloud beyond the eager tree build (stall at the first tie, tree on demand)."""
    if path == "fused":
        sizes, cfg_nn = (500, 900, 120), None
    elif path == "large_grid_sorted":
        sizes, cfg_nn = (40000, 70000, 300), "grid"
    else:
        sizes, cfg_nn = (5000, 6000, 200), path.split("_")[1]
    # 2D vanilla: matches of begin (they never change, vanilla.cpp:80)
    q, t, planted = tie_clouds(2, *sizes, 7)
    cfg = icp.Config()
    if cfg_nn:
        cfg.set("nn", cfg_nn)
    x = icp.ICP2.from_method("vanilla", cfg)
    x.begin(q, t)
    p, pair, c = x.get_matches_arrays()
    ci, cd = ob.nn_kdcanon(q, t)
    assert np.array_equal(pair, ci), np.nonzero(pair != ci)[0][:8]
    assert np.array_equal(c, np.sqrt(cd))
    if ob.ref_lib() is not None:
        assert np.array_equal(pair, ob.ref_nn_kdtree(q, t)[0])
    assert (ob.nn_brute(q, t)[0] != ci).sum() > 20  # the ties matter
    crit = dict(max_iterations=20, transform_tol=(1e-6, 1e-6))
    r = driver(icp, icp.ICP2, "vanilla", crit, cfg).converge(q, t)
    ro = dev_oracle(ob, 2, "vanilla").converge(q, t, None, ob.make_criteria(**crit))
    assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform) and r.cost == ro.cost
    # both 3D methods: stepwise and whole converge
    for method in ("vanilla", "trimmed"):
        q, t, planted = tie_clouds(3, *sizes, 9)
        cfg.set("max_distance", 0.5)
        x = icp.ICP3.from_method(method, cfg)
        o = dev_oracle(ob, 3, method, max_distance=0.5)
        x.begin(q, t)
        o.begin(q, t)
        for it in range(3):
            x.iterate()
            o.iterate()
            xi, xd = x.debug_neighbors()
            oi, od = o.neighbors()
            assert np.array_equal(xi, oi) and np.array_equal(xd, od), (method, it, np.nonzero(xi != oi)[0][:8])
            assert np.array_equal(x.current_transform(), o.current_transform()), (method, it)
        crit = dict(max_iterations=12, transform_tol=(1e-7, 1e-7))
        r = driver(icp, icp.ICP3, method, crit, cfg).converge(q, t)
        ro = o.converge(q, t, None, ob.make_criteria(**crit))
        assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform), method


@pytest.mark.parametrize("dim", [2, 3])
def test_device_built_kd_tree_equals_oracle_tree(icp, ob, dim):
    """The median-split tree (kdtree.h:96-116; equal coordinates by index) the device builds for the kd rule — one
    workgroup in LDS up to 4096 points, device-wide sorts + level passes + one workgroup per subtree beyond — against
    the oracle's KdCanon, node for node: sizes around every switch of the build, clouds with repeated coordinates
    (a lattice: every split axis has many equal values) and queen.ply (1 956 duplicate vertices)."""
    from icp_amd import api
    rng = np.random.default_rng(77 + dim)
    one_query = np.zeros((1, dim))
    clouds = [rng.normal(size=(m, dim)) for m in (1, 2, 3, 7, 100, 4095, 4096, 4097, 8192, 8193, 16385, 70001, 300007)]
    clouds.append(rng.integers(0, 12, size=(9000, dim)).astype(np.float64))      # heavy repeats on every axis
    clouds.append(np.round(rng.normal(size=(20000, dim)), 1))
    if dim == 3:
        from icp_amd import io as iio
        clouds.append(iio.read_ply_vertices(os.path.join(GOLDEN, "ex_data", "ply", "queen.ply")))
    for t in clouds:
        want = ob.nn_kdcanon(one_query, t, want_nodes=True)[2]
        got = api.debug_kd_tree(t, device=0)
        assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), (dim, len(t), np.nonzero(got != want)[0][:8])


def test_kd_tie_rule_in_batches(icp, ob):
    """icpx_converge_batch (grid-pruned kernels detect, the take-over kernel resolves): pairs with planted ties next to
    pairs without, both 3D methods and 2D vanilla, against the oracle in canonical form."""
    from icp_amd import synth
    for dim, method in ((3, "vanilla"), (3, "trimmed"), (2, "vanilla")):
        srcs, dsts = [], []
        for k in range(10):
            if k % 3 == 0:
                s, t, _, _ = synth.config5_pairs([100 + k], dim, 700)
            else:
                s, t, _ = tie_clouds(dim, 600, 800, 40, 30 + k)
            srcs.append(s)
            dsts.append(t)
        so = np.concatenate([[0], np.cumsum([len(s) for s in srcs])])
        to = np.concatenate([[0], np.cumsum([len(t) for t in dsts])])
        crit = dict(max_iterations=12, transform_tol=(1e-7, 1e-7))
        cfg = icp.Config()
        cfg.set("max_distance", 0.5)
        cost, its, T = icp.converge_batch(dim, method, np.concatenate(srcs), so, np.concatenate(dsts), to,
                                          icp.make_criteria(**crit), cfg)
        for k in range(10):
            ro = dev_oracle(ob, dim, method, max_distance=0.5).converge(srcs[k], dsts[k], None, ob.make_criteria(**crit))
            assert its[k] == ro.iteration_count and np.array_equal(T[k], ro.transform), (dim, method, k)
            assert cost[k] == ro.cost


# ------------------------------------------------------------------------------------------
# whole converge() runs
# ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("prune_handles", ["0", "1"])
@pytest.mark.parametrize("method", ["vanilla", "trimmed"])
@pytest.mark.parametrize("k", [1, 2, 3])
def test_scans_converge_bit_exact_and_golden(icp, ob, scans, traces, method, k, prune_handles, knob):
    knob.set("ICPX_PRUNE_HANDLES", prune_handles)  # "1": handles through the grid-pruned kernels
    a, b = scans[k]
    d = driver(icp, icp.ICP2, method, BENCH_CRIT)
    r = d.converge(a, b)
    o = dev_oracle(ob, 2, method)
    ro = o.converge(a, b, None, ob.make_criteria(**BENCH_CRIT))
    assert r.iteration_count == ro.iteration_count
    assert np.array_equal(r.transform, ro.transform)
    assert r.cost == ro.cost
    p, q, c = d.icp_.get_matches_arrays()
    po, qo, co = o.get_matches()
    assert np.array_equal(p, po) and np.array_equal(q, qo) and np.array_equal(c, co)
    # committed golden vectors (device order: exact; reference order: north_star tolerance)
    g = traces[f"scan{k}/{method}/dev"]
    assert r.iteration_count == g["iterations"] and np.array_equal(r.transform, g["transform"])
    gr = traces[f"scan{k}/{method}/ref"]
    assert r.iteration_count == gr["iterations"]
    assert np.abs(r.transform - gr["transform"]).max() < TOL  # north_star, every (scan, method)
    # reference summation order and the real kd-tree walk (ORC_NN_KDTREE): within 1e-6 too
    rs = ob.OracleICP(2, method, nn_mode=ob.NN_KDTREE).converge(a, b, None, ob.make_criteria(**BENCH_CRIT))
    assert rs.iteration_count == r.iteration_count
    assert np.abs(rs.transform - r.transform).max() < TOL
    assert abs(rs.cost - r.cost) < 1e-12


@pytest.mark.parametrize("nn", ["auto", "grid", "bvh"])
@pytest.mark.parametrize("method", ["vanilla", "trimmed"])
def test_queen_large_path(icp, ob, queen, traces, method, nn):
    """BASELINE config 3 (10 720 points) on the multi-kernel path: tiled brute-force NN (auto at this
    size), the uniform-grid NN and the BVH (forced), multi-kernel reductions."""
    src, dst = queen
    cfg = icp.Config()
    cfg.set("nn", nn)
    d = driver(icp, icp.ICP3, method, PLY_CRIT, cfg)
    r = d.converge(src, dst)
    g = traces[f"queen/{method}/dev"]
    assert r.iteration_count == g["iterations"] == 12
    assert np.array_equal(r.transform, g["transform"])
    assert np.abs(r.transform - traces[f"queen/{method}/ref"]["transform"]).max() < TOL
    assert r.cost == 0.0  # Q4
    p, q, c = d.icp_.get_matches_arrays()
    assert not p.any() and not q.any() and not c.any()
    # neighbours of the last iteration against the oracle's
    o = dev_oracle(ob, 3, method)
    o.converge(src, dst, None, ob.make_criteria(**PLY_CRIT))
    oi, od = o.neighbors()
    idx, dist = d.icp_.debug_neighbors()
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


@pytest.mark.parametrize("prune_handles", ["0", "1"])
@pytest.mark.parametrize("dim,method", [(2, "vanilla"), (2, "trimmed"), (3, "vanilla"), (3, "trimmed")])
def test_stepwise_begin_iterate_matches_oracle_every_iteration(icp, ob, scans, dim, method, prune_handles, knob):
    knob.set("ICPX_PRUNE_HANDLES", prune_handles)  # "1": handles through the grid-pruned kernels
    if dim == 2:
        a, b = scans[3]
        cfg = None
        okw = {}
    else:
        from icp_amd import synth
        a, b, _, _ = synth.config5_pairs([5], 3, 700)
        cfg = icp.Config()
        cfg.set("max_distance", 0.35)
        okw = {"max_distance": 0.35}
    cls = icp.ICP2 if dim == 2 else icp.ICP3
    g = np.eye(dim + 1)
    g[:dim, dim] = 0.05
    x = cls.from_method(method, cfg)
    o = dev_oracle(ob, dim, method, **okw)
    x.begin(a, b, g)
    o.begin(a, b, g)
    assert x.calculate_cost() == o.calculate_cost()
    assert np.array_equal(x.current_transform(), o.current_transform())
    for it in range(6):
        x.iterate()
        o.iterate()
        assert np.array_equal(x.current_transform(), o.current_transform()), it
        assert x.calculate_cost() == o.calculate_cost(), it
        if dim == 2:
            for u, v in zip(x.get_matches_arrays(), o.get_matches()):
                assert np.array_equal(u, v), it
        else:
            for u, v in zip(x.debug_neighbors(), o.neighbors()):
                assert np.array_equal(u, v), it


@pytest.mark.parametrize("single_max", ["1e18", None])
def test_fused_sizes_and_overlap_variants(icp, ob, knob, single_max):
    """Every fused-kernel geometry (queries per lane 1/2/4/8) in 2D and 3D, ragged sizes — with the
    single-pair size threshold lifted (every size in the fused kernel) and at its default (sizes above
    ~1024 x 1024 take the multi-kernel path)."""
    from icp_amd import synth
    if single_max:
        knob.set("ICPX_FUSED_SINGLE_MAX", single_max)
    crit = dict(max_iterations=12, transform_tol=(1e-5, 1e-5))
    for dim, n, m in [(2, 1, 1), (2, 63, 65), (2, 200, 300), (2, 500, 400), (2, 1000, 1024), (2, 2000, 1500),
                      (2, 4096, 4096), (3, 100, 90), (3, 700, 800), (3, 1500, 1400), (3, 4096, 4096)]:
        s, t, _, _ = synth.config5_pairs([n + dim], dim, max(n, m))
        s, t = s[:n], t[:m]
        for method, ov in (("vanilla", 0.9), ("trimmed", 0.9), ("trimmed", 0.5)):
            if dim == 3 and ov == 0.5:
                continue
            cfg = icp.Config()
            cfg.set("overlap_rate", ov)
            cfg.set("max_distance", 0.3)
            cls = icp.ICP2 if dim == 2 else icp.ICP3
            r = driver(icp, cls, method, crit, cfg).converge(s, t)
            ro = dev_oracle(ob, dim, method, overlap_rate=ov, max_distance=0.3).converge(
                s, t, None, ob.make_criteria(**crit))
            assert r.iteration_count == ro.iteration_count, (dim, n, m, method, ov)
            assert np.array_equal(r.transform, ro.transform, equal_nan=True), (dim, n, m, method, ov)
            assert r.cost == ro.cost or (np.isnan(r.cost) and np.isnan(ro.cost))


def test_grid_forced_on_small_clouds_uses_large_path(icp, ob, scans):
    """nn = grid routes scan-sized clouds through the multi-kernel path: same bits."""
    a, b = scans[2]
    cfg = icp.Config()
    cfg.set("nn", "grid")
    d = driver(icp, icp.ICP2, "vanilla", BENCH_CRIT, cfg)
    r = d.converge(a, b)
    o = dev_oracle(ob, 2, "vanilla")
    ro = o.converge(a, b, None, ob.make_criteria(**BENCH_CRIT))
    assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform)
    assert r.cost == ro.cost
    for u, v in zip(d.icp_.get_matches_arrays(), o.get_matches()):
        assert np.array_equal(u, v)
    from icp_amd import synth
    s, t, _, _ = synth.config5_pairs([77], 3, 900)
    for method in ("vanilla", "trimmed"):
        cfg = icp.Config()
        cfg.set("nn", "grid")
        cfg.set("max_distance", 0.4)
        crit = dict(max_iterations=15, transform_tol=(1e-6, 1e-6))
        r = driver(icp, icp.ICP3, method, crit, cfg).converge(s, t)
        ro = dev_oracle(ob, 3, method, max_distance=0.4).converge(s, t, None, ob.make_criteria(**crit))
        assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform)


@pytest.mark.parametrize("overlap", [0.9, 0.5, 1.0, 1e-9])
def test_trimmed2d_large_path(icp, ob, scans, overlap):
    """Trimmed ICP2 beyond the fused kernel's limits (and nn = grid on scan-sized clouds): NN of
    transform*a, radix-sorted (cost, point) selection, masked TREE64 sums — every iteration's
    transform, cost and sorted matches equal the oracle's bits."""
    from icp_amd import synth
    cases = [("scan1_grid", *scans[1], "grid")]
    s, t, _, _ = synth.config5_pairs([4242], 2, 9000)
    cases.append(("n9000", s[:9000], t[:7777], None))
    # exact cost ties: duplicated source points and a lattice target
    rng = np.random.default_rng(3)
    lat = rng.integers(0, 40, (6000, 2)).astype(float)
    cases.append(("ties", np.concatenate([lat[:3000], lat[:3000]]) + 0.25, lat, None))
    for name, a, b, nn in cases:
        cfg = icp.Config()
        cfg.set("overlap_rate", overlap)
        if nn:
            cfg.set("nn", nn)
        x = icp.ICP2.from_method("trimmed", cfg)
        o = dev_oracle(ob, 2, "trimmed", overlap_rate=overlap)
        g = np.eye(3)
        g[:2, 2] = (0.03, -0.02)
        x.begin(a, b, g)
        o.begin(a, b, g)
        assert x.calculate_cost() == o.calculate_cost(), name
        for u, v in zip(x.get_matches_arrays(), o.get_matches()):
            assert np.array_equal(u, v), name
        for it in range(4):
            x.iterate()
            o.iterate()
            assert np.array_equal(x.current_transform(), o.current_transform()), (name, it)
            assert x.calculate_cost() == o.calculate_cost(), (name, it)
            for u, v in zip(x.get_matches_arrays(), o.get_matches()):
                assert np.array_equal(u, v), (name, it)
        # whole converge through the driver, device-side termination
        crit = dict(max_iterations=20, transform_tol=(1e-5, 1e-5))
        d = driver(icp, icp.ICP2, "trimmed", crit, cfg)
        r = d.converge(a, b)
        ro = o.converge(a, b, None, ob.make_criteria(**crit))
        assert r.iteration_count == ro.iteration_count, name
        assert np.array_equal(r.transform, ro.transform) and r.cost == ro.cost, name
        for u, v in zip(d.icp_.get_matches_arrays(), o.get_matches()):
            assert np.array_equal(u, v), name


@pytest.mark.parametrize("nn", ["auto", "grid"])
def test_large_path_sizes_and_engines(icp, ob, nn):
    """Clouds beyond the fused kernel (n > 4096 or target not in LDS), all four methods, both
    correspondence engines of the multi-kernel path; ragged sizes around the tile / split limits."""
    from icp_amd import synth
    crit = dict(max_iterations=6, transform_tol=(1e-7, 1e-7))
    for dim, n, m in ((2, 5000, 4100), (2, 4097, 9000), (3, 4500, 1023), (3, 1030, 7001), (3, 9000, 9001)):
        s, t, _, _ = synth.config5_pairs([n + m], dim, max(n, m))
        s, t = s[:n], t[:m]
        for method in ("vanilla", "trimmed"):
            cfg = icp.Config()
            cfg.set("nn", nn)
            cfg.set("max_distance", 0.35)
            cls = icp.ICP2 if dim == 2 else icp.ICP3
            if nn == "auto" and n <= 4096 and m < 4000:
                continue  # would take the fused kernel
            d = driver(icp, cls, method, crit, cfg)
            r = d.converge(s, t)
            o = dev_oracle(ob, dim, method, max_distance=0.35)
            ro = o.converge(s, t, None, ob.make_criteria(**crit))
            assert r.iteration_count == ro.iteration_count, (dim, n, m, method)
            assert np.array_equal(r.transform, ro.transform), (dim, n, m, method)
            assert r.cost == ro.cost, (dim, n, m, method)
            if dim == 3:
                for u, v in zip(d.icp_.debug_neighbors(), o.neighbors()):
                    assert np.array_equal(u, v), (dim, n, m, method)
            else:
                for u, v in zip(d.icp_.get_matches_arrays(), o.get_matches()):
                    assert np.array_equal(u, v), (dim, n, m, method)
    # exact ties and far-from-origin data through the tiled search
    rng = np.random.default_rng(5)
    lat = rng.integers(0, 30, (6000, 3)).astype(float)
    for s, t in ((lat[:5000] + 0.5, lat), (lat[:5000] * 1e-3 + 7.0e5, lat * 1e-3 + 7.0e5 + 1e-4)):
        cfg = icp.Config()
        cfg.set("nn", nn)
        r = driver(icp, icp.ICP3, "vanilla", crit, cfg).converge(s, t)
        ro = dev_oracle(ob, 3, "vanilla").converge(s, t, None, ob.make_criteria(**crit))
        assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform)


def test_large_path_sum_layouts_agree(icp, ob, knob):
    """The multi-kernel path sums per 64-point group (n <= 2^18) or per 4096-point block (beyond, or
    ICPX_NO_GROUP_SUMS): both walk the same TREE64 tree, so a converge must not change by one bit;
    sizes around the group / block boundaries, and one cloud beyond 2^18 against the oracle."""
    from icp_amd import synth
    crit = dict(max_iterations=5, transform_tol=(1e-9, 1e-9))
    for dim, method, n, m in ((3, "vanilla", 4160, 5000), (3, "trimmed", 8193, 4500), (2, "trimmed", 12289, 6000),
                              (2, "vanilla", 4097, 4097), (3, "vanilla", 65, 9000), (3, "vanilla", 64, 9000)):
        s, t, _, _ = synth.config5_pairs([n + m], dim, max(n, m))
        s, t = s[:n], t[:m]
        cfg = icp.Config()
        cfg.set("nn", "grid")  # the multi-kernel path at any size
        cfg.set("max_distance", 0.35)
        cls = icp.ICP2 if dim == 2 else icp.ICP3
        out = []
        for env in (None, "1"):
            if env:
                knob.set("ICPX_NO_GROUP_SUMS", env)
            d = driver(icp, cls, method, crit, cfg)
            r = d.converge(s, t)
            out.append((r.iteration_count, r.transform.copy(), r.cost))
            knob.reset("ICPX_NO_GROUP_SUMS")
        assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2], (dim, method, n)
        ro = dev_oracle(ob, dim, method, max_distance=0.35).converge(s, t, None, ob.make_criteria(**crit))
        assert out[0][0] == ro.iteration_count and np.array_equal(out[0][1], ro.transform) and out[0][2] == ro.cost
    # beyond 2^18 source points: block sums only; few targets keep the oracle's brute force short
    n, m = (1 << 18) + 77, 700
    s, t, _, _ = synth.config5_pairs([n + m], 3, n)
    s, t = s[:n], t[:m]
    crit = dict(max_iterations=2)
    r = driver(icp, icp.ICP3, "vanilla", crit).converge(s, t)
    ro = dev_oracle(ob, 3, "vanilla").converge(s, t, None, ob.make_criteria(**crit))
    assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform)


@pytest.mark.parametrize("prune_handles", ["0", "1"])
def test_filtered_search_adversarial_inputs(icp, ob, prune_handles, knob):
    knob.set("ICPX_PRUNE_HANDLES", prune_handles)  # "1": the same inputs through the grid-pruned kernels
    """The f32 candidate filter of the fused kernel must never change a result: exact ties
    (lattices, duplicated targets), clouds far from the origin (the filter works about the target's
    bounding-box centre), tiny and huge scales (f32 underflow / the >= 1e18 fall-back to the plain
    f64 search), and sizes that leave padding in the last block of four."""
    rng = np.random.default_rng(21)
    crit = dict(max_iterations=8, transform_tol=(1e-9, 1e-9))

    def clouds(dim, n, m, kind):
        if kind == "lattice":
            t = rng.integers(0, 9, (m, dim)).astype(float)
            s = t[rng.integers(0, m, n)] + 0.5          # every query has many equidistant targets
        elif kind == "duplicates":
            base = rng.uniform(-5, 5, (m // 4 + 1, dim))
            t = np.concatenate([base] * 4)[:m]            # every target four times
            s = base[rng.integers(0, len(base), n)] + rng.normal(0, 0.05, (n, dim))
        else:
            t = rng.uniform(-5, 5, (m, dim))
            s = t[rng.integers(0, m, n)] + rng.normal(0, 0.05, (n, dim))
            if kind == "far":
                t, s = t + 1.0e6, s + 1.0e6
            elif kind == "far_small":
                t, s = t * 1e-3 + 12345.678, s * 1e-3 + 12345.678
            elif kind == "tiny":
                t, s = t * 1e-25, s * 1e-25
            elif kind == "huge":
                t, s = t * 1e19, s * 1e19
            elif kind == "mixed_scale":
                t[: m // 2] *= 1e-4
        return s, t

    for kind in ("lattice", "duplicates", "far", "far_small", "tiny", "huge", "mixed_scale"):
        for dim, n, m, method in ((2, 777, 1001, "trimmed"), (3, 1300, 1501, "vanilla"), (3, 4096, 4095, "trimmed"),
                                  (2, 100, 7, "trimmed"), (3, 65, 3, "vanilla")):
            s, t = clouds(dim, n, m, kind)
            cls = icp.ICP2 if dim == 2 else icp.ICP3
            cfg = icp.Config()
            md = 0.3 * float(np.abs(t).max()) + 1e-30
            cfg.set("max_distance", md)
            r = driver(icp, cls, method, crit, cfg).converge(s, t)
            ro = dev_oracle(ob, dim, method, max_distance=md).converge(
                s, t, None, ob.make_criteria(**crit))
            assert r.iteration_count == ro.iteration_count, (kind, dim, n, m, method)
            assert np.array_equal(r.transform, ro.transform, equal_nan=True), (kind, dim, n, m, method)
            assert r.cost == ro.cost or (np.isnan(r.cost) and np.isnan(ro.cost)), (kind, dim, n, m, method)
        # stepwise with matches / neighbours (seeds carried between launches)
        s, t = clouds(2, 900, 800, kind)
        x, o = icp.ICP2.from_method("trimmed"), dev_oracle(ob, 2, "trimmed")
        x.begin(s, t)
        o.begin(s, t)
        for it in range(4):
            x.iterate()
            o.iterate()
            for u, v in zip(x.get_matches_arrays(), o.get_matches()):
                assert np.array_equal(u, v), (kind, it)
        s, t = clouds(3, 2500, 2600, kind)
        x, o = icp.ICP3.from_method("vanilla"), dev_oracle(ob, 3, "vanilla")
        x.begin(s, t)
        o.begin(s, t)
        for it in range(3):
            x.iterate()
            o.iterate()
            for u, v in zip(x.debug_neighbors(), o.neighbors()):
                assert np.array_equal(u, v), (kind, it)


@pytest.mark.parametrize("chunk", ["1", "0"])
def test_trimmed2d_chunk_skipping_search(icp, ob, scans, knob, chunk):
    """Trimmed ICP2 in the brute-force kernels scans only the chunks of 32 consecutive targets some query of the wave
    can reach (nn_lds_chunked, icpx_filter.h; "0": the f32 filter instead).  Matches (point, target index, cost) equal
    the oracle's after begin and after every iteration on: a lidar scan in scan order (the coherent case), a cloud
    sorted along x with every target duplicated in another chunk (exact ties across chunks: the lowest index must
    win), points on a line, a shuffled cloud (every chunk needed), targets with NaN / inf coordinates."""
    knob.set("ICPX_CHUNK_SKIP", chunk)
    rng = np.random.default_rng(77)
    cases = [("scan", *scans[1])]
    t = rng.uniform(-5, 5, (600, 2))
    t = t[np.argsort(t[:, 0])]
    t = np.concatenate([t, t])                      # target j and j + 600 coincide
    cases.append(("sorted_dups", t[rng.integers(0, 600, 900)] + rng.normal(0, 0.02, (900, 2)), t))
    t = np.zeros((700, 2))
    t[:, 0] = np.sort(rng.uniform(-5, 5, 700))
    cases.append(("line", t[rng.integers(0, 700, 500)] + [0.0, 0.3], t))
    t = rng.uniform(-5, 5, (1000, 2))
    cases.append(("shuffled", t[rng.permutation(1000)[:800]] + rng.normal(0, 0.05, (800, 2)), t))
    t = scans[2][1].copy()
    t[5] = [np.nan, 0.1]
    t[40] = [np.inf, 0.0]
    t[41] = [0.3, -np.inf]
    t[300] = [np.nan, np.nan]
    cases.append(("nonfinite", scans[2][0], t))
    for name, s, t in cases:
        x, o = icp.ICP2.from_method("trimmed"), dev_oracle(ob, 2, "trimmed")
        x.begin(s, t)
        o.begin(s, t)
        for it in range(5):
            for u, v in zip(x.get_matches_arrays(), o.get_matches()):
                assert np.array_equal(u, v, equal_nan=True), (name, it)
            assert np.array_equal(x.current_transform(), o.current_transform(), equal_nan=True), (name, it)
            x.iterate()
            o.iterate()


def test_feature_aware_matches_oracle(icp, ob, scans):
    """FeatureAware (ICP2) on the multi-kernel path — features, normalised feature
    distances, weighted-cost matches, trimming, Kabsch — bit-exact against the oracle in device
    summation order, every iteration and through the driver; 1e-6 against the reference order."""
    for k, (a, b) in scans.items():
        x = icp.ICP2.from_method("feature_aware")
        o = dev_oracle(ob, 2, "feature_aware")
        g = np.eye(3)
        g[:2, 2] = (0.02, -0.01)
        x.begin(a, b, g)
        o.begin(a, b, g)
        assert x.calculate_cost() == o.calculate_cost(), k
        for u, v in zip(x.get_matches_arrays(), o.get_matches()):
            assert np.array_equal(u, v), k
        for it in range(3):
            x.iterate()
            o.iterate()
            assert np.array_equal(x.current_transform(), o.current_transform()), (k, it)
            assert x.calculate_cost() == o.calculate_cost(), (k, it)
            for u, v in zip(x.get_matches_arrays(), o.get_matches()):
                assert np.array_equal(u, v), (k, it)
        r = driver(icp, icp.ICP2, "feature_aware", BENCH_CRIT).converge(a, b)
        ro = o.converge(a, b, None, ob.make_criteria(**BENCH_CRIT))
        assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform), k
        assert r.cost == ro.cost
        rs = ob.OracleICP(2, "feature_aware").converge(a, b, None, ob.make_criteria(**BENCH_CRIT))
        assert rs.iteration_count == r.iteration_count
        assert np.abs(rs.transform - r.transform).max() < TOL
    # other parameters, ragged sizes, the reference's own test configuration (tests/test.cpp:228-232)
    rng = np.random.default_rng(9)
    for n, m, sn, fw, ov in ((300, 417, 1, 0.7, 1.0), (5000, 4100, 3, 0.2, 0.5), (64, 5, 0, 0.7, 0.9), (7, 900, 25, 1.0, 0.9)):
        t = np.cumsum(rng.normal(0, 0.05, (m, 2)), axis=0) + rng.uniform(-3, 3, 2)
        s = np.cumsum(rng.normal(0, 0.05, (n, 2)), axis=0) + rng.uniform(-3, 3, 2)
        cfg = icp.Config()
        cfg.set("overlap_rate", ov)
        cfg.set("feature_weight", fw)
        cfg.set("symmetric_neighbors", sn)
        crit = dict(max_iterations=5)
        r = driver(icp, icp.ICP2, "feature_aware", crit, cfg).converge(s, t)
        ro = dev_oracle(ob, 2, "feature_aware", overlap_rate=ov, feature_weight=fw, symmetric_neighbors=sn).converge(
            s, t, None, ob.make_criteria(**crit))
        assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform), (n, m, sn)
        assert r.cost == ro.cost, (n, m, sn)
    cfg = icp.Config()
    cfg.set("overlap_rate", 1.0)
    cfg.set("feature_weight", 0.7)
    cfg.set("symmetric_neighbors", 1)
    for case in reference_cases_2d():
        r = driver(icp, icp.ICP2, "feature_aware", TEST_CRIT, cfg).converge(case[1], case[2])
        check_case_2d(r.transform, r.iteration_count, case)
    assert icp.ICP3.from_method("feature_aware") is None  # registered for ICP2 only (lib/icp/icp.cpp:17-27)


def test_corrected_modes_match_oracle(icp, ob, scans):
    """the opt-in corrections of reference quirks (Config keys match_transformed /
    fill_matches / apply_guess; NOT reference behaviour, defined by the oracle) — bit-exact against
    the oracle run with the same flags, and the defaults untouched."""
    from icp_amd import synth
    # Q1: vanilla ICP2 matching transform*a in every iteration
    for k, (a, b) in scans.items():
        cfg = icp.Config()
        cfg.set("match_transformed", 1)
        d = driver(icp, icp.ICP2, "vanilla", BENCH_CRIT, cfg)
        r = d.converge(a, b)
        o = dev_oracle(ob, 2, "vanilla", flags=ob.FLAG_MATCH_TRANSFORMED)
        ro = o.converge(a, b, None, ob.make_criteria(**BENCH_CRIT))
        assert r.iteration_count == ro.iteration_count and r.iteration_count > 2, k
        assert np.array_equal(r.transform, ro.transform) and r.cost == ro.cost, k
        for u, v in zip(d.icp_.get_matches_arrays(), o.get_matches()):
            assert np.array_equal(u, v), k
    # Q4 / Q5: 3D methods filling matches and starting from guess*a
    s, t, _, _ = synth.config5_pairs([12], 3, 1500)
    g = np.eye(4)
    g[:3, 3] = (0.08, -0.05, 0.03)
    crit = dict(max_iterations=25, transform_tol=(1e-6, 1e-6))
    for method in ("vanilla", "trimmed"):
        for fill, guess in ((1, 0), (0, 1), (1, 1)):
            cfg = icp.Config()
            cfg.set("fill_matches", fill)
            cfg.set("apply_guess", guess)
            cfg.set("max_distance", 0.5)
            flags = (ob.FLAG_FILL_MATCHES if fill else 0) | (ob.FLAG_APPLY_GUESS if guess else 0)
            x = icp.ICP3.from_method(method, cfg)
            o = dev_oracle(ob, 3, method, max_distance=0.5, flags=flags)
            x.begin(s, t, g)
            o.begin(s, t, g)
            assert x.calculate_cost() == o.calculate_cost(), (method, fill, guess)
            for it in range(3):
                x.iterate()
                o.iterate()
                assert np.array_equal(x.current_transform(), o.current_transform()), (method, fill, guess, it)
                assert x.calculate_cost() == o.calculate_cost(), (method, fill, guess, it)
                for u, v in zip(x.get_matches_arrays(), o.get_matches()):
                    assert np.array_equal(u, v), (method, fill, guess, it)
            if fill:
                assert x.calculate_cost() > 0.0
            d = driver(icp, icp.ICP3, method, crit, cfg)
            r = d.converge(s, t, g)
            ro = o.converge(s, t, g, ob.make_criteria(**crit))
            assert r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform)
            assert r.cost == ro.cost


def test_handles_from_several_threads_and_reuse(icp, ob, scans, queen):
    """Different handles used concurrently from several host threads (the reference's instances are
    independent objects too), each re-used for clouds of different sizes: every result equals the
    single-threaded one; the handle-free entry points serialise on the per-device workspace."""
    import threading
    from icp_amd import synth
    s3, t3, off3, _ = synth.config5_pairs(np.arange(6), 3, 900)
    jobs = [(icp.ICP2, "trimmed", *scans[1], BENCH_CRIT), (icp.ICP2, "vanilla", *scans[2], BENCH_CRIT),
            (icp.ICP3, "vanilla", *queen, PLY_CRIT), (icp.ICP2, "feature_aware", *scans[3], BENCH_CRIT),
            (icp.ICP3, "trimmed", s3[:900], t3[:900], dict(max_iterations=10))]
    expect = [driver(icp, c, m, crit).converge(a, b) for c, m, a, b, crit in jobs]
    exp_batch = icp.converge_batch(3, "vanilla", s3, off3, t3, off3, icp.make_criteria(max_iterations=5), device=0)
    errors = []

    def worker(w):
        try:
            d = {}
            for rep in range(6):
                k = (w + rep) % len(jobs)
                c, m, a, b, crit = jobs[k]
                if (c, m) not in d:
                    d[(c, m)] = driver(icp, c, m, crit)
                r = d[(c, m)].converge(a, b)
                if not (np.array_equal(r.transform, expect[k].transform) and r.iteration_count == expect[k].iteration_count
                        and r.cost == expect[k].cost):
                    errors.append(("handle", w, rep, k))
                got = icp.converge_batch(3, "vanilla", s3, off3, t3, off3, icp.make_criteria(max_iterations=5), device=0)
                if not all(np.array_equal(u, v) for u, v in zip(got, exp_batch)):
                    errors.append(("batch", w, rep))
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", w, repr(e)))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_driver_criteria_on_device(icp, ob, scans):
    a, b = scans[1]
    for crit in [dict(max_iterations=5), dict(max_iterations=0), dict(min_iterations=7, max_iterations=50, stop_cost=1e9),
                 dict(max_iterations=50, absolute=1e-3), dict(max_iterations=50, relative=1e-2),
                 dict(max_iterations=50, stop_cost=0.3), dict(min_iterations=3, max_iterations=3)]:
        r = driver(icp, icp.ICP2, "trimmed", crit).converge(a, b)
        ro = dev_oracle(ob, 2, "trimmed").converge(a, b, None, ob.make_criteria(**crit))
        assert r.iteration_count == ro.iteration_count, crit
        assert np.array_equal(r.transform, ro.transform), crit
    # Q4 in 3D: the cost is 0, so absolute tolerance stops after 1 iteration, stop_cost at once
    from icp_amd import synth
    s, t, _, _ = synth.config5_pairs([1], 3, 300)
    assert driver(icp, icp.ICP3, "vanilla", dict(max_iterations=50, absolute=1e-9)).converge(s, t).iteration_count == 1
    assert driver(icp, icp.ICP3, "vanilla", dict(max_iterations=50, stop_cost=1e-9)).converge(s, t).iteration_count == 0
    # time limit: generous limit does not stop early; zero limit stops at the first check
    d = driver(icp, icp.ICP2, "trimmed", BENCH_CRIT)
    d.set_time_limit(60.0)
    assert d.converge(a, b).iteration_count == 22
    d.set_time_limit(0.0)
    assert d.converge(a, b).iteration_count == 0


@pytest.mark.parametrize("method,overlap", [("vanilla", 0.9), ("trimmed", 1.0)])
def test_reference_2d_cases_through_product_api(icp, method, overlap):
    """tests/test.cpp:83-216 with the reference's own tolerances."""
    cfg = icp.Config()
    cfg.set("overlap_rate", overlap)
    d = driver(icp, icp.ICP2, method, TEST_CRIT, cfg)
    for case in reference_cases_2d():
        r = d.converge(case[1], case[2])
        check_case_2d(r.transform, r.iteration_count, case)


def test_reference_3d_cases_through_product_api(icp):
    """tests/test3d.cpp:36-202 (Vanilla3d) with the reference's own tolerances."""
    d = driver(icp, icp.ICP3, "vanilla", TEST_CRIT)
    for case in reference_cases_3d():
        r = d.converge(case[1], case[2])
        check_case_3d(r.transform, r.iteration_count, case)


def test_empty_clouds_are_noops(icp):
    e2, p2 = np.zeros((0, 2)), np.array([[1.0, 2.0], [3.0, 4.0]])
    for method in ("vanilla", "trimmed"):
        x = icp.ICP2.from_method(method)
        x.begin(e2, p2)
        assert np.isnan(x.calculate_cost())  # sqrt(0/0), icp.h:154
        x.iterate()
        assert np.array_equal(x.current_transform(), np.eye(3))
        x.begin(p2, e2)
        x.iterate()
        assert np.array_equal(x.current_transform(), np.eye(3))
        assert x.calculate_cost() == 0.0
    r = driver(icp, icp.ICP3, "vanilla", dict(max_iterations=3)).converge(np.zeros((0, 3)), np.zeros((5, 3)))
    assert r.iteration_count == 3 and np.array_equal(r.transform, np.eye(4))


def test_batch_equals_single_pairs(icp, ob):
    from icp_amd import synth
    crit = icp.make_criteria(max_iterations=30, transform_tol=(1e-4, 1e-4))
    for dim, n, method in [(2, 300, "trimmed"), (2, 300, "vanilla"), (3, 500, "vanilla"), (3, 500, "trimmed")]:
        P = 9
        s, t, off, _ = synth.config5_pairs(range(100, 100 + P), dim, n)
        # ragged: drop some points of every pair; pair 4 is empty
        so, to = [0], [0]
        ss, tt = [], []
        for p in range(P):
            ns = 0 if p == 4 else n - 7 * p
            nt = n - 3 * p
            ss.append(s[off[p]:off[p] + ns])
            tt.append(t[off[p]:off[p] + nt])
            so.append(so[-1] + ns)
            to.append(to[-1] + nt)
        S, T = np.concatenate(ss), np.concatenate(tt)
        cost, its, Ts = icp.converge_batch(dim, method, S, np.array(so), T, np.array(to), crit)
        for p in range(P):
            if len(ss[p]) == 0:
                # Q10: the reference is UB (2D trimmed) / NaN (3D) on an empty source; the product
                # defines iterate() as a no-op there — not a parity target
                assert np.array_equal(Ts[p], np.eye(dim + 1)), (dim, method, p)
                continue
            ro = dev_oracle(ob, dim, method).converge(ss[p], tt[p], None,
                                                      ob.make_criteria(max_iterations=30, transform_tol=(1e-4, 1e-4)))
            assert its[p] == ro.iteration_count, (dim, method, p)
            assert np.array_equal(Ts[p], ro.transform), (dim, method, p)
            assert cost[p] == ro.cost or (np.isnan(cost[p]) and np.isnan(ro.cost))


def test_batch_grid_pruned_search_adversarial(icp, ob, knob):
    """icpx_converge_batch searches through a per-pair LDS grid (icpx_prune.h) that must return the
    lexicographic minimum of (d2, cloud index) whatever the geometry: exact ties and duplicates,
    degenerate extents (collinear / planar / a single repeated point), clusters, data far from the
    origin, sources far outside the target's box, non-finite and huge coordinates (exact scan),
    ragged batches mixing all of these.  Checked pair by pair against the oracle, and against the
    same batch with the grid switched off (ICPX_PRUNE=0: brute-force filter)."""
    rng = np.random.default_rng(23)
    crit_kw = dict(max_iterations=8, transform_tol=(1e-9, 1e-9))
    crit = icp.make_criteria(**crit_kw)

    def clouds(dim):
        c = []
        lat = rng.integers(0, 7, (900, dim)).astype(float)
        c.append((lat[:700] + 0.5, lat))                                   # every distance tied many times
        c.append((lat[:650], lat))                                         # zero distances, duplicates
        u = rng.uniform(-5, 5, (1200, dim))
        c.append((u[:900] * 1e-3 + 7.0e5, u * 1e-3 + 7.0e5 + 1e-4))       # far from the origin
        line = np.zeros((800, dim)); line[:, 0] = np.linspace(-3, 3, 800); line[:, 1] = 2.0
        c.append((rng.uniform(-4, 4, (500, dim)), line))                   # collinear target (zero extent on an axis)
        c.append((rng.normal(size=(300, dim)), np.tile(rng.normal(size=(1, dim)), (400, 1))))  # one repeated point
        clus = np.concatenate([rng.normal(cc, 0.01, (300, dim)) for cc in (-6.0, 0.0, 5.0)])
        c.append((rng.uniform(-8, 8, (1000, dim)), clus))                  # clustered, most cells empty
        c.append((rng.uniform(50, 60, (400, dim)), rng.uniform(-1, 1, (1500, dim))))  # source outside the box
        c.append((rng.uniform(-1, 1, (1024, dim)), rng.uniform(-1, 1, (130, dim))))   # few targets
        c.append((rng.uniform(-1, 1, (200, dim)), rng.uniform(-1, 1, (100, dim))))    # below the grid's minimum size
        if dim == 3:
            plane = np.concatenate([rng.uniform(-5, 5, (2000, 2)), np.zeros((2000, 1))], axis=1)
            c.append((rng.uniform(-5, 5, (900, 3)), plane))
        big = rng.uniform(-1, 1, (600, dim)); big[17, 0] = 1e200; big[300, 1] = -3e160
        c.append((rng.uniform(-1, 1, (500, dim)), big))                    # huge target coordinates: exact scan
        inf_t = rng.uniform(-1, 1, (600, dim)); inf_t[5, 0] = np.inf
        c.append((rng.uniform(-1, 1, (500, dim)), inf_t))
        src_bad = rng.uniform(-1, 1, (500, dim)); src_bad[9, 1] = 1e180
        c.append((src_bad, rng.uniform(-1, 1, (700, dim))))                # one huge query in a good wave
        return c

    for dim, method in ((3, "vanilla"), (3, "trimmed"), (2, "trimmed")):
        cs = clouds(dim)
        so = np.cumsum([0] + [len(a) for a, _ in cs])
        to = np.cumsum([0] + [len(b) for _, b in cs])
        S = np.concatenate([a for a, _ in cs])
        T = np.concatenate([b for _, b in cs])
        with np.errstate(all="ignore"):
            cfg = icp.Config()
            cfg.set("max_distance", 0.8)
            res = icp.converge_batch(dim, method, S, so, T, to, crit, cfg)
            knob.set("ICPX_PRUNE", "0")
            ref = icp.converge_batch(dim, method, S, so, T, to, crit, cfg)
            knob.reset("ICPX_PRUNE")
            for x, y in zip(res, ref):
                assert np.array_equal(x, y, equal_nan=True), (dim, method)
            for p, (a, b) in enumerate(cs):
                if not (np.isfinite(a).all() and np.isfinite(b).all()):
                    continue  # the brute-force batch above is the reference there (oracle-pinned through handles)
                ro = dev_oracle(ob, dim, method, max_distance=0.8).converge(a, b, None, ob.make_criteria(**crit_kw))
                assert res[1][p] == ro.iteration_count, (dim, method, p)
                assert np.array_equal(res[2][p], ro.transform, equal_nan=True), (dim, method, p)
                assert res[0][p] == ro.cost or (np.isnan(res[0][p]) and np.isnan(ro.cost)), (dim, method, p)


def test_batch_iteration_granular_scheduler_is_bit_identical(icp, knob):
    """Batches whose pairs slightly outnumber the resident workgroups run through persistent
    workgroups that hand pairs over between iterations (agent-scope release/acquire).  Forced on
    and off here: states must be bit-identical, whatever workgroup ran which iteration."""
    from icp_amd import synth
    crit = icp.make_criteria(max_iterations=40, transform_tol=(1e-4, 1e-4))
    for dim, n, method, P in [(3, 600, "vanilla", 700), (3, 600, "trimmed", 300), (2, 500, "trimmed", 900)]:
        s, t, off, _ = synth.config5_pairs(range(P), dim, n)
        res = {}
        for mode in ("0", "1"):
            knob.set("ICPX_SCHED", mode)
            res[mode] = icp.converge_batch(dim, method, s, off, t, off, crit)
        knob.reset("ICPX_SCHED")
        for a, b in zip(res["0"], res["1"]):
            assert np.array_equal(a, b, equal_nan=True), (dim, method)
        assert res["0"][1].min() >= 1 and len(set(res["0"][1].tolist())) > 1  # ragged iteration counts


def test_device_resident_inputs_and_outputs(icp, scans):
    """torch CUDA tensors go into the C ABI as device pointers (no host staging)."""
    import torch
    a, b = scans[1]
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    r_host = driver(icp, icp.ICP2, "trimmed", BENCH_CRIT).converge(a, b)
    r_dev = driver(icp, icp.ICP2, "trimmed", BENCH_CRIT).converge(da, db)
    assert r_dev.iteration_count == r_host.iteration_count and np.array_equal(r_dev.transform, r_host.transform)
    out = torch.zeros(18, dtype=torch.float64, device="cuda")
    crit = icp.make_criteria(**BENCH_CRIT)
    icp.converge_batch(2, "trimmed", da, np.array([0, len(a)]), db, np.array([0, len(b)]), crit, out_states=out)
    o = out.cpu().numpy()
    assert o[0] == r_host.cost and o[1:2].view(np.uint64)[0] == r_host.iteration_count
    assert np.array_equal(o[2:11].reshape(3, 3, order="F"), r_host.transform)
    idx = torch.zeros(len(a), dtype=torch.int64, device="cuda")
    d2 = torch.zeros(len(a), dtype=torch.float64, device="cuda")
    icp.nearest_neighbors(da, db, icp.NN_BRUTE, out_idx=idx, out_d2=d2)
    hi, hd = icp.nearest_neighbors(a, b, icp.NN_GRID)
    assert np.array_equal(idx.cpu().numpy(), hi) and np.array_equal(d2.cpu().numpy(), hd)


def test_error_behaviour(icp):
    L = icp._lib
    lib = L.load()
    h = L._vp()
    assert lib.icpx_create(4, 0, None, C.byref(h)) == L.ERR_INVALID
    assert b"dim" in lib.icpx_last_error()
    assert lib.icpx_create(2, 7, None, C.byref(h)) == L.ERR_INVALID
    assert lib.icpx_create(2, 0, None, C.byref(h)) == L.OK
    assert lib.icpx_iterate(h) == L.ERR_STATE  # iterate before begin (icp.h:42-43 calls it UB)
    out = C.c_double()
    assert lib.icpx_cost(h, C.byref(out)) == L.ERR_STATE
    a = np.zeros((3, 2))
    g = np.eye(3)
    assert lib.icpx_begin(h, None, 3, a.ctypes.data, 3, g.ctypes.data) == L.ERR_INVALID
    assert lib.icpx_begin(h, a.ctypes.data, -1, a.ctypes.data, 3, g.ctypes.data) == L.ERR_INVALID
    assert lib.icpx_begin(h, a.ctypes.data, 3, a.ctypes.data, 3, g.ctypes.data) == L.OK
    assert lib.icpx_iterate(h) == L.OK
    assert lib.icpx_destroy(h) == L.OK
    assert lib.icpx_destroy(None) == L.OK
    assert lib.icpx_iterate(None) == L.ERR_INVALID
    # a source cloud above the multi-kernel path's limit: a clear error, not a wrong answer
    h = L._vp()
    assert lib.icpx_create(2, 1, None, C.byref(h)) == L.OK
    st = L.State()
    crit = icp.make_criteria(max_iterations=1)
    assert lib.icpx_converge(h, a.ctypes.data, (1 << 24) + 1, a.ctypes.data, 3, g.ctypes.data,
                             C.byref(crit), C.byref(st)) == L.ERR_UNSUPPORTED
    assert lib.icpx_destroy(h) == L.OK


# ------------------------------------------------------------------------------------------
# BASELINE full sizes: size-independent properties
# ------------------------------------------------------------------------------------------

def test_config4_grid_nn_one_million_points(icp, ob):
    """1 M x 1 M (BASELINE config 4): (1) self-query is the identity with d2 = 0 (idempotence);
    (2) 300 sampled queries bit-exact against the oracle's brute force over all 1 M targets;
    (3) the clustered variant (non-uniform cells) likewise."""
    from icp_amd import synth
    for clustered in (False, True):
        src, tgt, R, t = synth.config4(1_000_000, clustered=clustered)
        idx, d2 = icp.nearest_neighbors(tgt, tgt, icp.NN_GRID)
        uniq = len(np.unique(tgt, axis=0)) == len(tgt)
        assert np.all(d2 == 0.0)
        if uniq:
            assert np.array_equal(idx, np.arange(len(tgt)))
        idx, d2 = icp.nearest_neighbors(src, tgt, icp.NN_GRID)
        sample = np.random.default_rng(5).choice(len(src), 300, replace=False)
        oi, od2 = ob.nn_brute(src[sample], tgt)
        assert np.array_equal(idx[sample], oi) and np.array_equal(d2[sample], od2)
        assert np.all(d2 >= 0) and np.all(np.isfinite(d2))


def test_grid_cooperative_query_is_bit_identical(icp, knob):
    """The block-cooperative grid query (queries in tile order, a block's cells staged in LDS; from 32 768 queries on) against
    the per-lane walk (ICPX_GRID_COOP=0) on ALL queries, and against the brute-force kernel at a size it covers:
    uniform, clustered (non-uniform cells: many waves fall back), a lattice (every distance tied many times) and
    queries far outside the target's box."""
    from icp_amd import synth
    rng = np.random.default_rng(11)
    cases = []
    src, tgt, _, _ = synth.config4(1_000_000)
    cases.append(("config4", src, tgt, False))
    src, tgt, _, _ = synth.config4(300_000, clustered=True)
    cases.append(("clustered", src, tgt, False))
    lat = np.stack(np.meshgrid(np.arange(36.0), np.arange(36.0), np.arange(36.0), indexing="ij"), -1).reshape(-1, 3)
    cases.append(("lattice", lat[rng.permutation(len(lat))][:40_000] + 0.5, lat, True))
    far = rng.uniform(-30, 30, (40_000, 3))
    cases.append(("outside", far, rng.uniform(-10, 10, (50_000, 3)), True))
    q2 = rng.uniform(-10, 10, (60_000, 2))
    cases.append(("2d", q2, rng.uniform(-10, 10, (45_000, 2)), True))
    for name, q, t, brute in cases:
        knob.set("ICPX_GRID_COOP", "2")  # block-cooperative whatever the grid's fill (1, the default: well-filled grids only)
        i1, d1 = icp.nearest_neighbors(q, t, icp.NN_GRID)
        knob.set("ICPX_GRID_COOP", "0")
        i0, d0 = icp.nearest_neighbors(q, t, icp.NN_GRID)
        assert np.array_equal(i1, i0) and np.array_equal(d1, d0), name
        if brute:
            ib, db = icp.nearest_neighbors(q, t, icp.NN_BRUTE)
            assert np.array_equal(i1, ib) and np.array_equal(d1, db), name


def test_bvh_search_is_bit_identical(icp, ob, scans, queen):
    """ICPX_NN_BVH (linear BVH, icpx_bvh.h) returns the same lexicographic (d2, index) minimum as the brute-force and
    grid searches: the reference's scans and queen.ply (duplicated vertices: exact ties), a lattice (every distance tied
    many times), clustered clouds (the case it exists for), queries far outside the targets' box, degenerate extents,
    non-finite coordinates, tiny and empty clouds."""
    from icp_amd import synth
    rng = np.random.default_rng(23)
    cases = [("scan1", *scans[1]), ("scan2", *scans[2]), ("queen", *queen)]
    lat = np.stack(np.meshgrid(np.arange(30.0), np.arange(30.0), np.arange(30.0), indexing="ij"), -1).reshape(-1, 3)
    cases.append(("lattice", lat[rng.permutation(len(lat))][:20_000] + 0.5, lat))
    src, tgt, _, _ = synth.config4(200_000, clustered=True)
    cases.append(("clustered", src[:50_000], tgt))
    cases.append(("outside", rng.uniform(-40, 40, (40_000, 3)), rng.uniform(-10, 10, (60_000, 3))))
    flat = rng.uniform(-5, 5, (30_000, 3))
    flat[:, 2] = 1.25                                           # all targets in one plane
    cases.append(("plane", rng.uniform(-6, 6, (35_000, 3)), flat))
    cases.append(("2d", rng.uniform(-10, 10, (40_000, 2)), rng.uniform(-10, 10, (33_333, 2))))
    cases.append(("dups", rng.uniform(0, 1, (2_000, 3)), np.repeat(rng.uniform(0, 1, (300, 3)), 7, axis=0)))
    cases.append(("tiny", rng.uniform(0, 1, (100, 3)), rng.uniform(0, 1, (5, 3))))
    bad_t = rng.uniform(-1, 1, (5_000, 3))
    bad_t[::97, 1] = np.nan
    bad_t[5::131, 0] = np.inf
    bad_q = rng.uniform(-1, 1, (3_000, 3))
    bad_q[::53, 2] = np.nan
    bad_q[7::61, 0] = -np.inf
    cases.append(("nonfinite", bad_q, bad_t))
    for name, q, t in cases:
        ib, db = icp.nearest_neighbors(q, t, icp.NN_BRUTE)
        iv, dv = icp.nearest_neighbors(q, t, icp.NN_BVH)
        assert np.array_equal(iv, ib), name
        assert np.array_equal(dv, db), name
    idx, d2 = icp.nearest_neighbors(np.zeros((4, 3)), np.zeros((0, 3)), icp.NN_BVH)
    assert np.all(idx == 0) and np.all(np.isinf(d2))
    # two tight clusters far apart: the grid's cell budget leaves each cluster in a handful of cells (degenerate),
    # so nn = auto goes to the BVH by itself; same answers as the forced modes
    a = np.concatenate([rng.normal(0.0, 0.004, (30_000, 3)), rng.normal(100.0, 0.004, (30_000, 3))])
    b = np.concatenate([rng.normal(0.0, 0.004, (30_000, 3)), rng.normal(100.0, 0.004, (30_000, 3))])
    ia, da = icp.nearest_neighbors(a, b, icp.NN_AUTO)
    iv, dv = icp.nearest_neighbors(a, b, icp.NN_BVH)
    ib, db = icp.nearest_neighbors(a, b, icp.NN_BRUTE)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    assert np.array_equal(iv, ib) and np.array_equal(dv, db)
    # at BASELINE config 4's size: against the grid on every query, uniform and clustered
    for clustered in (False, True):
        src, tgt, _, _ = synth.config4(1_000_000, clustered=clustered)
        ig, dg = icp.nearest_neighbors(src, tgt, icp.NN_GRID)
        iv, dv = icp.nearest_neighbors(src, tgt, icp.NN_BVH)
        assert np.array_equal(iv, ig) and np.array_equal(dv, dg), clustered


def test_config4_converge_recovers_known_transform(icp):
    """Full converge at 1 M points (50 iterations; at this density point-to-point ICP creeps, the
    oracle does the same at smaller n): re-running is bit-reproducible and the estimate has moved
    most of the way from the identity guess towards the generator's inverse (R, t)."""
    from icp_amd import synth
    src, tgt, R, t = synth.config4(1_000_000)
    crit = dict(max_iterations=50, transform_tol=(1e-4, 1e-4))
    r1 = driver(icp, icp.ICP3, "vanilla", crit).converge(src, tgt)
    r2 = driver(icp, icp.ICP3, "vanilla", crit).converge(src, tgt)
    assert r1.iteration_count == r2.iteration_count and np.array_equal(r1.transform, r2.transform)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    assert np.abs(r1.transform @ T - np.eye(4)).max() < 0.5 * np.abs(T - np.eye(4)).max()


def test_grid_seeded_query_is_bit_identical(icp, ob, knob):
    """From the second correspondence step of a begin() on, the HBM grid's per-lane query starts from the previous
    step's neighbour and searches only its ball (grid_query_seeded); ICPX_GRID_SEED=0 keeps the unseeded query.  Both
    must give the same converge bit for bit — transforms, cost, every neighbour and distance — on uniform and
    clustered clouds, a lattice (every distance tied many times, the seed is one of the tied targets), sources far
    outside the targets' box (large balls: the general ball scan), non-finite points, 2D trimmed; and the oracle's
    result at a size it covers.  The 3D methods' seeded query scans 16-byte f32 FILTER records and re-evaluates the
    survivors from the cloud in f64 (grid_scan_rows_filtered; ICPX_GRID_FILTER=0: the f64 records) — also bit for bit."""
    from icp_amd import synth
    rng = np.random.default_rng(31)
    cases = []
    src, tgt, _, _ = synth.config4(300_000)
    cases.append(("config4", 3, "vanilla", src, tgt, False))
    src, tgt, _, _ = synth.config4(200_000, clustered=True)
    cases.append(("clustered", 3, "trimmed", src, tgt, False))
    lat = np.stack(np.meshgrid(np.arange(24.0), np.arange(24.0), np.arange(24.0), indexing="ij"), -1).reshape(-1, 3)
    cases.append(("lattice", 3, "vanilla", lat[rng.permutation(len(lat))][:9_000] + 0.5, lat, True))
    cases.append(("outside", 3, "vanilla", rng.uniform(-10, 10, (20_000, 3)) + 25.0, rng.uniform(-10, 10, (30_000, 3)), False))
    bad_t = rng.uniform(-1, 1, (9_000, 3))
    bad_t[::97, 1] = np.nan
    bad_q = rng.uniform(-1, 1, (8_000, 3))
    bad_q[::53, 2] = np.nan
    bad_q[7::61, 0] = -np.inf
    cases.append(("nonfinite-targets", 3, "vanilla", rng.uniform(-1, 1, (8_000, 3)), bad_t, False))
    s2, t2, _, _ = synth.config5_pairs([14_000], 2, 8_000)
    cases.append(("2d", 2, "trimmed", s2[:6_000], t2[:8_000], True))
    # far from the origin, and a cloud of tiny extent there: the filter records are relative to the grid's corner
    so, to, _, _ = synth.config4(40_000)
    cases.append(("offset", 3, "vanilla", so + 7.0e5, to + 7.0e5, False))
    cases.append(("offset-tiny", 3, "trimmed", so * 1e-6 + 3.0e3, to * 1e-6 + 3.0e3, False))
    crit = dict(max_iterations=7, transform_tol=(1e-9, 1e-9))
    for name, dim, method, s, t, oracle in cases:
        out = []
        # seeded through the f32 filter records (default), unseeded (f64 records only), seeded on the f64 records
        for env, filt in (("1", "1"), ("0", "1"), ("1", "0")):
            knob.set("ICPX_GRID_SEED", env)
            knob.set("ICPX_GRID_FILTER", filt)
            cfg = icp.Config()
            cfg.set("nn", "grid")
            cfg.set("max_distance", 0.35)
            d = driver(icp, icp.ICP2 if dim == 2 else icp.ICP3, method, crit, cfg)
            r = d.converge(s, t)
            nb = d.icp_.debug_neighbors() if dim == 3 else d.icp_.get_matches_arrays()
            out.append((r.iteration_count, r.transform.copy(), r.cost, [np.array(u) for u in nb]))
        knob.reset("ICPX_GRID_SEED")
        knob.reset("ICPX_GRID_FILTER")
        for other in (out[1], out[2]):
            assert out[0][0] == other[0] and np.array_equal(out[0][1], other[1]), name
            assert out[0][2] == other[2] or (np.isnan(out[0][2]) and np.isnan(other[2])), name
            for u, v in zip(out[0][3], other[3]):
                assert np.array_equal(u, v, equal_nan=True), name
        if oracle:
            ro = dev_oracle(ob, dim, method, max_distance=0.35).converge(s, t, None, ob.make_criteria(**crit))
            assert out[0][0] == ro.iteration_count and np.array_equal(out[0][1], ro.transform) and out[0][2] == ro.cost, name
    # a non-finite SOURCE point: the reference's sums go NaN from the first iteration; seeded and unseeded agree on that too
    # (compared through the neighbours of the first two steps)
    for env in ("1", "0"):
        knob.set("ICPX_GRID_SEED", env)
        cfg = icp.Config()
        cfg.set("nn", "grid")
        x = icp.ICP3.from_method("vanilla", cfg)
        x.begin(bad_q, bad_t, np.eye(4))
        x.iterate()
        x.iterate()
        nb = [np.array(u) for u in x.debug_neighbors()]
        if env == "1":
            first = nb
        else:
            for u, v in zip(first, nb):
                assert np.array_equal(u, v, equal_nan=True)
    knob.reset("ICPX_GRID_SEED")


def test_tiled_search_queries_per_thread_variants(icp, ob):
    """The tiled multi-kernel search picks one, two or four queries per thread from the cloud sizes (large_begin):
    sizes that select two (14 000 x 14 000) and four (20 000 x 20 000) — the other large-path tests all select one —
    against the oracle, transforms, cost and every neighbour; ragged sizes so that the last tile and split are partial."""
    from icp_amd import synth
    crit = dict(max_iterations=3, transform_tol=(1e-9, 1e-9))
    for n, m in ((14_003, 13_999), (20_001, 20_010)):
        s, t, _, _ = synth.config4(max(n, m))
        s, t = s[:n], t[:m]
        d = driver(icp, icp.ICP3, "vanilla", crit)
        r = d.converge(s, t)
        ro = dev_oracle(ob, 3, "vanilla").converge(s, t, None, ob.make_criteria(**crit))
        assert r.iteration_count == ro.iteration_count, (n, m)
        assert np.array_equal(r.transform, ro.transform) and r.cost == ro.cost, (n, m)
    # neighbours after two explicit iterations at the four-queries size
    x = icp.ICP3.from_method("vanilla")
    o = dev_oracle(ob, 3, "vanilla")
    x.begin(s, t, np.eye(4))
    o.begin(s, t, np.eye(4))
    for _ in range(2):
        x.iterate()
        o.iterate()
    for u, v in zip(x.debug_neighbors(), o.neighbors()):
        assert np.array_equal(u, v)


def test_grid_build_deferred_occupancy_transitions(icp, queen):
    """The stand-alone grid query reads the build's occupied-cell count after it has enqueued the query order
    (grid_settle) and rebuilds when the cloud needs a finer cell; an index that was rebuilt counts the patient way next
    time.  Clouds that do (queen.ply: a surface) and do not (uniform) need the finer round, alternating on one
    workspace, each answer against the brute-force kernel."""
    rng = np.random.default_rng(41)
    uni_t = rng.uniform(-1, 1, (40_000, 3))
    uni_q = rng.uniform(-1.1, 1.1, (40_000, 3))
    qs, qt = queen
    big_q = np.concatenate([qs] * 4) + rng.normal(0, 1e-3, (4 * len(qs), 3))   # >= 32 768 queries: ordered by tile
    for q, t in ((uni_q, uni_t), (big_q, qt), (big_q, qt), (uni_q, uni_t), (uni_q, uni_t), (big_q, qt)):
        ig, dg = icp.nearest_neighbors(q, t, icp.NN_GRID)
        ib, db = icp.nearest_neighbors(q, t, icp.NN_BRUTE)
        assert np.array_equal(ig, ib) and np.array_equal(dg, db)


def test_polled_done_flag_equals_copied_flag_across_reuse(icp, queen, knob):
    """Multi-kernel converge: the done flag / iteration count polled from pinned host memory (ICPX_POLL_DONE=1, the
    default) against the copy + stream synchronisation per chunk (ICPX_POLL_DONE=0) — same state, bit for bit, also
    when ONE handle is reused: converge, stepwise begin / iterate (whose kernels must not publish into the polled
    flags of a finished converge), converge again."""
    from icp_amd import synth
    src, tgt, _, _ = synth.config4(50_000)
    cases = [("queen", queen[0], queen[1], dict(max_iterations=100, transform_tol=(1e-4, 1e-4))),
             ("grid", src, tgt, dict(max_iterations=9, transform_tol=(1e-9, 1e-9)))]
    for name, a, b, crit in cases:
        out = []
        for env in ("1", "0"):
            knob.set("ICPX_POLL_DONE", env)
            d = driver(icp, icp.ICP3, "vanilla", crit)
            r1 = d.converge(a, b)
            x = d.icp_
            x.begin(a, b, np.eye(4))
            for _ in range(3):
                x.iterate()
            mid = x.current_transform().copy()
            r2 = d.converge(a, b)
            r3 = d.converge(a[: len(a) // 2], b)
            out.append((r1, mid, r2, r3))
        knob.reset("ICPX_POLL_DONE")
        (p1, pm, p2, p3), (c1, cm, c2, c3) = out
        assert p1.iteration_count == p2.iteration_count == c1.iteration_count == c2.iteration_count, name
        assert np.array_equal(p1.transform, p2.transform) and np.array_equal(p1.transform, c1.transform), name
        assert np.array_equal(c1.transform, c2.transform) and np.array_equal(pm, cm), name
        assert p3.iteration_count == c3.iteration_count and np.array_equal(p3.transform, c3.transform), name


def test_tiled_search_interleaved_chunks_are_bit_identical(icp, ob, queen, knob):
    """The tiled search of the multi-kernel path hands split s every S-th chunk of 32 consecutive targets and skips
    the chunks no query of a wave can need (ICPX_TILE_CHUNK=1, the default); ICPX_TILE_CHUNK=0 keeps contiguous splits
    and scans every block.  Same converge bit for bit — queen.ply (coherent vertex order: most chunks skipped), a
    lattice (exact ties everywhere), random order, and target counts around the chunk and split boundaries (1, 31, 32,
    33 ... 9 999 targets); the oracle at the sizes it covers."""
    rng = np.random.default_rng(77)
    crit = dict(max_iterations=8, transform_tol=(1e-9, 1e-9))
    cases = [("queen", queen[0], queen[1], "vanilla", False)]
    lat = np.stack(np.meshgrid(np.arange(20.0), np.arange(20.0), np.arange(20.0), indexing="ij"), -1).reshape(-1, 3)
    cases.append(("lattice", lat[rng.permutation(len(lat))][:5000] + 0.5, lat, "vanilla", True))
    for m in (1, 31, 32, 33, 95, 500, 1025, 3071, 3073, 9999):
        t = rng.uniform(-5, 5, (m, 3))
        s = rng.uniform(-5, 5, (5000, 3))
        cases.append((f"m{m}", s, t, "trimmed" if m % 2 else "vanilla", m <= 3073))
    srt = rng.uniform(-5, 5, (12000, 3))
    srt = srt[np.argsort(srt[:, 2])]                      # slabs along z, like a mesh
    cases.append(("sorted", srt[::2] + rng.normal(0, 0.02, (6000, 3)), srt, "vanilla", False))
    for name, s, t, method, oracle in cases:
        out = []
        for env in ("1", "0"):
            knob.set("ICPX_TILE_CHUNK", env)
            cfg = icp.Config()
            cfg.set("nn", "brute")
            cfg.set("max_distance", 0.8)
            d = driver(icp, icp.ICP3, method, crit, cfg)
            r = d.converge(s, t)
            nb = d.icp_.debug_neighbors()
            out.append((r.iteration_count, r.transform.copy(), np.array(nb[0]), np.array(nb[1])))
        knob.reset("ICPX_TILE_CHUNK")
        assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]), name
        assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3]), name
        if oracle:
            o = dev_oracle(ob, 3, method, max_distance=0.8)
            ro = o.converge(s, t, None, ob.make_criteria(**crit))
            assert out[0][0] == ro.iteration_count and np.array_equal(out[0][1], ro.transform), name
            for u, v in zip(out[0][2:], o.neighbors()):
                assert np.array_equal(u, v), name


def test_kd_tree_eager_or_on_demand_is_the_same_converge(icp, ob, queen, knob):
    """kd-tree tie rule on the multi-kernel path: the target's tree is built at begin() (ICPX_KD_EAGER=1), or only when
    a search meets tie partners — the device stalls, the host builds the tree, the flagged queries are resolved, the
    step resumes (ICPX_KD_EAGER=0) — or, the default, on demand until a handle has stalled once and at begin() from
    then on.  Same converge bit for bit, neighbours included: a lattice (ties in every step) through the tiled search
    and through the grid, queen.ply (no ties: no tree at all by default); the oracle on the lattice."""
    rng = np.random.default_rng(78)
    crit = dict(max_iterations=8, transform_tol=(1e-9, 1e-9))
    lat = np.stack(np.meshgrid(np.arange(20.0), np.arange(20.0), np.arange(20.0), indexing="ij"), -1).reshape(-1, 3)
    lsrc = lat[rng.permutation(len(lat))][:5000] + 0.5
    for name, s, t, nn, oracle in (("lattice-tiled", lsrc, lat, "brute", True), ("lattice-grid", lsrc, lat, "grid", False),
                                   ("queen", queen[0], queen[1], "brute", False)):
        cfg = icp.Config()
        cfg.set("nn", nn)
        cfg.set("max_distance", 0.8)

        def run(d):
            r = d.converge(s, t)
            nb = d.icp_.debug_neighbors()
            return r.iteration_count, r.transform.copy(), np.array(nb[0]), np.array(nb[1])

        out = {}
        for env in ("1", "0"):
            knob.set("ICPX_KD_EAGER", env)
            out[env] = run(driver(icp, icp.ICP3, "vanilla", crit, cfg))
        knob.reset("ICPX_KD_EAGER")
        d = driver(icp, icp.ICP3, "vanilla", crit, cfg)
        out["first"] = run(d)    # on demand
        out["second"] = run(d)   # the lattice: at begin() now; queen: on demand again
        for k in ("0", "first", "second"):
            assert out[k][0] == out["1"][0] and np.array_equal(out[k][1], out["1"][1]), (name, k)
            assert np.array_equal(out[k][2], out["1"][2]) and np.array_equal(out[k][3], out["1"][3]), (name, k)
        if oracle:
            o = dev_oracle(ob, 3, "vanilla", max_distance=0.8)
            ro = o.converge(s, t, None, ob.make_criteria(**crit))
            assert out["1"][0] == ro.iteration_count and np.array_equal(out["1"][1], ro.transform), name
            for u, v in zip(out["1"][2:], o.neighbors()):
                assert np.array_equal(u, v), name


def test_first_tiled_search_seeds_from_a_sample(icp, ob):
    """The first tiled search of a begin() starts every query from the nearest of a strided sample of the targets
    (lg_tile_seed_kernel).  The neighbours after one iteration are those of the brute-force rule all the same — on a
    random cloud, far from the origin, and with non-finite values among the queries and the targets (the sampled
    positions included), where the search falls back to its unseeded form."""
    rng = np.random.default_rng(79)
    cfg = icp.Config()
    cfg.set("nn", "brute")
    cases = []
    q, t = rng.uniform(-5, 5, (5000, 3)), rng.uniform(-5, 5, (6001, 3))
    cases.append(("random", q, t))
    cases.append(("offset", q + 3.0e6, t + 3.0e6))
    tb = t.copy()
    tb[0] = np.nan                      # the sample's origin
    tb[12] = [np.inf, 0.0, 0.0]         # a sampled position (stride 12)
    tb[100:140, 1] = np.nan
    cases.append(("bad-targets", q, tb))
    qb = q.copy()
    qb[7] = np.nan
    qb[4000, 2] = np.inf
    cases.append(("bad-queries", qb, t))
    for name, a, b in cases:
        x = icp.ICP3.from_method("vanilla", cfg)
        x.begin(a, b, np.eye(4))
        x.iterate()
        got = np.array(x.debug_neighbors()[0])
        oi, _ = ob.nn_brute(a, b)
        ok = np.isfinite(a).all(axis=1)
        assert np.array_equal(got[ok], oi[ok]), (name, np.nonzero(got != oi)[0][:8])
