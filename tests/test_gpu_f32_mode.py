"""Opt-in f32 correspondence mode (ICPX_NN_GRID_F32 / Config nn = "grid_f32").

NOT reference arithmetic and never the default: the grid search reads 16-byte f32 records and evaluates candidates
in f32, so the reference holds no assertion for it ("parity unpinned" by construction).  What these tests pin is the
mode's own contract (include/icpx/icpx.h): the returned target is the nearest one up to the f32 rounding of the
coordinates, everything downstream of that choice is the f64 path, and the mode is refused where it is not offered.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def icp():
    import icp_amd
    icp_amd._lib.check(icp_amd._lib.load().icpx_init(0))
    return icp_amd


def exact_d2(q, t, idx):
    d = t[idx] - q
    return (d * d).sum(axis=1)


def check_nearest_up_to_rounding(icp, q, t, name):
    i64, d64 = icp.nearest_neighbors(q, t, icp.NN_GRID)
    i32, d32 = icp.nearest_neighbors(q, t, icp.NN_GRID_F32)
    assert i32.min() >= 0 and i32.max() < len(t), name
    ext = float(np.max(t.max(axis=0) - t.min(axis=0)))
    got = exact_d2(q, t, i32)
    # the target the f32 search returns is as near as the true nearest one up to the rounding of the moved
    # coordinates (2^-24 relative to the extent per coordinate, a few of them per distance): 2^-20 of the extent
    tol = ext * 2.0 ** -20
    assert np.all(np.sqrt(got) <= np.sqrt(d64) + tol), (name, float(np.max(np.sqrt(got) - np.sqrt(d64))), tol)
    # ... and on generic data it is the same target nearly always
    assert np.mean(i32 == i64) > 0.999, (name, float(np.mean(i32 == i64)))
    # d2 is the f32-evaluated squared distance, widened: close to the exact one to the returned target
    reach = np.sqrt(got) + ext * 2.0 ** -22
    assert np.all(np.abs(d32 - got) <= 4.0 * reach * ext * 2.0 ** -22 + got * 2.0 ** -21), name


def test_f32_grid_query_is_nearest_up_to_rounding(icp):
    from icp_amd import synth
    rng = np.random.default_rng(41)
    # per-lane kernel (few queries: no query order), block-cooperative kernel (sorted queries, well-filled grid),
    # per-lane kernel on sorted queries (clustered: most cells empty), 2D, queries outside the targets' box
    check_nearest_up_to_rounding(icp, rng.uniform(-10, 10, (5_000, 3)), rng.uniform(-10, 10, (7_001, 3)), "small3d")
    src, tgt, _, _ = synth.config4(200_000)
    check_nearest_up_to_rounding(icp, src, tgt, "config4_200k")
    src, tgt, _, _ = synth.config4(200_000, clustered=True)
    check_nearest_up_to_rounding(icp, src, tgt, "clustered_200k")
    check_nearest_up_to_rounding(icp, rng.uniform(-10, 10, (60_000, 2)), rng.uniform(-10, 10, (45_000, 2)), "2d")
    check_nearest_up_to_rounding(icp, rng.uniform(-30, 30, (40_000, 3)), rng.uniform(-10, 10, (50_000, 3)), "outside")
    # far from the origin: the records are relative to the grid's corner, so nothing is lost to the offset
    check_nearest_up_to_rounding(icp, rng.uniform(-1, 1, (40_000, 3)) + 7.0e5, rng.uniform(-1, 1, (40_000, 3)) + 7.0e5, "offset")
    # the f64 grid mode after an f32 call on the same workspace is the exact search again
    q, t = rng.uniform(-10, 10, (40_000, 3)), rng.uniform(-10, 10, (40_000, 3))
    icp.nearest_neighbors(q, t, icp.NN_GRID_F32)
    ig, dg = icp.nearest_neighbors(q, t, icp.NN_GRID)
    ib, db = icp.nearest_neighbors(q, t, icp.NN_BRUTE)
    assert np.array_equal(ig, ib) and np.array_equal(dg, db)
    # empty target cloud: the conventions of the other modes
    idx, d2 = icp.nearest_neighbors(np.zeros((4, 3)), np.zeros((0, 3)), icp.NN_GRID_F32)
    assert np.all(idx == 0) and np.all(np.isinf(d2))


@pytest.mark.parametrize("method", ["vanilla", "trimmed"])
@pytest.mark.parametrize("n", [20_000, 120_000])
def test_f32_grid_converge_follows_the_f64_converge(icp, method, n):
    """Handles with nn = grid_f32 (20 000 points: per-lane queries in point order; 120 000: the sorted domain): the
    correspondence index is chosen in f32, cost / selection / sums / Kabsch are the f64 path.  First iteration (the
    queries are the source cloud itself in both modes): every neighbour is the nearest up to the rounding.  Whole
    converge: it follows the f64 converge; the few queries that sit within the rounding of the midplane between two
    targets take the other one — a target a grid cell away — which moves the estimate in the sixth digit."""
    from icp_amd import synth
    src, tgt, _, _ = synth.config4(n)
    ext = float(np.max(tgt.max(axis=0) - tgt.min(axis=0)))
    out = {}
    for nn in ("grid", "grid_f32"):
        cfg = icp.Config()
        cfg.set("nn", nn)
        cfg.set("max_distance", 0.6)
        x = icp.ICP3.from_method(method, cfg)
        x.begin(src, tgt, np.eye(4))
        x.iterate()
        first = x.debug_neighbors()
        x.iterate()
        x.iterate()
        third = x.debug_neighbors()
        d = icp.ICPDriver(icp.ICP3.from_method(method, cfg))
        d.set_max_iterations(12)
        d.set_transform_tolerance(1e-7, 1e-7)
        r = d.converge(src, tgt)
        out[nn] = (r, first, third, d.icp_.debug_neighbors())
    r64, f64_, t64, l64 = out["grid"]
    r32, f32_, t32, l32 = out["grid_f32"]
    # first iteration: same queries in both modes
    # (debug_neighbors' distance is sqrt of the Euclidean distance: the reference's Q2 quirk, kept)
    got = np.sqrt(exact_d2(src, tgt, f32_[0]))
    assert np.all(got <= f64_[1] ** 2 + ext * 2.0 ** -20), float(np.max(got - f64_[1] ** 2))
    assert np.mean(f32_[0] == f64_[0]) > 0.999
    # the published distance is the exact f64 distance to the chosen target (as in every other mode)
    assert np.allclose(f32_[1], np.sqrt(got), rtol=1e-14, atol=0.0)
    # later iterations (seeded queries; the running clouds differ in the sixth digit by now)
    assert np.mean(t32[0] == t64[0]) > 0.995 and np.mean(l32[0] == l64[0]) > 0.995
    assert r32.iteration_count == r64.iteration_count
    assert np.abs(r32.transform - r64.transform).max() < 2e-4, np.abs(r32.transform - r64.transform).max()


def test_f32_mode_is_refused_where_it_is_not_offered(icp):
    rng = np.random.default_rng(3)
    q, t = rng.uniform(-1, 1, (500, 2)), rng.uniform(-1, 1, (500, 2))
    with pytest.raises(icp.IcpxError):
        icp.nearest_neighbors_kdtree(q, t, icp.NN_GRID_F32)  # no kd-tree tie rule without index parity
    cfg = icp.Config()
    cfg.set("nn", "grid_f32")
    d = icp.ICPDriver(icp.ICP2.from_method("vanilla", cfg))
    d.set_max_iterations(3)
    with pytest.raises(icp.IcpxError):
        d.converge(q, t)  # 2D methods: unsupported
    cfg.set("fill_matches", 1)
    d3 = icp.ICPDriver(icp.ICP3.from_method("vanilla", cfg))
    d3.set_max_iterations(3)
    q3, t3 = rng.uniform(-1, 1, (500, 3)), rng.uniform(-1, 1, (500, 3))
    with pytest.raises(icp.IcpxError):
        d3.converge(q3, t3)  # corrected modes stay f64
