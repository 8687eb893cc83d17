"""GPU tests of the batch entry points of the C ABI beyond the fused fast path: routing of pairs the fused
kernel cannot take, feature_aware / flags in a batch (bench/bench.cpp:123 iterates every registered method),
the multi-device entry point, stream-ordered calls on several streams, the criteria check and the time
limit on the multi-kernel path."""
import ctypes as C
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def icp():
    import icp_amd
    icp_amd._lib.check(icp_amd._lib.load().icpx_init(0))
    return icp_amd


def handle_results(icp, dim, method, cfg, crit, clouds):
    cls = icp.ICP2 if dim == 2 else icp.ICP3
    out = []
    for a, b in clouds:
        d = icp.ICPDriver(cls.from_method(method, cfg))
        d.set_max_iterations(crit["max_iterations"])
        d.set_transform_tolerance(*crit["transform_tol"])
        out.append(d.converge(a, b))
    return out


def test_batch_routes_pairs_the_fused_kernel_cannot_take(icp):
    """A trajectory with one large scan: the large pair runs on the multi-kernel path, its neighbours in
    fused launches, and every pair's result equals that of its own handle (icpx_converge)."""
    from icp_amd import synth
    rng = np.random.default_rng(3)
    crit_kw = dict(max_iterations=15, transform_tol=(1e-5, 1e-5))
    crit = icp.make_criteria(**crit_kw)
    for dim, method in ((3, "vanilla"), (2, "trimmed"), (3, "trimmed")):
        sizes = [(700, 800), (900, 600), (6000, 5000), (300, 300), (5000, 40000 if dim == 3 else 12000), (1000, 1024)]
        clouds = []
        for k, (n, m) in enumerate(sizes):
            s, t, _, _ = synth.config5_pairs([100 + k], dim, max(n, m))
            clouds.append((s[:n].copy(), t[:m].copy()))
        so = np.cumsum([0] + [len(a) for a, _ in clouds])
        to = np.cumsum([0] + [len(b) for _, b in clouds])
        S = np.concatenate([a for a, _ in clouds])
        T = np.concatenate([b for _, b in clouds])
        cost, its, Ts = icp.converge_batch(dim, method, S, so, T, to, crit)
        for p, r in enumerate(handle_results(icp, dim, method, None, crit_kw, clouds)):
            assert its[p] == r.iteration_count, (dim, method, p)
            assert np.array_equal(Ts[p], r.transform), (dim, method, p)
            assert cost[p] == r.cost, (dim, method, p)
        guesses = np.tile(np.eye(dim + 1), (len(clouds), 1, 1))
        guesses[:, :dim, dim] = rng.uniform(-0.05, 0.05, (len(clouds), dim))
        cost_g, its_g, Ts_g = icp.converge_batch(dim, method, S, so, T, to, crit, guesses=guesses)
        cls = icp.ICP2 if dim == 2 else icp.ICP3
        for p in (1, 2):
            d = icp.ICPDriver(cls.from_method(method))
            d.set_max_iterations(15)
            d.set_transform_tolerance(1e-5, 1e-5)
            r = d.converge(*clouds[p], guesses[p])
            assert its_g[p] == r.iteration_count and np.array_equal(Ts_g[p], r.transform), (dim, method, p)


def test_batch_accepts_every_registered_method(icp, scans):
    """feature_aware and the opt-in flags run through the batch entry point too (one pair after the other
    on the multi-kernel path): results equal the handles'."""
    crit_kw = dict(max_iterations=20, transform_tol=(0.1 * np.pi / 180, 0.01))
    crit = icp.make_criteria(**crit_kw)
    clouds = [scans[1], scans[2], scans[3]]
    so = np.cumsum([0] + [len(a) for a, _ in clouds]).astype(np.int64)
    to = np.cumsum([0] + [len(b) for _, b in clouds]).astype(np.int64)
    S = np.concatenate([a for a, _ in clouds])
    T = np.concatenate([b for _, b in clouds])
    lib = icp._lib.load()
    p = icp._lib.Params()
    lib.icpx_default_params(C.byref(p))
    p.overlap_rate, p.symmetric_neighbors = 0.8, 5
    for method, flags in (("feature_aware", 0), ("vanilla", icp._lib.FLAG_MATCH_TRANSFORMED)):
        p.flags = flags
        states = (icp._lib.State * 3)()
        mcode = {"feature_aware": icp._lib.FEATURE_AWARE, "vanilla": icp._lib.VANILLA}[method]
        icp._lib.check(lib.icpx_converge_batch(2, mcode, C.byref(p), 3, S.ctypes.data, so.ctypes.data,
                                                T.ctypes.data, to.ctypes.data, None, C.byref(crit),
                                                C.addressof(states)))
        hcfg = icp.Config()
        hcfg.set("overlap_rate", 0.8)
        hcfg.set("symmetric_neighbors", 5)
        if flags:
            hcfg.set("match_transformed", 1)
        for k, r in enumerate(handle_results(icp, 2, method, hcfg, crit_kw, clouds)):
            assert states[k].iteration_count == r.iteration_count, (method, k)
            assert np.array_equal(np.array(states[k].transform[:9]).reshape(3, 3, order="F"), r.transform), (method, k)
            assert states[k].cost == r.cost


def test_multi_device_entry_point(icp):
    """icpx_converge_batch_multi with one shard, and with the one visible device listed two and three times
    (shards on separate streams of the same device): every layout returns the single-call results."""
    from icp_amd import synth
    lib = icp._lib.load()
    crit = icp.make_criteria(max_iterations=30, transform_tol=(1e-4, 1e-4))
    for dim, method, n, P in ((3, "vanilla", 600, 37), (2, "trimmed", 500, 50)):
        s, t, off, _ = synth.config5_pairs(range(P), dim, n)
        ref = icp.converge_batch(dim, method, s, off, t, off, crit)
        mcode = icp._lib.VANILLA if method == "vanilla" else icp._lib.TRIMMED
        for devices in ([0], [0, 0], [0, 0, 0]):
            arr = (C.c_int * len(devices))(*devices)
            states = (icp._lib.State * P)()
            icp._lib.check(lib.icpx_converge_batch_multi(dim, mcode, None, len(devices), arr, P, s.ctypes.data,
                                                          off.ctypes.data, t.ctypes.data, off.ctypes.data, None,
                                                          C.byref(crit), C.addressof(states)))
            raw = np.frombuffer(states, dtype=np.dtype([("cost", "f8"), ("it", "u8"), ("T", "f8", 16)]))
            d1 = dim + 1
            T = raw["T"][:, : d1 * d1].reshape(P, d1, d1).transpose(0, 2, 1)
            assert np.array_equal(raw["it"].astype(np.int64), ref[1]), (dim, devices)
            assert np.array_equal(T, ref[2]) and np.array_equal(raw["cost"], ref[0]), (dim, devices)
    bad = (C.c_int * 1)(7)
    states = (icp._lib.State * 4)()
    rc = lib.icpx_converge_batch_multi(2, 0, None, 1, bad, 4, s.ctypes.data, off.ctypes.data, t.ctypes.data,
                                       off.ctypes.data, None, C.byref(crit), C.addressof(states))
    assert rc == icp._lib.ERR_INVALID and b"out of range" in lib.icpx_last_error()


def test_stream_ordered_calls_on_two_streams(icp):
    """Device-resident arrays: the call returns once the work is enqueued; two streams run concurrently and
    both produce the results of the plain call."""
    import torch
    from icp_amd import synth
    crit = icp.make_criteria(max_iterations=30, transform_tol=(1e-4, 1e-4))
    s3, t3, off3, _ = synth.config5_pairs(range(40), 3, 900)
    s2, t2, off2, _ = synth.config5_pairs(range(60), 2, 700)
    ref3 = icp.converge_batch(3, "vanilla", s3, off3, t3, off3, crit)
    ref2 = icp.converge_batch(2, "trimmed", s2, off2, t2, off2, crit)
    d = [torch.from_numpy(x).cuda() for x in (s3, t3, s2, t2)]
    o3 = torch.zeros((40, 18), dtype=torch.float64, device="cuda")
    o2 = torch.zeros((60, 18), dtype=torch.float64, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        o3.zero_()
        o2.zero_()
        torch.cuda.synchronize()
        icp.converge_batch(3, "vanilla", d[0], off3, d[1], off3, crit, out_states=o3, stream=sa.cuda_stream)
        icp.converge_batch(2, "trimmed", d[2], off2, d[3], off2, crit, out_states=o2, stream=sb.cuda_stream)
        off_scratch = off3.copy()
        off_scratch[:] = -1  # the caller's offset arrays are free as soon as the call has returned
        sa.synchronize()
        sb.synchronize()
        dt = np.dtype([("cost", "f8"), ("it", "u8"), ("T", "f8", 16)])
        r3 = o3.cpu().numpy().reshape(-1).view(dt)
        r2 = o2.cpu().numpy().reshape(-1).view(dt)
        assert np.array_equal(r3["it"].astype(np.int64), ref3[1]) and np.array_equal(r3["cost"], ref3[0])
        assert np.array_equal(r2["it"].astype(np.int64), ref2[1]) and np.array_equal(r2["cost"], ref2[0])
        assert np.array_equal(r3["T"].reshape(40, 4, 4).transpose(0, 2, 1), ref3[2])
        assert np.array_equal(r2["T"][:, :9].reshape(60, 3, 3).transpose(0, 2, 1), ref2[2])


def test_criteria_that_never_terminate_are_rejected(icp, scans):
    L = icp._lib
    lib = L.load()
    a, b = scans[2]
    g = np.eye(3)
    st = L.State()
    h = L._vp()
    assert lib.icpx_create(2, L.TRIMMED, None, C.byref(h)) == L.OK
    for kw in (dict(), dict(stop_cost=0.1), dict(relative=1e-3), dict(min_iterations=9, max_iterations=3)):
        crit = icp.make_criteria(**kw)
        rc = lib.icpx_converge(h, a.ctypes.data, len(a), b.ctypes.data, len(b), g.ctypes.data, C.byref(crit), C.byref(st))
        assert rc == L.ERR_INVALID, kw
    crit = icp.make_criteria(time_limit=0.0)  # a time limit alone is enough to end the loop
    assert lib.icpx_converge(h, a.ctypes.data, len(a), b.ctypes.data, len(b), g.ctypes.data, C.byref(crit), C.byref(st)) == L.OK
    assert lib.icpx_destroy(h) == L.OK


def test_time_limit_on_the_multi_kernel_path_is_not_halved(icp):
    """the multi-kernel path counted device time twice from its second chunk on, so a time limit
    fired at about half its value.  A 1e5-point pair that would otherwise run all its iterations: the call
    must last at least the limit, and stop well before the iteration cap."""
    from icp_amd import synth
    s, t, _, _ = synth.config5_pairs([77], 3, 100_000)
    cfg = icp.Config()
    cfg.set("nn", "grid")
    d = icp.ICPDriver(icp.ICP3.from_method("vanilla", cfg))
    d.set_max_iterations(100000)
    d.set_time_limit(0.5)
    d.converge(s[:1000], t[:1000])  # warm-up of the path
    limit = 0.25
    d.set_time_limit(limit)
    t0 = time.perf_counter()
    r = d.converge(s, t)
    wall = time.perf_counter() - t0
    assert r.iteration_count < 100000
    assert wall >= limit, (wall, r.iteration_count)
    assert wall < limit + 1.0, (wall, r.iteration_count)


@pytest.mark.parametrize("dim,method", [(2, "trimmed"), (2, "vanilla"), (3, "trimmed"), (3, "vanilla")])
def test_pairs_whose_grid_is_a_poor_index_are_handed_over(icp, scans, knob, dim, method):
    """Clouds with most points in few cells of the per-pair grid (the reference's lidar scans, ex_data/scan*;
    in 3D points on a curve, or half of them in a small clump): the grid-pruned kernels hand such pairs to the
    brute-force kernel launched behind them.  A batch that mixes them with uniform clouds returns, pair by
    pair, bit for bit what the pruned kernels alone (ICPX_DEFER_SPARSE=0), the brute-force kernels alone
    (ICPX_PRUNE=0) and the pairs' own handles return."""
    from icp_amd import synth
    rng = np.random.default_rng(17)
    crit_kw = dict(max_iterations=30, transform_tol=(0.1 * np.pi / 180, 0.01))
    crit = icp.make_criteria(**crit_kw)
    clouds = []
    for k in range(12):
        if k % 3 == 2:
            s, t, _, _ = synth.config5_pairs([300 + k], dim, 900)
            clouds.append((s[:700 + 10 * k].copy(), t.copy()))
        elif dim == 2:
            a, b = scans[1 + k % 3]
            clouds.append((a, b + 0.01 * k))
        else:
            if k % 2:
                x = rng.uniform(size=1000)
                t = 5.0 * np.c_[x, np.sin(6 * x), np.cos(5 * x)]
            else:
                t = np.r_[rng.uniform(0, 5, (500, 3)), rng.uniform(2, 3, (500, 3))]
                rng.shuffle(t)
            a = np.radians(2.0 + k)
            R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            clouds.append(((t[:800] @ R.T + [0.1, -0.05, 0.02 * k]).copy(), t))
    so = np.cumsum([0] + [len(a) for a, _ in clouds]).astype(np.int64)
    to = np.cumsum([0] + [len(b) for _, b in clouds]).astype(np.int64)
    S = np.ascontiguousarray(np.concatenate([a for a, _ in clouds]))
    T = np.ascontiguousarray(np.concatenate([b for _, b in clouds]))
    got = icp.converge_batch(dim, method, S, so, T, to, crit)
    knob.set("ICPX_DEFER_SPARSE", 0)
    pruned = icp.converge_batch(dim, method, S, so, T, to, crit)
    knob.reset("ICPX_DEFER_SPARSE")
    knob.set("ICPX_PRUNE", 0)
    brute = icp.converge_batch(dim, method, S, so, T, to, crit)
    knob.reset("ICPX_PRUNE")
    for other in (pruned, brute):
        for a, b in zip(got, other):
            assert np.array_equal(a, b)
    cost, its, Ts = got
    for p, r in enumerate(handle_results(icp, dim, method, None, crit_kw, clouds)):
        assert its[p] == r.iteration_count and cost[p] == r.cost and np.array_equal(Ts[p], r.transform), p
