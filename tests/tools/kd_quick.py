"""Development check of the kd-tree tie rule on the GPU (fused kernels): planted ties against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import icp_amd  # noqa: E402
from conftest import tie_clouds  # noqa: E402
from oracle import binding as ob  # noqa: E402

icp_amd._lib.check(icp_amd._lib.load().icpx_init(0))
fails = 0


def check(name, cond, extra=""):
    global fails
    print(("ok   " if cond else "FAIL ") + name, extra, flush=True)
    fails += 0 if cond else 1


# A: 2D vanilla, fused handle
for seed, (nr, mr, pl) in enumerate([(500, 900, 120), (100, 200, 30), (800, 1500, 120), (3000, 3500, 200)]):
    q, t, pi = tie_clouds(2, nr, mr, pl, 10 + seed)
    x = icp_amd.ICP2.from_method("vanilla")
    x.begin(q, t)
    p, pair, c = x.get_matches_arrays()
    ci, cd = ob.nn_kdcanon(q, t)
    ri, rd = ob.ref_nn_kdtree(q, t) if ob.ref_lib() is not None else (ci, cd)
    check(f"A{seed} 2D vanilla begin matches == canon", np.array_equal(pair, ci), f"diff {(pair != ci).sum()} of {len(q)}")
    check(f"A{seed} == real kdtree.h", np.array_equal(pair, ri))
    check(f"A{seed} cost", np.array_equal(c, np.sqrt(cd)))
    o = ob.OracleICP(2, "vanilla", nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64)
    crit = dict(max_iterations=20, transform_tol=(1e-6, 1e-6))
    d = icp_amd.ICPDriver(icp_amd.ICP2.from_method("vanilla"))
    d.set_max_iterations(20)
    d.set_transform_tolerance(1e-6, 1e-6)
    r = d.converge(q, t)
    ro = o.converge(q, t, None, ob.make_criteria(**crit))
    check(f"A{seed} converge", r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform))

# B: 3D handles (plain kernels + take-over)
for method in ("vanilla", "trimmed"):
    for seed, (nr, mr, pl) in enumerate([(500, 900, 60), (90, 150, 20)]):
        q, t, pi = tie_clouds(3, nr, mr, pl, 20 + seed)
        cfg = icp_amd.Config()
        cfg.set("max_distance", 0.5)
        x = icp_amd.ICP3.from_method(method, cfg)
        o = ob.OracleICP(3, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64, max_distance=0.5)
        x.begin(q, t)
        o.begin(q, t)
        for it in range(4):
            x.iterate()
            o.iterate()
            xi, xd = x.debug_neighbors()
            oi, od = o.neighbors()
            check(f"B {method} {seed} it{it} neighbours", np.array_equal(xi, oi) and np.array_equal(xd, od),
                  f"diff {(xi != oi).sum()}")
            check(f"B {method} {seed} it{it} transform", np.array_equal(x.current_transform(), o.current_transform()))
        crit = dict(max_iterations=15, transform_tol=(1e-7, 1e-7))
        d = icp_amd.ICPDriver(icp_amd.ICP3.from_method(method, cfg))
        d.set_max_iterations(15)
        d.set_transform_tolerance(1e-7, 1e-7)
        r = d.converge(q, t)
        ro = o.converge(q, t, None, ob.make_criteria(**crit))
        check(f"B {method} {seed} converge", r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform),
              f"{r.iteration_count} {ro.iteration_count}")

SKIP_ABC = os.environ.get('KD_SKIP_ABC') == '1'
# C: 3D batches (pruned kernels + take-over); some pairs with ties, some without
from icp_amd import synth  # noqa: E402

for method in ("vanilla", "trimmed"):
    srcs, dsts = [], []
    for k in range(12):
        if k % 3 == 0:
            s, t, _, _ = synth.config5_pairs([100 + k], 3, 700)
        else:
            s, t, _ = tie_clouds(3, 600, 800, 40, 30 + k)
        srcs.append(s)
        dsts.append(t)
    so = np.concatenate([[0], np.cumsum([len(s) for s in srcs])])
    to = np.concatenate([[0], np.cumsum([len(t) for t in dsts])])
    crit = dict(max_iterations=12, transform_tol=(1e-7, 1e-7))
    cfg = icp_amd.Config()
    cfg.set("max_distance", 0.5)
    cost, its, T = icp_amd.converge_batch(3, method, np.concatenate(srcs), so, np.concatenate(dsts), to,
                                          icp_amd.make_criteria(**crit), cfg)
    for k in range(12):
        ro = ob.OracleICP(3, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64, max_distance=0.5).converge(
            srcs[k], dsts[k], None, ob.make_criteria(**crit))
        check(f"C {method} pair {k}", its[k] == ro.iteration_count and np.array_equal(T[k], ro.transform), f"{its[k]} {ro.iteration_count}")

# D: the stand-alone kd-tree query, every engine
for dim in (2, 3):
    q, t, pi = tie_clouds(dim, 3000, 5000, 300, 40 + dim)
    ri, rd = ob.ref_nn_kdtree(q, t) if ob.ref_lib() is not None else ob.nn_kdcanon(q, t)
    for mode, name in ((icp_amd.NN_AUTO, "auto"), (icp_amd.NN_BRUTE, "brute"), (icp_amd.NN_GRID, "grid"), (icp_amd.NN_BVH, "bvh")):
        idx, dist = icp_amd.nearest_neighbors_kdtree(q, t, mode)
        check(f"D {dim}D {name} == real kdtree.h", np.array_equal(idx, ri) and np.array_equal(dist, rd), f"diff {(idx != ri).sum()}")
    lat = np.random.default_rng(1).integers(0, 5, (800, dim)).astype(float)
    ci, cd = ob.nn_kdcanon(lat[:300] + 0.5, lat)
    for mode, name in ((icp_amd.NN_BRUTE, "brute"), (icp_amd.NN_GRID, "grid"), (icp_amd.NN_BVH, "bvh")):
        idx, dist = icp_amd.nearest_neighbors_kdtree(lat[:300] + 0.5, lat, mode)
        check(f"D {dim}D lattice {name} == canon", np.array_equal(idx, ci) and np.array_equal(dist, cd), f"diff {(idx != ci).sum()}")

# E: large path (multi-kernel): 2D vanilla and 3D, tiled / grid / bvh
for nn in ("auto", "grid", "bvh"):
    q, t, pi = tie_clouds(2, 5000, 6000, 200, 50)
    cfg = icp_amd.Config()
    cfg.set("nn", nn)
    x = icp_amd.ICP2.from_method("vanilla", cfg)
    x.begin(q, t)
    p, pair, c = x.get_matches_arrays()
    ci, cd = ob.nn_kdcanon(q, t)
    check(f"E 2D vanilla large {nn} begin", np.array_equal(pair, ci), f"diff {(pair != ci).sum()}")
    for method in ("vanilla", "trimmed"):
        q, t, pi = tie_clouds(3, 5000, 6000, 200, 60)
        cfg.set("max_distance", 0.5)
        x = icp_amd.ICP3.from_method(method, cfg)
        o = ob.OracleICP(3, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64, max_distance=0.5)
        x.begin(q, t)
        o.begin(q, t)
        for it in range(3):
            x.iterate()
            o.iterate()
            xi, xd = x.debug_neighbors()
            oi, od = o.neighbors()
            check(f"E 3D {method} {nn} it{it}", np.array_equal(xi, oi) and np.array_equal(x.current_transform(), o.current_transform()),
                  f"diff {(xi != oi).sum()}")
        crit = dict(max_iterations=10, transform_tol=(1e-7, 1e-7))
        d = icp_amd.ICPDriver(icp_amd.ICP3.from_method(method, cfg))
        d.set_max_iterations(10)
        d.set_transform_tolerance(1e-7, 1e-7)
        r = d.converge(q, t)
        ro = o.converge(q, t, None, ob.make_criteria(**crit))
        check(f"E 3D {method} {nn} converge", r.iteration_count == ro.iteration_count and np.array_equal(r.transform, ro.transform),
              f"{r.iteration_count} {ro.iteration_count}")

print("FAILS", fails)
sys.exit(1 if fails else 0)
