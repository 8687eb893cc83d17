"""Quick GPU sanity run (development aid): product vs oracle on the example data."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import icp_amd
from icp_amd import io as iio
from oracle import binding as ob

G = "tests/golden/ex_data"
def scans(k):
    return iio.parse_lidar_scan(f"{G}/scan{k}/first.csv"), iio.parse_lidar_scan(f"{G}/scan{k}/second.csv")

print("init", icp_amd._lib.load().icpx_init(-1), flush=True)
a, b = scans(1)
t = time.time(); idx, d2 = icp_amd.nearest_neighbors(a, b, icp_amd.NN_BRUTE); print("nn brute", time.time() - t, flush=True)
oi, od = ob.nn_brute(a, b)
print("brute idx equal", np.array_equal(idx, oi), "d2 equal", np.array_equal(d2, od), flush=True)
idx, d2 = icp_amd.nearest_neighbors(a, b, icp_amd.NN_GRID)
print("grid idx equal", np.array_equal(idx, oi), "d2 equal", np.array_equal(d2, od), flush=True)

crit_kw = dict(max_iterations=50, transform_tol=(0.1 * np.pi / 180, 0.01))
for method in ("vanilla", "trimmed"):
    for k in (1, 2, 3):
        a, b = scans(k)
        icp = icp_amd.ICP2.from_method(method)
        drv = icp_amd.ICPDriver(icp)
        drv.set_max_iterations(50); drv.set_transform_tolerance(0.1 * np.pi / 180, 0.01)
        t = time.time(); r = drv.converge(a, b); dt = time.time() - t
        o = ob.OracleICP(2, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64)
        ro = o.converge(a, b, None, ob.make_criteria(**crit_kw))
        print(method, k, "iters", r.iteration_count, ro.iteration_count, "cost", r.cost, ro.cost,
              "T bitexact", np.array_equal(r.transform, ro.transform), "maxdiff", np.abs(r.transform - ro.transform).max(), f"{dt*1e3:.2f}ms", flush=True)
        p, q, c = icp.get_matches_arrays(); po, qo, co = o.get_matches()
        print("   matches equal", np.array_equal(p, po), np.array_equal(q, qo), np.array_equal(c, co), flush=True)

q = iio.read_ply_vertices(f"{G}/ply/queen.ply"); qt = iio.read_ply_vertices(f"{G}/ply/queen_transformed.ply")
idx, d2 = icp_amd.nearest_neighbors(qt, q, icp_amd.NN_BRUTE); oi, od = ob.nn_brute(qt, q)
print("queen brute idx equal", np.array_equal(idx, oi), np.array_equal(d2, od), flush=True)
idx, d2 = icp_amd.nearest_neighbors(qt, q, icp_amd.NN_GRID)
print("queen grid idx equal", np.array_equal(idx, oi), np.array_equal(d2, od), flush=True)
for method in ("vanilla", "trimmed"):
    icp = icp_amd.ICP3.from_method(method)
    drv = icp_amd.ICPDriver(icp); drv.set_max_iterations(100); drv.set_transform_tolerance(1e-4, 1e-4)
    t = time.time(); r = drv.converge(qt, q); dt = time.time() - t
    o = ob.OracleICP(3, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64)
    ro = o.converge(qt, q, None, ob.make_criteria(max_iterations=100, transform_tol=(1e-4, 1e-4)))
    print("queen", method, "iters", r.iteration_count, ro.iteration_count, "T bitexact", np.array_equal(r.transform, ro.transform),
          "maxdiff", np.abs(r.transform - ro.transform).max(), f"{dt*1e3:.2f}ms", flush=True)

# small 3D through the fused kernel
rng = np.random.default_rng(0)
b3 = rng.uniform(-10, 10, (1500, 3)); th = 0.05
R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
a3 = (b3[:1400] - [0.1, 0.05, -0.02]) @ R
for method in ("vanilla", "trimmed"):
    icp = icp_amd.ICP3.from_method(method)
    drv = icp_amd.ICPDriver(icp); drv.set_max_iterations(30); drv.set_transform_tolerance(1e-6, 1e-6)
    r = drv.converge(a3, b3)
    o = ob.OracleICP(3, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64)
    ro = o.converge(a3, b3, None, ob.make_criteria(max_iterations=30, transform_tol=(1e-6, 1e-6)))
    print("fused3d", method, "iters", r.iteration_count, ro.iteration_count, "T bitexact", np.array_equal(r.transform, ro.transform),
          "maxdiff", np.abs(r.transform - ro.transform).max(), flush=True)
print("done")
