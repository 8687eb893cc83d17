"""dev script: config 3 (queen) and config 2 (scan1) single-pair timings, with the kd-rule service's debug output"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import icp_amd
from icp_amd import io as iio
g = os.path.join(ROOT, "tests", "golden", "ex_data")
q = iio.read_ply_vertices(os.path.join(g, "ply", "queen.ply"))
qt = iio.read_ply_vertices(os.path.join(g, "ply", "queen_transformed.ply"))
print("queen", q.shape, qt.shape, "unique rows", len(np.unique(q, axis=0)), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    sys.exit(0)
for method in ("vanilla", "trimmed"):
    drv3 = icp_amd.ICPDriver(icp_amd.ICP3.from_method(method, device=0))
    drv3.set_max_iterations(100)
    drv3.set_transform_tolerance(1e-4, 1e-4)
    r = drv3.converge(qt, q)
    os.environ.pop("ICPX_KD_DEBUG", None)
    t0 = time.perf_counter()
    for _ in range(10):
        r = drv3.converge(qt, q)
    print(method, "queen ms", (time.perf_counter() - t0) / 10 * 1e3, "it", r.iteration_count, flush=True)
for nn in ("grid", "bvh"):
    cfg = icp_amd.Config()
    cfg.set("nn", nn)
    drv3 = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", cfg, device=0))
    drv3.set_max_iterations(100)
    drv3.set_transform_tolerance(1e-4, 1e-4)
    r0 = drv3.converge(qt, q)
    t0 = time.perf_counter()
    for _ in range(10):
        r = drv3.converge(qt, q)
    print("vanilla nn=%s queen ms" % nn, (time.perf_counter() - t0) / 10 * 1e3, "it", r.iteration_count,
          "same as auto:", np.array_equal(r.transform, r0.transform), flush=True)
