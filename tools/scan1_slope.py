"""Development aid: latency of ONE converge call on ex_data/scan1 (trimmed ICP2, host arrays, the handle API) at
fixed iteration counts: the slope is the cost of an iteration, the intercept that of the call.
usage: scan1_slope.py [method]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import io as iio
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ex_data", "scan1")
a, b = iio.parse_lidar_scan(os.path.join(g, "first.csv")), iio.parse_lidar_scan(os.path.join(g, "second.csv"))
method = sys.argv[1] if len(sys.argv) > 1 else "trimmed"
for iters in (1, 6, 11, 22, 44):
    d = icp_amd.ICPDriver(icp_amd.ICP2.from_method(method, device=0))
    d.set_min_iterations(iters)
    d.set_max_iterations(iters)
    ts = []
    for r in range(8):
        t0 = time.perf_counter()
        res = d.converge(a, b)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"scan1 {len(a)} x {len(b)} {method}: {res.iteration_count} iterations {np.median(ts[2:]):.3f} ms", flush=True)
