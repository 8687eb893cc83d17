"""Latency probe for single-pair converge calls (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import io as iio, synth
G = "tests/golden/ex_data"
which = sys.argv[1] if len(sys.argv) > 1 else "queen"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if which == "queen":
    src = iio.read_ply_vertices(f"{G}/ply/queen_transformed.ply"); dst = iio.read_ply_vertices(f"{G}/ply/queen.ply")
    drv = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla")); drv.set_max_iterations(100); drv.set_transform_tolerance(1e-4, 1e-4)
elif which == "c4":
    src, dst, _, _ = synth.config4(1_000_000)
    drv = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla")); drv.set_max_iterations(10)
else:
    src = iio.parse_lidar_scan(f"{G}/scan1/first.csv"); dst = iio.parse_lidar_scan(f"{G}/scan1/second.csv")
    drv = icp_amd.ICPDriver(icp_amd.ICP2.from_method("trimmed")); drv.set_max_iterations(50); drv.set_transform_tolerance(0.1*np.pi/180, 0.01)
for r in range(reps):
    t = time.perf_counter(); res = drv.converge(src, dst); dt = time.perf_counter() - t
    print(which, r, res.iteration_count, f"{dt*1e3:.3f} ms", flush=True)
