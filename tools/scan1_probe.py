"""Development aid: BASELINE config 2 (ex_data/scan1, trimmed ICP2, one pair) through ICPDriver.converge, host arrays.
usage: scan1_probe.py [reps [method]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import io
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
method = sys.argv[2] if len(sys.argv) > 2 else "trimmed"
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ex_data", "scan1")
a = io.parse_lidar_scan(os.path.join(g, "first.csv")); b = io.parse_lidar_scan(os.path.join(g, "second.csv"))
torch.cuda.is_available()
d = icp_amd.ICPDriver(icp_amd.ICP2.from_method(method, device=0))
d.set_max_iterations(50); d.set_transform_tolerance(0.1 * np.pi / 180, 0.01)
for _ in range(40):
    r = d.converge(a, b)
t0 = time.perf_counter()
for _ in range(reps):
    r = d.converge(a, b)
print(f"scan1 ({len(a)} x {len(b)}) {method} ICP2 converge: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms, {r.iteration_count} iterations")
