"""Development aid (stats build: ICPX_LIB=icp_amd/lib_stats/libicpx.so): where an iteration of the single-pair
kernel spends its time on ex_data/scan1 (trimmed ICP2): phase clocks 8..11 of g_icpx_stats (100 MHz ticks)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import io as iio, _lib
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ex_data", "scan1")
a, b = iio.parse_lidar_scan(os.path.join(g, "first.csv")), iio.parse_lidar_scan(os.path.join(g, "second.csv"))
L = _lib.load()
d = icp_amd.ICPDriver(icp_amd.ICP2.from_method("trimmed", device=0))
d.set_max_iterations(22); d.set_min_iterations(22)
d.converge(a, b)
buf = (C.c_ulonglong * 32)()
L.icpx_debug_stats(buf, 1)
r = d.converge(a, b)
L.icpx_debug_stats(buf, 0)
v = np.array(buf[:], dtype=np.float64)
it = r.iteration_count
print("iterations", it)
for s, name in ((8, "search"), (9, "selection"), (10, "sums"), (11, "kabsch+broadcast")):
    print(f"  {name:18s} {v[s] / 100.0 / it:7.2f} us per iteration")
for s, name in ((12, "staging+boxes+state"), (13, "setup search+cost"), (14, "loop"), (15, "publish")):
    print(f"  {name:22s} {v[s] / 100.0:7.2f} us per call")
print("all slots", [int(x) for x in v])
