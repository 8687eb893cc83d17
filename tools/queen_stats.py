import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import io
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ex_data", "ply")
q = io.read_ply_vertices(os.path.join(g, "queen.ply")); qt = io.read_ply_vertices(os.path.join(g, "queen_transformed.ply"))
torch.cuda.is_available()
lib = icp_amd._lib.load()
d = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", device=0))
d.set_max_iterations(100); d.set_transform_tolerance(1e-4, 1e-4)
out = (C.c_ulonglong * 32)()
d.converge(qt, q)
lib.icpx_debug_stats(out, 1)
r = d.converge(qt, q)
lib.icpx_debug_stats(out, 0)
print("iterations", r.iteration_count, "seeded scans (wave)", out[0], "exact flushes (wave)", out[1], "two-min scans", out[2], "near-tie lanes", out[3], "rescans", out[4])
print("flushes per seeded scan", out[1] / max(out[0], 1))
