import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import io
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ex_data", "ply")
q = io.read_ply_vertices(os.path.join(g, "queen.ply")); qt = io.read_ply_vertices(os.path.join(g, "queen_transformed.ply"))
torch.cuda.is_available()
d = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", device=0))
d.set_max_iterations(100); d.set_transform_tolerance(1e-4, 1e-4)
ts = []
for _ in range(40):
    t0 = time.perf_counter(); r = d.converge(qt, q); ts.append((time.perf_counter() - t0) * 1e3)
print("host arrays:", " ".join(f"{t:.2f}" for t in ts))
dq, dqt = torch.from_numpy(q).cuda(), torch.from_numpy(qt).cuda()
ts = []
for _ in range(40):
    t0 = time.perf_counter(); r = d.converge(dqt, dq); ts.append((time.perf_counter() - t0) * 1e3)
print("device arrays:", " ".join(f"{t:.2f}" for t in ts))
