"""Development aid: BASELINE config 3 (queen_transformed.ply -> queen.ply, vanilla ICP3, one pair) through
ICPDriver.converge as bench.py's extras run it (host arrays).  usage: queen_probe.py [reps [nn]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icp_amd
from icp_amd import io
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ex_data", "ply")
q = io.read_ply_vertices(os.path.join(g, "queen.ply"))
qt = io.read_ply_vertices(os.path.join(g, "queen_transformed.ply"))
torch.cuda.is_available()
cfg = icp_amd.Config()
if len(sys.argv) > 2:
    cfg.set("nn", sys.argv[2])
d = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", cfg, device=0))
d.set_max_iterations(100)
d.set_transform_tolerance(1e-4, 1e-4)
r = d.converge(qt, q)
t0 = time.perf_counter()
for _ in range(reps):
    r = d.converge(qt, q)
print(f"queen ({len(q)} points) vanilla ICP3 converge: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms, {r.iteration_count} iterations, nn = {sys.argv[2] if len(sys.argv) > 2 else 'auto'}")
