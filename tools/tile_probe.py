"""Development aid: the tiled multi-kernel path (nn = auto below ~3e4 x 3e4) on synthetic clouds of a given size.
usage: tile_probe.py n [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icp_amd
from icp_amd import synth
n = int(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
src, tgt, _, _ = synth.config4(n)
dev = torch.device("cuda", 0)
q, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
d = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", device=0))
d.set_max_iterations(iters)
d.set_min_iterations(iters)
r = d.converge(q, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    r = d.converge(q, t)
torch.cuda.synchronize()
print(f"tiled path {n} x {n}, vanilla ICP3: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms for {r.iteration_count} iterations")
