"""Development aid: one correspondence step (build + query) on BASELINE config 4's cloud, grid against BVH.
usage: bvh_probe.py [clustered]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icp_amd
from icp_amd import synth
dev = torch.device("cuda", 0)
clustered = len(sys.argv) > 1
src, tgt, _, _ = synth.config4(1_000_000, clustered=clustered)
q, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
idx = torch.zeros(len(src), dtype=torch.int64, device=dev)
d2 = torch.zeros(len(src), dtype=torch.float64, device=dev)
for name, mode in (("grid", icp_amd.NN_GRID), ("bvh", icp_amd.NN_BVH)):
    icp_amd.nearest_neighbors(q, t, mode, device=0, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        icp_amd.nearest_neighbors(q, t, mode, device=0, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"config 4{' (clustered)' if clustered else ''} {name}: {dt * 1e3:.3f} ms per step (build + query), {len(src) / dt / 1e6:.1f} M queries/s")
