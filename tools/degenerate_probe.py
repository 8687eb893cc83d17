"""Development aid: two tight clusters far apart (200 000 points each way) — the uniform grid's degenerate case —
grid (forced) against BVH (forced) against auto.  One correspondence step, build + query."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
n = 100_000
a = np.concatenate([rng.normal(0.0, 0.004, (n, 3)), rng.normal(100.0, 0.004, (n, 3))])
b = np.concatenate([rng.normal(0.0, 0.004, (n, 3)), rng.normal(100.0, 0.004, (n, 3))])
q, t = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
idx = torch.zeros(len(a), dtype=torch.int64, device=dev)
d2 = torch.zeros(len(a), dtype=torch.float64, device=dev)
ref = None
for name, mode in (("bvh", icp_amd.NN_BVH), ("auto", icp_amd.NN_AUTO), ("grid", icp_amd.NN_GRID)):
    icp_amd.nearest_neighbors(q, t, mode, device=0, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    icp_amd.nearest_neighbors(q, t, mode, device=0, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = "" if ref is None else f"  identical to bvh: {bool(torch.equal(idx, ref[0]) and torch.equal(d2, ref[1]))}"
    if ref is None:
        ref = (idx.clone(), d2.clone())
    print(f"two clusters, 200k x 200k, {name}: {dt * 1e3:.3f} ms per step{same}", flush=True)
