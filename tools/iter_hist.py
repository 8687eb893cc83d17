"""Development aid: iteration-count histogram of the bench's 3D / 2D shards."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import synth
dev = torch.device("cuda", 0)
crit = icp_amd.make_criteria(max_iterations=50, transform_tol=(1e-4, 1e-4))
for dim, n, method, ids in ((3, 4096, "vanilla", np.arange(0, 512)), (2, 1024, "trimmed", np.arange(0, 512))):
    s, t, off, _ = synth.config5_pairs(ids, dim, n)
    _, it, _ = icp_amd.converge_batch(dim, method, s, off, t, off, crit, device=0)
    print(dim, method, "hist", np.bincount(it), "max", it.max(), "mean", it.mean())
