import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import synth
dev = torch.device("cuda", 0)
P, n = 512, 1024
s, t, off, _ = synth.config5_pairs(np.arange(P), 2, n)
crit = icp_amd.make_criteria(min_iterations=0, max_iterations=0)
out = torch.zeros((P, 18), dtype=torch.float64, device=dev)
src = t + 0.3 if sys.argv[1] == "fast" else t * 0.9
ds, dt = torch.from_numpy(src).to(dev), torch.from_numpy(t).to(dev)
for _ in range(3):
    icp_amd.converge_batch(2, "trimmed", ds, off, dt, off, crit, out_states=out, device=0)
torch.cuda.synchronize()
