"""Development aid: config 4 converge with the source cloud pre-sorted in space on the host (the query order of the
grid search is then nearly the identity): how much of the correspondence kernel's time is the gather by that order?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import icp_amd
from icp_amd import synth
dev = torch.device("cuda", 0)
src, tgt, _, _ = synth.config4(1_000_000)
def morton(p, bits=7):
    lo, hi = p.min(0), p.max(0)
    c = np.minimum(((p - lo) / (hi - lo) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    key = np.zeros(len(p), np.int64)
    for b in range(bits):
        for d in range(3):
            key |= ((c[:, d] >> b) & 1) << (3 * b + d)
    return key
for name, s in (("as generated", src), ("source pre-sorted (Morton, 7 bits)", src[np.argsort(morton(src), kind="stable")])):
    q, t = torch.from_numpy(np.ascontiguousarray(s)).to(dev), torch.from_numpy(tgt).to(dev)
    d = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", device=0))
    d.set_max_iterations(50)
    d.set_transform_tolerance(1e-4, 1e-4)
    r = d.converge(q, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = d.converge(q, t)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) * 1e3:.2f} ms, {r.iteration_count} iterations", flush=True)
