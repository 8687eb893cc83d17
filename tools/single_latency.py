"""Development aid: latency of ONE converge call (handle API, host numpy inputs) with a fixed
iteration count.  usage: single_latency.py dim n method iters [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth
dim, n, method, iters = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
s, t, off, _ = synth.config5_pairs([3], dim, n)
cls = icp_amd.ICP2 if dim == 2 else icp_amd.ICP3
d = icp_amd.ICPDriver(cls.from_method(method, device=0))
d.set_min_iterations(iters)
d.set_max_iterations(iters)
ts = []
for r in range(reps + 1):
    t0 = time.perf_counter()
    res = d.converge(s, t)
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"dim={dim} n={n} {method} iters={res.iteration_count}: {np.median(ts[1:]):.3f} ms per converge "
      f"({np.median(ts[1:]) / max(iters, 1) * 1e3:.1f} us per iteration)  FUSED_MAX={os.environ.get('ICPX_FUSED_SINGLE_MAX')}")
