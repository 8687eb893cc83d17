"""Perf probe (development aid): fixed-iteration batches so that every workgroup does equal work.
usage: perf_probe.py [dim] [n] [pairs] [iters] [method] [reps] [m]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import synth

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 256
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
method = sys.argv[5] if len(sys.argv) > 5 else "vanilla"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
m = int(sys.argv[7]) if len(sys.argv) > 7 else n
dev = torch.device("cuda", 0)
s, t, off, _ = synth.config5_pairs(np.arange(pairs), dim, max(n, m))
offs, offt = off, off
if m != n:
    # ragged probe: keep the first n source / first m target points of every pair
    big = max(n, m)
    s = s.reshape(pairs, big, dim)[:, :n].reshape(-1, dim).copy()
    t = t.reshape(pairs, big, dim)[:, :m].reshape(-1, dim).copy()
    offs = np.arange(pairs + 1, dtype=np.int64) * n
    offt = np.arange(pairs + 1, dtype=np.int64) * m
ds, dt = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
out = torch.zeros(pairs * 18, dtype=torch.float64, device=dev)
crit = icp_amd.make_criteria(min_iterations=iters, max_iterations=iters)
stream = torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for r in range(reps + 1):
    e0.record()
    icp_amd.converge_batch(dim, method, ds, offs, dt, offt, crit, out_states=out, device=0, stream=stream)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = float(np.median(ts[1:]))
nn_steps = iters + (1 if (dim == 2 and method == "trimmed") else 0)
evals = pairs * n * m * nn_steps
flops = evals * (3 * dim - 1)
print(f"dim={dim} n={n} m={m} pairs={pairs} iters={iters} {method}: {ms:.3f} ms  {evals/ms/1e6:.1f} G pair-evals/s  "
      f"{flops/ms/1e9:.2f} TFLOP/s  per-iteration {ms/max(nn_steps,1)*1e3:.1f} us  all={['%.2f' % x for x in ts]}", flush=True)
