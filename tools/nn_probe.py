"""Development aid: stand-alone correspondence query throughput (icpx_nn), device-resident data.
usage: nn_probe.py dim n m mode(brute|grid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
dim, n, m, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
q = torch.from_numpy(rng.uniform(-10, 10, (n, dim))).to(dev)
t = torch.from_numpy(rng.uniform(-10, 10, (m, dim))).to(dev)
idx = torch.zeros(n, dtype=torch.int64, device=dev)
d2 = torch.zeros(n, dtype=torch.float64, device=dev)
md = icp_amd.NN_BRUTE if mode == "brute" else icp_amd.NN_GRID
for _ in range(2):
    icp_amd.nearest_neighbors(q, t, md, device=0, out_idx=idx, out_d2=d2)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 10
for _ in range(R):
    icp_amd.nearest_neighbors(q, t, md, device=0, out_idx=idx, out_d2=d2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print(f"dim={dim} n={n} m={m} {mode} FILTER={os.environ.get('ICPX_FILTER')}: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} M queries/s  "
      f"{n*m*(3*dim-1)/dt/1e12:.2f} TFLOP/s algorithmic")
