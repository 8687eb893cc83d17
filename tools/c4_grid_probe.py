"""Development aid / profile target: BASELINE config 4's correspondence step alone — icpx_nn (uniform grid) on the
1 M x 1 M synthetic cloud of bench.py's extras, inputs and outputs resident in HBM.  Run under rocprofv3
(tools/collect_profiles.sh) for the grid kernels' durations and FETCH_SIZE / WRITE_SIZE.
usage: c4_grid_probe.py [reps [clustered]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icp_amd
from icp_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
clustered = len(sys.argv) > 2
src, tgt, _, _ = synth.config4(1_000_000, clustered=clustered)
q, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
idx = torch.zeros(len(src), dtype=torch.int64, device=dev)
d2 = torch.zeros(len(src), dtype=torch.float64, device=dev)
icp_amd.nearest_neighbors(q, t, icp_amd.NN_GRID, device=0, out_idx=idx, out_d2=d2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    icp_amd.nearest_neighbors(q, t, icp_amd.NN_GRID, device=0, out_idx=idx, out_d2=d2)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
tag = " (clustered)" if clustered else ""
print(f"config 4{tag} grid NN, 1M x 1M: {dt * 1e3:.3f} ms per correspondence step (build + query), {len(src) / dt / 1e6:.1f} M queries/s")
