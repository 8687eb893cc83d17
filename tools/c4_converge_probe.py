"""Development aid: BASELINE config 4 full converge (vanilla ICP3, 50 iterations), device-resident clouds.
usage: c4_converge_probe.py [clustered] [f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import icp_amd
from icp_amd import synth
dev = torch.device("cuda", 0)
src, tgt, _, _ = synth.config4(1_000_000, clustered="clustered" in sys.argv)
q, t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
cfg = icp_amd.Config()
if "f32" in sys.argv:
    cfg.set("nn", "grid_f32")  # opt-in f32 correspondence mode
d = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", cfg, device=0))
d.set_max_iterations(50)
d.set_transform_tolerance(1e-4, 1e-4)
r = d.converge(q, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
r = d.converge(q, t)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"config 4 converge{' (clustered)' if 'clustered' in sys.argv else ''}{' f32 mode' if 'f32' in sys.argv else ''}: {dt * 1e3:.2f} ms, {r.iteration_count} iterations, "
      f"{r.iteration_count * len(src) / dt / 1e9:.3f} G queries/s  GRID_COOP={os.environ.get('ICPX_GRID_COOP')}")
