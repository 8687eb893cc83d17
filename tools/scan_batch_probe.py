"""Development aid: a batch of copies of the reference's scan pairs (ex_data scan1..3, trimmed ICP2) through
icpx_converge_batch — the grid-pruned kernels (default) against the brute-force filter kernels (ICPX_PRUNE=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import api, io as iio
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ex_data")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
src, dst, so, do = [], [], [0], [0]
for r in range(reps):
    for s in ("scan1", "scan2", "scan3"):
        a = iio.parse_lidar_scan(os.path.join(g, s, "first.csv")); b = iio.parse_lidar_scan(os.path.join(g, s, "second.csv"))
        src.append(a); dst.append(b); so.append(so[-1] + len(a)); do.append(do[-1] + len(b))
dev = torch.device("cuda", 0)
S, T = torch.from_numpy(np.concatenate(src)).to(dev), torch.from_numpy(np.concatenate(dst)).to(dev)
so, do = np.array(so, np.int64), np.array(do, np.int64)
crit = api.make_criteria(max_iterations=50, transform_tol=(0.1 * np.pi / 180, 0.01))
out = torch.zeros((len(so) - 1, 18), dtype=torch.float64, device=dev)
for method in ("trimmed", "vanilla"):
    api.converge_batch(2, method, S, so, T, do, crit, out_states=out, device=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        api.converge_batch(2, method, S, so, T, do, crit, out_states=out, device=0)
    torch.cuda.synchronize()
    print(f"{len(so) - 1} scan pairs, {method} ICP2, ICPX_PRUNE={os.environ.get('ICPX_PRUNE')}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per batch", flush=True)
