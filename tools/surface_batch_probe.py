"""Development aid: a batch of 3D pairs whose points lie on a sphere's surface (few occupied cells of the per-pair
grid) through icpx_converge_batch — default (sparse grids handed to the brute-force kernel), ICPX_DEFER_SPARSE=0
and ICPX_PRUNE=0 are set from the environment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from icp_amd import api
shape = sys.argv[1] if len(sys.argv) > 1 else "sphere"
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 192
rng = np.random.default_rng(5)
src, dst, so, do = [], [], [0], [0]
D = 2 if shape in ("circle", "clump2") else 3
for k in range(n_pairs):
    if shape == "sphere":
        u = rng.normal(size=(1024, 3)); t = 5.0 * u / np.linalg.norm(u, axis=1, keepdims=True)
    elif shape == "curve":
        x = rng.uniform(size=1024); t = 5.0 * np.c_[x, np.sin(6 * x), np.cos(5 * x)]
    elif shape == "circle":
        th = rng.uniform(0, 2 * np.pi, 1024); t = 5.0 * np.c_[np.cos(th), np.sin(th)]
    elif shape == "clump2":   # half the points in 1 % of the area
        t = np.r_[rng.uniform(0, 5, (512, 2)), rng.uniform(2, 2.5, (512, 2))]; rng.shuffle(t)
    elif shape == "clump3":
        t = np.r_[rng.uniform(0, 5, (512, 3)), rng.uniform(2, 3, (512, 3))]; rng.shuffle(t)
    a = np.radians(2.0 + k % 7); R = np.eye(D); R[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    src.append(t[:900] @ R.T + [0.1, -0.05, 0.02][:D]); dst.append(t); so.append(so[-1] + 900); do.append(do[-1] + 1024)
dev = torch.device("cuda", 0)
S, T = torch.from_numpy(np.concatenate(src)).to(dev), torch.from_numpy(np.concatenate(dst)).to(dev)
so, do = np.array(so, np.int64), np.array(do, np.int64)
crit = api.make_criteria(max_iterations=50, transform_tol=(0.1 * np.pi / 180, 0.01))
out = torch.zeros((n_pairs, 18), dtype=torch.float64, device=dev)
for method in ("trimmed", "vanilla"):
    api.converge_batch(D, method, S, so, T, do, crit, out_states=out, device=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        api.converge_batch(D, method, S, so, T, do, crit, out_states=out, device=0)
    torch.cuda.synchronize()
    print(f"{n_pairs} {shape} pairs, {method} ICP{D}, PRUNE={os.environ.get('ICPX_PRUNE')} DEFER={os.environ.get('ICPX_DEFER_SPARSE')}: "
          f"{(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per batch, iterations {out[:, 16].mean().item():.1f}", flush=True)
