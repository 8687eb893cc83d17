"""Development aid: does sorting the queries by grid cell speed up the grid query? (config 4)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import synth
dev = torch.device("cuda", 0)
src, tgt, _, _ = synth.config4(1_000_000)
lib = icp_amd._lib.load()
icp_amd._lib.check(lib.icpx_init(0))
prof = (C.c_double * 4)()
d_t = torch.from_numpy(tgt).to(dev)
def run(q, name):
    d_q = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
    best = 1e9
    for _ in range(4):
        icp_amd._lib.check(lib.icpx_nn_grid_profile(3, d_q.data_ptr(), len(q), d_t.data_ptr(), len(tgt), 0, None, prof))
        best = min(best, prof[1])
    print(f"{name}: query {best:.3f} ms, candidates {prof[2]:.1f}, cells {int(prof[3])}")
run(src, "random order")
h = 20.0 / 80
cell = np.floor((src + 10) / h).astype(np.int64).clip(0, 79)
key = (cell[:, 2] * 80 + cell[:, 1]) * 80 + cell[:, 0]
run(src[np.argsort(key, kind="stable")], "sorted by cell (z,y,x)")
