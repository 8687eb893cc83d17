"""Development aid (needs the -DICPX_STATS build): queries, certified seeds, evaluated candidates and cooperative passes of a
3D (or 2D) batch with a fixed number of iterations; differences between runs give per-iteration figures.
usage: ICPX_LIB=icp_amd/lib_stats/libicpx.so cert_stats.py dim n pairs method"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth
dim, n, pairs, method = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
lib = icp_amd._lib.load()
s, t, off, _ = synth.config5_pairs(np.arange(pairs), dim, n)
out = (C.c_ulonglong * 32)()
prev = None
for iters in (1, 2, 3, 4, 5, 6, 8):
    crit = icp_amd.make_criteria(min_iterations=iters, max_iterations=iters)
    lib.icpx_debug_stats(out, 1)
    icp_amd.converge_batch(dim, method, s, off, t, off, crit, device=0)
    lib.icpx_debug_stats(out, 0)
    cur = np.array([out[29], out[30], out[26], out[28], out[23]], dtype=np.int64)
    d = cur if prev is None else cur - prev
    prev = cur
    print(f"up to iteration {iters}: +queries {d[0]}, certified {d[1]} ({d[1] / max(1, d[0]):.2f}), candidates {d[2]} "
          f"({d[2] / max(1, d[0] - d[1]):.1f} per searched query), coop passes {d[3]} ({d[3] * 64 / max(1, d[0]):.2f} per query-slot), walks {d[4]}")
