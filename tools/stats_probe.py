"""Development aid: run one batch against a -DICPX_STATS build and print the kernel counters.
usage: stats_probe.py dim n pairs iters method"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth
dim, n, pairs, iters, method = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
lib = icp_amd._lib.load()
s, t, off, _ = synth.config5_pairs(np.arange(pairs), dim, n)
crit = icp_amd.make_criteria(min_iterations=iters, max_iterations=iters)
out = (C.c_ulonglong * 8)()
lib.icpx_debug_stats(out, 1)
icp_amd.converge_batch(dim, method, s, off, t, off, crit, device=0)
lib.icpx_debug_stats(out, 0)
names = ["seeded scans (wave)", "exact flushes (wave)", "two-min scans (wave)", "near-tie lanes", "slot rescans (wave)"]
for i, nm in enumerate(names):
    print(f"{nm}: {out[i]}")
waves = pairs * ((n + 63) // 64) / max(1, 1)
print("queries", pairs * n, "wave-slots per scan", pairs * n // 64)
