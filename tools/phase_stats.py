"""Development aid (needs a -DICPX_STATS build): time per phase of the fused iteration (thread 0 of every
workgroup, 100 MHz wall clock).  usage: phase_stats.py dim n pairs iters method"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth
dim, n, pairs, iters, method = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
lib = icp_amd._lib.load()
s, t, off, _ = synth.config5_pairs(np.arange(pairs), dim, n)
crit = icp_amd.make_criteria(min_iterations=iters, max_iterations=iters)
out = (C.c_ulonglong * 32)()
icp_amd.converge_batch(dim, method, s, off, t, off, crit, device=0)
lib.icpx_debug_stats(out, 1)
icp_amd.converge_batch(dim, method, s, off, t, off, crit, device=0)
lib.icpx_debug_stats(out, 0)
tot = pairs * max(iters, 1)
for nm, i in (("search", 8), ("selection", 9), ("sums", 10), ("kabsch+broadcast", 11)):
    print(f"{nm:18s} {out[i] / tot / 100:8.2f} us per pair-iteration")
print("raw counters:", [int(out[i]) for i in range(32)])
if out[16]:
    c = out[16]
    print(f"coop_box (workgroup 0... thread 0 of every workgroup): {c} calls, {out[22] / c:.0f} candidates per wave-slot; cycles per call: "
          f"rows {out[17] / c:.0f}, scan {out[18] / c:.0f}, list {out[19] / c:.0f}, eval {out[20] / c:.0f}, resolve {out[21] / c:.0f}; "
          f"{out[23]} calls walked large boxes ({out[24] / max(1, out[23]):.0f} cycles each)")
if out[15]:
    print(f"wave 0, per search call: {out[14] / out[15] / 100:.1f} us in pruned_search, {out[13] / out[15] / 100:.1f} us of it in nn_grid_slot; "
          f"per grid walk: {out[12] / max(1, out[0]):.1f} steps, {out[1] / max(1, out[0]):.1f} lane-evaluations")
