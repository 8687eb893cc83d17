"""Development aid (needs a -DICPX_STATS build): busy / alive time of the persistent workgroups for the
bench's 3D shard."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth
lib = icp_amd._lib.load()
s, t, off, _ = synth.config5_pairs(np.arange(512), 3, 4096)
crit = icp_amd.make_criteria(max_iterations=50, transform_tol=(1e-4, 1e-4))
out = (C.c_ulonglong * 32)()
icp_amd.converge_batch(3, "vanilla", s, off, t, off, crit, device=0)
lib.icpx_debug_stats(out, 1)
_, it, _ = icp_amd.converge_batch(3, "vanilla", s, off, t, off, crit, device=0)
lib.icpx_debug_stats(out, 0)
busy, tasks, alive = out[5], out[6], out[7]
print(f"tasks {tasks}  iterations {it.sum()}  busy ticks {busy}  alive ticks {alive}  busy/alive {busy / max(alive, 1):.3f}  "
      f"mean task {busy / max(tasks, 1) / 100:.1f} us (100 MHz clock)  alive per WG {alive / 256 / 100:.1f} us")
print("seeded scans", out[0], "flushes", out[1], "two-min scans", out[2], "near-tie lanes", out[3], "rescans", out[4])
