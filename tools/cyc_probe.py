"""Development aid (needs a -DICPX_STATS -DICPX_CYCONLY build: only thread 0's cycle stamps, no counters in the hot
loops): cycles of the cooperative search by part, for the iterations first..last of a fixed-iteration 3D batch.
usage: ICPX_LIB=icp_amd/lib_cyc/libicpx.so cyc_probe.py [pairs [iters ...]]      (ICPX_CYC_DIM=2: the 2D trimmed batch)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import icp_amd
from icp_amd import synth
lib = icp_amd._lib.load()
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters_list = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 5, 8]
dim = int(os.environ.get("ICPX_CYC_DIM", "3"))
npts, method = (4096, "vanilla") if dim == 3 else (1024, "trimmed")
s, t, off, _ = synth.config5_pairs(np.arange(pairs), dim, npts)
prev = None
for iters in iters_list:
    crit = icp_amd.make_criteria(min_iterations=iters, max_iterations=iters)
    out = (C.c_ulonglong * 32)()
    icp_amd.converge_batch(dim, method, s, off, t, off, crit, device=0)
    lib.icpx_debug_stats(out, 1)
    t0 = time.perf_counter()
    icp_amd.converge_batch(dim, method, s, off, t, off, crit, device=0)
    ms = (time.perf_counter() - t0) * 1e3
    lib.icpx_debug_stats(out, 0)
    v = np.array([int(out[i]) for i in range(32)], dtype=np.float64)
    d = v - prev if prev is not None else v
    prev = v
    calls = max(d[4], 1)
    print(f"iterations ..{iters}: {ms:.3f} ms host; thread 0 of {pairs} workgroups: {d[4]:.0f} slot searches, {d[5] / calls:.2f} trips, "
          f"{d[6] / calls:.2f} passes and {d[7] / calls:.0f} candidates per slot search; cycles per slot search: total {d[3] / calls:.0f} = "
          f"rows {d[17] / calls:.0f} + list {d[19] / calls:.0f} + eval {d[20] / calls:.0f} + resolve {d[21] / calls:.0f} + run scans {d[24] / calls:.0f} "
          f"+ other {(d[3] - d[17] - d[19] - d[20] - d[21] - d[24]) / calls:.0f}", flush=True)
