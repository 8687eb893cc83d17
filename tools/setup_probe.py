"""Development aid: setup-only (0 iterations) time of a 2D trimmed batch for different source clouds.
usage: setup_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import icp_amd
from icp_amd import synth
dev = torch.device("cuda", 0)
P, n = 512, 1024
s, t, off, _ = synth.config5_pairs(np.arange(P), 2, n)
crit = icp_amd.make_criteria(min_iterations=0, max_iterations=0)
out = torch.zeros((P, 18), dtype=torch.float64, device=dev)
for name, src in (("config5", s), ("source = target", t.copy()), ("source = target * 0.9 (all inside)", t * 0.9),
                  ("source = target + 0.3", t + 0.3), ("source = target * 1.1 (rim outside)", t * 1.1)):
    ds, dt = torch.from_numpy(src).to(dev), torch.from_numpy(t).to(dev)
    for _ in range(2):
        icp_amd.converge_batch(2, "trimmed", ds, off, dt, off, crit, out_states=out, device=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        icp_amd.converge_batch(2, "trimmed", ds, off, dt, off, crit, out_states=out, device=0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    extra = ""
    lib = icp_amd._lib.load()
    if hasattr(lib, "icpx_debug_stats"):
        import ctypes as C
        st = (C.c_ulonglong * 32)()
        lib.icpx_debug_stats(st, 1)
        icp_amd.converge_batch(2, "trimmed", ds, off, dt, off, crit, out_states=out, device=0)
        lib.icpx_debug_stats(st, 0)
        extra = (f"  queries {st[29]}, candidates {st[26]}, lane slots {st[27]}, coop passes {st[28]}, walk calls (wave 0) {st[23]}, "
                 f"walk steps (wave 0) {st[12]}, walk evals (wave 0) {st[1]}")
    print(f"{name}: {ms:.3f} ms{extra}", flush=True)
