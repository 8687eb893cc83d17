"""GPU test of icpx_converge_batch_noalloc (the entry point that allocates nothing): through the C ABI,
its output against icpx_converge_batch's on the same device-resident inputs, bit for bit, on two calls in
a row — the second with a smaller input, so stale contents of the shared scratch array would show — plus
the limits it must reject.  Collected last (see the file name): a failure here must not hide other tests."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_both(icp, torch, dim, method, P, n, first_id):
    from icp_amd import synth
    L = icp._lib
    lib = L.load()
    s, t, off, _ = synth.config5_pairs(range(first_id, first_id + P), dim, n)
    d_s, d_t = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    crit = icp.make_criteria(max_iterations=30, transform_tol=(1e-4, 1e-4))
    mcode = L.VANILLA if method == "vanilla" else L.TRIMMED
    outs = []
    for fn in (lib.icpx_converge_batch_noalloc, lib.icpx_converge_batch):
        out = torch.full((P, 18), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        L.check(fn(dim, mcode, None, P, d_s.data_ptr(), off.ctypes.data, d_t.data_ptr(), off.ctypes.data, None,
                   C.byref(crit), out.data_ptr()))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1], equal_nan=True), (dim, method, P, n)
    its = outs[0][:, 1].copy().view(np.uint64)
    assert its.min() >= 1 and its.max() <= 30
    return outs[0]


def test_noalloc_entry_point_matches_the_allocating_one():
    import torch

    import icp_amd as icp
    L = icp._lib
    lib = L.load()
    icp._lib.check(lib.icpx_init(0))
    # two calls in a row, the second smaller (3D: scheduler + grid-pruned search + running clouds in scratch)
    run_both(icp, torch, 3, "vanilla", 120, 1500, 0)
    run_both(icp, torch, 3, "vanilla", 33, 700, 500)
    run_both(icp, torch, 2, "trimmed", 256, 1024, 0)
    run_both(icp, torch, 2, "trimmed", 40, 300, 900)
    run_both(icp, torch, 3, "trimmed", 200, 600, 40)
    # limits: too many pairs, too many points, host arrays, a pair beyond the fused kernel
    crit = icp.make_criteria(max_iterations=5)
    x = torch.zeros((600000, 3), dtype=torch.float64, device="cuda")
    out = torch.zeros((300, 18), dtype=torch.float64, device="cuda")
    off = np.arange(258, dtype=np.int64) * 10
    assert lib.icpx_converge_batch_noalloc(3, 0, None, 257, x.data_ptr(), off.ctypes.data, x.data_ptr(), off.ctypes.data,
                                           None, C.byref(crit), out.data_ptr()) == L.ERR_UNSUPPORTED
    assert b"256" in lib.icpx_last_error()
    off = np.array([0, 3000, 600000], dtype=np.int64)
    assert lib.icpx_converge_batch_noalloc(3, 0, None, 2, x.data_ptr(), off.ctypes.data, x.data_ptr(), off.ctypes.data,
                                           None, C.byref(crit), out.data_ptr()) == L.ERR_UNSUPPORTED
    off = np.array([0, 3000, 9000], dtype=np.int64)  # second pair: 6000 source points > 4096
    assert lib.icpx_converge_batch_noalloc(3, 0, None, 2, x.data_ptr(), off.ctypes.data, x.data_ptr(), off.ctypes.data,
                                           None, C.byref(crit), out.data_ptr()) == L.ERR_UNSUPPORTED
    host = np.zeros((100, 3))
    off = np.array([0, 100], dtype=np.int64)
    assert lib.icpx_converge_batch_noalloc(3, 0, None, 1, host.ctypes.data, off.ctypes.data, x.data_ptr(), off.ctypes.data,
                                           None, C.byref(crit), out.data_ptr()) == L.ERR_INVALID
