"""GPU parity test of the BENCHED shape: the exact shard bench.py times at N = 1 —
512 2D pairs (n = m = 1024, trimmed ICP2) + 512 3D pairs (n = m = 4096, vanilla ICP3), default knobs,
scheduler on, both batches in flight on two streams — through icpx_converge_batch, driven by the same
icp_amd.workload.Config5Shard object bench.py uses.

(a) 24 deterministically chosen pairs per kind (the 8 with the most iterations, the 8 with the fewest, 8
    spread) bit for bit against the CPU oracle in device order (brute-force (d2, index) NN, TREE64 sums);
(b) EVERY pair's state against the same call with the grid-pruned search switched off (ICPX_PRUNE=0,
    brute-force filter kernels) and with the cooperative walk switched off (ICPX_COOP=0);
(c) a second step reproduces the first bit for bit (persistent scheduler: which workgroup runs which
    iteration differs from run to run, results must not).
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

pytestmark = pytest.mark.gpu


def test_benched_shard_parity(ob, knob):
    import torch

    import icp_amd
    from icp_amd import workload
    import bench_check

    icp_amd._lib.check(icp_amd._lib.load().icpx_init(0))
    sh = workload.Config5Shard(torch, 0, 1, 512, 512, 0)
    assert (sh.p2, sh.p3) == (512, 512)
    sh.step()
    torch.cuda.synchronize()
    first = sh.out.cpu().numpy().copy()
    st2, st3 = sh.states()
    assert st2["it"].min() >= 1 and st3["it"].min() >= 1
    assert st2["it"].max() <= workload.MAX_ITER and st3["it"].max() <= workload.MAX_ITER
    # (a) oracle, device order, bit for bit
    assert bench_check.verify_sample(ob, sh, 24) == 48
    # (c) reproducible
    sh.step()
    torch.cuda.synchronize()
    assert np.array_equal(sh.out.cpu().numpy(), first, equal_nan=True)
    # (b) every pair against the other kernels
    for name in ("ICPX_COOP", "ICPX_PRUNE"):
        knob.set(name, "0")
        sh.step()
        torch.cuda.synchronize()
        knob.reset(name)
        other = sh.out.cpu().numpy()
        bad = np.nonzero((other != first).any(axis=1))[0]
        assert len(bad) == 0, (name, bad[:10])


def test_whole_config5_batch_once(ob):
    """All of config 5 (4096 2D + 4096 3D pairs, ~1 GB of clouds) in ONE step on one GPU — the strong-scaling
    figure bench.py reports as extras.c5_full_8192_ms.  16 sampled pairs bit for bit against the oracle; the first
    512 pairs of either kind are rank 0's shard at N = 8 and must equal that shard run on its own."""
    import torch

    import icp_amd
    from icp_amd import workload
    import bench_check

    icp_amd._lib.check(icp_amd._lib.load().icpx_init(0))
    full = workload.Config5Shard(torch, 0, 1, 4096, 4096, 0)
    assert (full.p2, full.p3) == (4096, 4096)
    full.step()
    torch.cuda.synchronize()
    f2, f3 = full.states()
    assert f2["it"].min() >= 1 and f3["it"].min() >= 1
    assert bench_check.verify_sample(ob, full, 8) == 16
    sh = workload.Config5Shard(torch, 0, 8, 4096, 4096, 0)
    sh.step()
    torch.cuda.synchronize()
    s2, s3 = sh.states()
    assert np.array_equal(f2[:512].view(np.uint8), s2.view(np.uint8))
    assert np.array_equal(f3[:512].view(np.uint8), s3.view(np.uint8))
