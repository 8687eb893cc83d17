"""BASELINE config 5 as bench.py runs it, in one place so that the GPU parity test of the benched
shape (tests/test_gpu_bench_shape.py) and bench.py drive exactly the same code.

A rank's shard: `pairs_2d` 2D pairs (n = m = 1024, trimmed ICP2, overlap 0.9) and `pairs_3d` 3D pairs
(n = m = 4096, vanilla ICP3), ICPDriver criteria max_iterations = 50, transform tolerance
(1e-4 rad, 1e-4).  Pair ids are global; which ids a rank owns comes from icp_amd.shard.shard_range.
The two batches are independent, so a step enqueues them on two streams (icpx_converge_batch is
stream-ordered for device-resident arrays) and joins them on the caller's stream.
"""
from __future__ import annotations

import numpy as np

from . import api, shard, synth

N_2D, N_3D = 1024, 4096
MAX_ITER, TOL = 50, (1e-4, 1e-4)
STATE_DT = np.dtype([("cost", "f8"), ("it", "u8"), ("T", "f8", 16)])


class Config5Shard:
    def __init__(self, torch, rank: int, world: int, pairs_2d_total: int, pairs_3d_total: int, device: int):
        self.torch = torch
        self.device = device
        self.dev = torch.device("cuda", device)
        self.lo2, self.hi2 = shard.shard_range(pairs_2d_total, rank, world)
        self.lo3, self.hi3 = shard.shard_range(pairs_3d_total, rank, world)
        self.ids2 = np.arange(self.lo2, self.hi2)
        self.ids3 = np.arange(self.lo3, self.hi3)
        self.s2, self.t2, self.off2, _ = synth.config5_pairs(self.ids2, 2, N_2D)
        self.s3, self.t3, self.off3, _ = synth.config5_pairs(self.ids3, 3, N_3D)
        self.d_s2, self.d_t2 = torch.from_numpy(self.s2).to(self.dev), torch.from_numpy(self.t2).to(self.dev)
        self.d_s3, self.d_t3 = torch.from_numpy(self.s3).to(self.dev), torch.from_numpy(self.t3).to(self.dev)
        self.p2, self.p3 = len(self.ids2), len(self.ids3)
        # icpx_state records (144 B = 18 doubles), 2D pairs first
        self.out = torch.zeros((self.p2 + self.p3, 18), dtype=torch.float64, device=self.dev)
        self.crit = api.make_criteria(max_iterations=MAX_ITER, transform_tol=TOL)
        self.st2 = torch.cuda.Stream(self.dev)
        self.st3 = torch.cuda.Stream(self.dev)
        self.ev = None          # the four events of the last timed step
        self.timed = []         # ... of every timed step since the last launch_ms_all()
        self.ev_begin = torch.cuda.Event()

    def step(self, timed: bool = False):
        """One pass of the hot path over the shard; results in self.out, ordered on the current stream."""
        torch = self.torch
        cur = torch.cuda.current_stream(self.dev)
        # The 2D launch goes first and on the caller's stream: its pairs are latency-bound (a chain of short iterations,
        # two workgroups per compute unit) and finish at very different times; the persistent 3D workgroups need a whole
        # compute unit each, start on the units the 2D pairs leave and take their tasks from the 3D launch's queue.
        # The 3D launch runs on a second stream that waits for what was on the caller's stream BEFORE this step (not
        # for the 2D launch), and the caller's stream joins it at the end: one cross-stream hand-over per step.
        if timed:  # events of this step's own (read after the timed region: no host synchronisation between steps)
            self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            self.timed.append(self.ev)
        self.ev_begin.record(cur)
        if self.p3:
            self.st3.wait_event(self.ev_begin)
        if self.p2:
            if timed:
                self.ev[2].record(cur)
            api.converge_batch(2, "trimmed", self.d_s2, self.off2, self.d_t2, self.off2, self.crit,
                               out_states=self.out[:self.p2], device=self.device, stream=cur.cuda_stream)
            if timed:
                self.ev[3].record(cur)
        if self.p3:
            if timed:
                self.ev[0].record(self.st3)
            api.converge_batch(3, "vanilla", self.d_s3, self.off3, self.d_t3, self.off3, self.crit,
                               out_states=self.out[self.p2:], device=self.device, stream=self.st3.cuda_stream)
            if timed:
                self.ev[1].record(self.st3)
            cur.wait_stream(self.st3)

    def launch_ms(self):
        """(3D launch, 2D launch) durations of the last timed step, HIP events on the launch streams."""
        self.torch.cuda.synchronize(self.dev)
        k3 = self.ev[0].elapsed_time(self.ev[1]) if self.p3 else 0.0
        k2 = self.ev[2].elapsed_time(self.ev[3]) if self.p2 else 0.0
        return k3, k2

    def launch_ms_all(self):
        """([3D launch ms], [2D launch ms]) of every timed step since the last call; synchronises once."""
        self.torch.cuda.synchronize(self.dev)
        k3 = [e[0].elapsed_time(e[1]) for e in self.timed] if self.p3 else []
        k2 = [e[2].elapsed_time(e[3]) for e in self.timed] if self.p2 else []
        self.timed = []
        return k3, k2

    def launch_alone_ms(self, dim: int, reps: int = 5) -> float:
        """Mean duration of the 3D (or 2D) batch launched alone on its stream (HIP events on that stream): in
        step() the two launches share the chip, so each one's events span most of the step."""
        torch = self.torch
        st = self.st3 if dim == 3 else self.st2
        if not (self.p3 if dim == 3 else self.p2):
            return 0.0
        torch.cuda.synchronize(self.dev)
        ms = []
        for rep in range(reps + 1):  # (the first one, untimed, creates the stream's workspace)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            if dim == 3:
                api.converge_batch(3, "vanilla", self.d_s3, self.off3, self.d_t3, self.off3, self.crit,
                                   out_states=self.out[self.p2:], device=self.device, stream=st.cuda_stream)
            else:
                api.converge_batch(2, "trimmed", self.d_s2, self.off2, self.d_t2, self.off2, self.crit,
                                   out_states=self.out[:self.p2], device=self.device, stream=st.cuda_stream)
            b.record(st)
            st.synchronize()
            if rep:
                ms.append(a.elapsed_time(b))
        return float(np.mean(ms))

    def states(self):
        raw = self.out.cpu().numpy().reshape(-1).view(STATE_DT)
        return raw[:self.p2], raw[self.p2:]

    def host_pair(self, dim: int, k: int):
        """(source, target) of the k-th local pair of the given dimension (host arrays)."""
        if dim == 2:
            return self.s2[self.off2[k]:self.off2[k + 1]], self.t2[self.off2[k]:self.off2[k + 1]]
        return self.s3[self.off3[k]:self.off3[k + 1]], self.t3[self.off3[k]:self.off3[k + 1]]


def transform_of(rec, dim: int) -> np.ndarray:
    d1 = dim + 1
    return np.array(rec["T"][: d1 * d1]).reshape(d1, d1, order="F")


def sample_pairs(iters: np.ndarray, count: int) -> list:
    """Deterministic sample of `count` local pair indices: a third with the most iterations, a third with
    the fewest, the rest evenly spaced."""
    n = len(iters)
    if n <= count:
        return list(range(n))
    order = np.argsort(iters, kind="stable")
    third = count // 3
    pick = list(order[::-1][:third]) + list(order[:third])
    for k in np.linspace(0, n - 1, count).astype(int):
        if len(pick) >= count:
            break
        if k not in pick:
            pick.append(int(k))
    k = 0
    while len(pick) < count:
        if k not in pick:
            pick.append(k)
        k += 1
    return sorted(int(x) for x in pick)
