"""Multi-GPU plumbing: independent scan pairs shard across ranks (one process per GPU); the only
exchange is one gather of the per-pair results (icpx_state records, 144 B each) — RCCL over xGMI on
GPUs, gloo in the CPU tests.  No collective touches the data path."""
from __future__ import annotations


def shard_range(n_pairs: int, rank: int, world: int):
    """Contiguous block of pairs owned by `rank`: pair p -> rank floor(p * world / n_pairs)."""
    lo = (n_pairs * rank + world - 1) // world
    hi = (n_pairs * (rank + 1) + world - 1) // world
    return lo, hi


def gather_states(local, n_pairs: int, rank: int, world: int, counts=None, scratch=None):
    """All ranks receive the (n_pairs, 18) f64 table of icpx_state records.  `local` holds this
    rank's rows; shards may be ragged, so rows are padded to the largest shard for the
    all_gather and trimmed afterwards.  counts: the batch is several independently sharded groups
    of pairs (bench.py: 2D pairs, then 3D pairs) whose sizes sum to n_pairs; a rank's rows are its
    block of every group, in group order, and so are the rows of the result (rank by rank).
    scratch: a dict the caller keeps between calls; the receive buffer lives in it, so a call that repeats every
    step allocates nothing (equal shards — the weak-scaling bench — then cost one collective and no other kernel;
    the returned table is that buffer: overwritten by the next call)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    groups = [n_pairs] if counts is None else list(counts)
    assert sum(groups) == n_pairs
    sizes = []
    for r in range(world):
        rows = sum(shard_range(g, r, world)[1] - shard_range(g, r, world)[0] for g in groups)
        sizes.append((0, rows))
    cap = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == cap for lo, hi in sizes) and local.shape[0] == cap:
        key = ("out", world * cap, local.shape[1], local.dtype, str(local.device))
        out = scratch.get(key) if scratch is not None else None
        if out is None:
            out = torch.empty((world * cap, local.shape[1]), dtype=local.dtype, device=local.device)
            if scratch is not None:
                scratch[key] = out
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    pad = torch.zeros((cap, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.zeros((world * cap, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    rows = [out[r * cap: r * cap + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(rows)
