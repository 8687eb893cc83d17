#!/usr/bin/env python3
"""bench.py — throughput of the ICP hot path on MI355X.

Workload (BASELINE.json config 5, "batch of independent synthetic 2D/3D scan pairs", one shard per
GPU, weak scaling): every rank converges 1024 scan pairs per step —
  512 2D pairs, n = m = 1024, trimmed ICP2 (overlap 0.9)            [config 2's method]
  512 3D pairs, n = m = 4096, vanilla ICP3 (point-to-point, 3x3 SVD) [config 3's method]
with ICPDriver criteria max_iterations = 50, transform tolerance (1e-4 rad, 1e-4).  One "step" =
icpx_converge_batch over the rank's whole shard (inputs resident in HBM before the clock starts)
followed by the RCCL all_gather of the per-pair results when N > 1.

Metric: NN correspondence queries/s = source points answered by nearest-neighbour steps per second,
summed over all ranks; converged scan pairs/s is printed in the same JSON line.

roofline: dominant kernel = the fused 3D pair kernel, timed with events on its launch stream; work =
what the kernel executes (roofline_3d below), the brute-force count of SURVEY §8d kept as a separate figure.
cpu_baseline: the CPU oracle in reference order (kd-tree NN, std::sort, sequential sums), one core,
on a bounded sample of the same pairs (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PAIRS_2D, N_2D = 512, 1024
PAIRS_3D, N_3D = 512, 4096
MAX_ITER, TOL = 50, (1e-4, 1e-4)
FP64_PEAK_TFLOPS = 78.6  # MI355X vector FP64, FMA counted as 2 flops (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-executed-work", action="store_true",
                    help="skip the counting launch of the -DICPX_STATS build (profiling runs: it would appear under the "
                         "timed kernel's name)")
    ap.add_argument("--pairs-total", type=int, default=0,
                    help="strong scaling: this many pairs in all (half 2D, half 3D; config 5 names 8192) split over the "
                         "ranks; default 0 = weak scaling, 512 + 512 pairs per GPU")
    ap.add_argument("--verify-pairs", type=int, default=2, help="pairs per kind checked against the oracle after the timed region")
    ap.add_argument("--cpu-sample", type=int, default=512, help="pairs per dim timed on the CPU")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (gloo, no kernels): launch, shard ranges, result gather")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU) BEFORE anything touches a
    # GPU, relay rank 0's JSON line and leave with the launcher's exit code.  Under torch.distributed.run (the
    # driver's N > 1 launch) WORLD_SIZE is set and this process is one of the ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"error: --gpus {args.gpus} but WORLD_SIZE is {world_env}: start it as `python bench.py --gpus N` "
              "or under torch.distributed.run with --nproc-per-node N", file=sys.stderr)
        sys.exit(2)
    if args.dry_ranks:
        sys.exit(dry_ranks(args))

    import torch
    import torch.distributed as dist

    import icp_amd
    from icp_amd import shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
        rccl_ranks = dist.get_world_size()

    icp_amd._lib.check(icp_amd._lib.load().icpx_init(local_rank))

    # ---- this rank's shard of the batch: icp_amd.shard decides who owns which pair ids ----
    from icp_amd import workload
    if args.pairs_total:
        total2 = total3 = args.pairs_total // 2   # strong scaling: the whole config-5 batch over `world` ranks
        scaling = "strong"
    else:
        total2, total3 = PAIRS_2D * world, PAIRS_3D * world   # weak scaling: a fixed shard per GPU
        scaling = "weak"
    sh = workload.Config5Shard(torch, rank, world, total2, total3, local_rank)
    s2, t2, off2, s3, t3, off3 = sh.s2, sh.t2, sh.off2, sh.s3, sh.t3, sh.off3
    k3_ms, k2_ms = [], []
    gather_scratch = {}

    def step(timed_kernel=False):
        sh.step(timed=timed_kernel)
        if world > 1:
            gathered = shard.gather_states(sh.out, total2 + total3, rank, world, counts=(total2, total3), scratch=gather_scratch)
        # (the launches' HIP events are read after the timed region: nothing synchronises the host between steps)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed_kernel=True)
    barrier()
    elapsed = time.perf_counter() - t0
    k3_ms, k2_ms = sh.launch_ms_all()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- work accounting from the results (deterministic across steps) ----
    st2, st3 = sh.states()
    it2, it3 = st2["it"].astype(np.int64), st3["it"].astype(np.int64)
    # NN steps: trimmed ICP2 matches once in setup (trimmed.cpp:31-36) and once per iterate;
    # vanilla ICP3 once per iterate (vanilla_3d.cpp:85)
    q2 = int(((it2 + 1) * N_2D).sum())
    q3 = int((it3 * N_3D).sum())
    queries = torch.tensor([q2 + q3, sh.p2 + sh.p3], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(queries, op=dist.ReduceOp.SUM)
    total_q, total_pairs = float(queries[0].item()), float(queries[1].item())

    # ---- after the timed region: a sample of this shard's results against the CPU oracle (the checker) ----
    verified = 0
    if args.verify_pairs > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
        import bench_check
        from oracle import binding as ob
        verified = bench_check.verify_sample(ob, sh, args.verify_pairs)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        k3 = float(np.mean(k3_ms)) if k3_ms else float("nan")
        k2 = float(np.mean(k2_ms)) if k2_ms else float("nan")
        alone3, alone2 = sh.launch_alone_ms(3), sh.launch_alone_ms(2)
        roof = roofline_3d(torch, sh, it3, alone3, not args.no_executed_work)
        roof["launch_ms_in_step"] = k3
        line = {
            "metric": "nn_correspondence_queries_per_s",
            "value": total_q / (elapsed / args.steps),
            "unit": "queries/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,   # dist.get_world_size() after the NCCL (= RCCL) initialisation
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "converged_pairs_per_s": total_pairs / (elapsed / args.steps),
            "verified_pairs": verified,
            # events on each launch's own stream: inside the timed steps (the two launches share the chip, so each
            # spans most of the step) and launched alone after the timed region
            "launch_ms": {"3d": k3, "2d": k2, "3d_alone": alone3, "2d_alone": alone2},
            "config": {
                "workload": f"BASELINE config 5 shard per GPU: {sh.p2} 2D pairs (n=m=1024, trimmed ICP2, "
                            f"overlap 0.9) + {sh.p3} 3D pairs (n=m=4096, vanilla ICP3), ICPDriver "
                            "max_iterations=50, transform tolerance (1e-4, 1e-4); results gathered with RCCL"
                            + (f"; strong scaling: {total2} + {total3} pairs in all" if args.pairs_total else ""),
                "pairs_per_gpu": sh.p2 + sh.p3,
                "mean_iterations_2d": float(it2.mean()),
                "mean_iterations_3d": float(it3.mean()),
                "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": roof,
        }
        if not args.no_extras and world == 1:
            line["extras"] = extras(torch, icp_amd, synth, dev, local_rank, torch.cuda.current_stream(dev).cuda_stream)
            if not args.pairs_total:
                line["extras"].update(extras_config5(torch, icp_amd, workload, sh, local_rank))
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(s2, t2, off2, s3, t3, off3, args.cpu_sample)
            line["cpu_baseline_all_cores"] = cpu_baseline_threads(s2, t2, off2, s3, t3, off3, args.cpu_sample)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def launch_ranks(n: int) -> int:
    """One process per GPU through torch.distributed.run, as the driver starts N > 1 runs; returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run(cmd, env=env).returncode


def dry_ranks(args) -> int:
    """--dry-ranks: what the N ranks do around the kernels, on the CPU over gloo — shard ranges of the config-5
    batch, a fake result table per rank (row = pair id), the same gather bench.py's step() performs, one JSON line
    from rank 0.  No GPU, no HIP library."""
    import torch
    import torch.distributed as dist
    from icp_amd import shard
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    total2 = total3 = (args.pairs_total // 2) if args.pairs_total else 0
    if not args.pairs_total:
        total2, total3 = PAIRS_2D * world, PAIRS_3D * world
    rows = []
    for gi, g in enumerate((total2, total3)):
        lo, hi = shard.shard_range(g, rank, world)
        rows += [[gi, p] + [0.0] * 16 for p in range(lo, hi)]
    local = torch.tensor(rows, dtype=torch.float64).reshape(-1, 18)
    scratch = {}
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        table = shard.gather_states(local, total2 + total3, rank, world, counts=(total2, total3), scratch=scratch)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = table.shape[0] == total2 + total3
    got = sorted((int(a), int(b)) for a, b in table[:, :2].tolist())
    ok = ok and got == sorted([(0, p) for p in range(total2)] + [(1, p) for p in range(total3)])
    ranks = dist.get_world_size() if world > 1 else 1
    if rank == 0:
        print(json.dumps({"dry_ranks": True, "n_gpus": args.gpus, "ranks": ranks, "backend": "gloo",
                          "pairs_total": total2 + total3, "pairs_per_rank": local.shape[0], "gather_ok": bool(ok),
                          "gather_ms": dt / max(1, args.steps) * 1e3}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


UNFUSED_F64_PEAK_TFLOPS = FP64_PEAK_TFLOPS / 2   # one f64 flop per lane per VALU f64 issue slot (no FMA pairing)
LDS_PEAK_TBS = 256 * 256 * 2.4e9 / 1e12          # conflict-free ds_read_b64: 256 B/clk/CU x 256 CUs x 2.4 GHz


def roofline_3d(torch, sh, it3, k3_ms, count_work=True):
    """Roofline of the dominant kernel (the fused 3D pair kernel) from the work it EXECUTES.

    The kernel answers every query exactly but evaluates only the targets its grid search cannot rule out, so
    the brute-force count of SURVEY 8(d) (8 flops per query-target pair) is not what runs: it is reported as
    `speedup_vs_bruteforce_model`, never as frac.  Executed work is counted by the same kernel compiled with
    -DICPX_STATS (icp_amd/lib_stats, counters in icpx_prune.h / icpx_pair_pruned.h), run here once on this
    rank's 3D batch after the timed region, its results compared bit for bit with the timed library's:
      executed_flops = (3*Dim - 1) x evaluated candidates   vs the unfused f64 VALU ceiling (39.3 TFLOP/s)
      lds_bytes      = staged reads and writes of the search  vs the conflict-free ds_read_b64 rate
      lane_fraction  = evaluated candidates / lane slots issued for them
    HBM bytes and the VALU issue share come from rocprofv3 --pmc passes (they cannot be read from inside this
    process): `from_profile` names the committed file."""
    import ctypes as C
    from icp_amd import _lib, api
    sec = k3_ms * 1e-3   # the 3D batch launched alone on its stream (pre-pass + pair kernel), HIP events
    brute = float((it3 * (8.0 * N_3D * N_3D)).sum())
    roof = {
        "bound": "valu_fp64",
        "kernel": "pair_kernel<3,vanilla,8,512,pruned> (fused 3D converge: grid-pruned exact NN search with seed "
                  "certificates, TREE64 sums, 3x3 Kabsch, termination; persistent workgroups)",
        "unit": "TFLOP/s",
        "peak": UNFUSED_F64_PEAK_TFLOPS,
        "launch_ms": k3_ms,
        "achieved": None, "frac": None, "traffic": None,
        "speedup_vs_bruteforce_model": brute / sec / 1e12 / FP64_PEAK_TFLOPS,
        "bruteforce_model": "8 flops x n x m x iterations of every pair (SURVEY 8d) / launch time / 78.6 TFLOP/s "
                            "(FMA peak): how many brute-force-roofline kernels this launch is worth; not a roofline",
    }
    Ls = _lib.load_variant("lib_stats") if (sh.p3 and count_work) else None
    if Ls is not None and hasattr(Ls, "icpx_debug_stats"):
        _lib.check(Ls.icpx_init(sh.device))
        out = torch.zeros((sh.p3, 18), dtype=torch.float64, device=sh.dev)
        buf = (C.c_ulonglong * 32)()
        Ls.icpx_debug_stats(buf, 1)
        api.converge_batch(3, "vanilla", sh.d_s3, sh.off3, sh.d_t3, sh.off3, sh.crit, out_states=out,
                           device=sh.device, stream=torch.cuda.current_stream(sh.dev).cuda_stream, lib=Ls)
        torch.cuda.synchronize(sh.dev)
        Ls.icpx_debug_stats(buf, 0)
        same = bool(torch.equal(out.view(torch.int64), sh.out[sh.p2:].view(torch.int64)))
        evals, issued, passes = int(buf[26]), int(buf[27]), int(buf[28])
        queries, certified, seeded = int(buf[29]), int(buf[30]), int(buf[31])
        flops = 8.0 * (evals + seeded)               # every seed is one more distance evaluation
        # LDS bytes of the search: per evaluated candidate the target (24 B) and the staged query (24 B), the list
        # entry written and read (8 B), the 64-bit minimum (atomic + re-read, 16 B), the slot report (8 B);
        # per cooperative pass and lane the 18 cell-table reads (2 B), the staged query and accumulators (40 B);
        # per seed its target (24 B) and the certificate radius comes from HBM
        lds = evals * (24 + 24 + 8 + 16 + 8) + passes * 64 * (18 * 2 + 40) + seeded * 24
        roof.update({
            "achieved": flops / sec / 1e12,
            "frac": flops / sec / 1e12 / UNFUSED_F64_PEAK_TFLOPS,
            "executed": {
                "source": "icp_amd/lib_stats (-DICPX_STATS build of the same kernel), one launch on this shard",
                "results_identical_to_timed_library": same,
                "queries": queries, "queries_answered_by_certified_seed": certified,
                "evaluated_candidates": evals, "seed_evaluations": seeded,
                "candidates_per_searched_query": evals / max(1, queries - certified),
                "executed_flops": flops,
                "lane_slots_issued": issued, "lane_fraction": evals / max(1, issued),
                "lds_bytes": lds,
                "lds": {"achieved": lds / sec / 1e12, "peak": LDS_PEAK_TBS, "unit": "TB/s",
                        "frac": lds / sec / 1e12 / LDS_PEAK_TBS},
            },
        })
    prof = profile_counters()
    if prof is not None:
        name, k = prof
        raw, dbl = k.get("hbm_bytes_per_launch_raw"), k.get("hbm_bytes_per_launch_fetch_doubled")
        valu = k.get("SQ_INSTS_VALU_per_launch")
        roof["from_profile"] = name
        roof["traffic"] = dbl
        roof["traffic_raw"] = raw
        if dbl is not None:
            roof["hbm"] = {"achieved": dbl / sec / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": dbl / sec / 1e9 / 8000.0}
        if valu is not None:
            # wave-level VALU instructions x 4 issue cycles over the cycles 1024 SIMDs have in the launch at 2.4 GHz
            roof["valu_issue_frac"] = valu * 4.0 / (1024 * 2.4e9 * sec)
    return roof


def profile_counters():
    """(file name, per-launch counters of the 3D pair kernel) from the newest committed PMC summary
    (profiles/rNN_*_counters.json; tools/collect_profiles.sh: rocprofv3 --pmc passes over this same command)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), reverse=True):
        try:
            with open(path) as f:
                k = json.load(f)["kernels"]
            # (the hot 3D kernel, not the kd rule's take-over kernel launched behind it: the one with the traffic)
            hot = [v for name, v in k.items() if "pair_kernel<3, 0" in name]
            if hot:
                return "profiles/" + os.path.basename(path), max(hot, key=lambda v: v.get("hbm_bytes_per_launch_raw") or 0)
        except (OSError, KeyError, ValueError):
            continue
    return None


def extras(torch, icp_amd, synth, dev, local_rank, stream):
    """Side measurements of the other single-GPU configs (not the headline value)."""
    from icp_amd import io as iio
    ex = {}
    g = os.path.join(ROOT, "tests", "golden", "ex_data")
    # config 2: ex_data scan1, trimmed ICP2, single pair (latency of one converge call, host data)
    a = iio.parse_lidar_scan(os.path.join(g, "scan1", "first.csv"))
    b = iio.parse_lidar_scan(os.path.join(g, "scan1", "second.csv"))
    drv = icp_amd.ICPDriver(icp_amd.ICP2.from_method("trimmed", device=local_rank))
    drv.set_max_iterations(50)
    drv.set_transform_tolerance(0.1 * np.pi / 180, 0.01)
    # (forty untimed calls first: after the idle moment between two measurements the GPU needs some ten milliseconds of
    # work to reach its clocks — the first calls of a series are 20 % slower than the steady state, tools/queen_variance.py)
    for _ in range(40):
        r = drv.converge(a, b)
    t0 = time.perf_counter()
    for _ in range(20):
        r = drv.converge(a, b)
    ex["c2_scan1_trimmed_converge_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    ex["c2_scan1_iterations"] = r.iteration_count
    # M1 / M2 of SURVEY 8(d) for this config: trimmed ICP2 matches once in setup and once per iterate
    ex["c2_scan1_queries_per_s"] = (r.iteration_count + 1) * len(a) / (ex["c2_scan1_trimmed_converge_ms"] * 1e-3)
    ex["c2_scan1_pairs_per_s"] = 1e3 / ex["c2_scan1_trimmed_converge_ms"]
    # config 3: queen.ply, vanilla ICP3
    q = iio.read_ply_vertices(os.path.join(g, "ply", "queen.ply"))
    qt = iio.read_ply_vertices(os.path.join(g, "ply", "queen_transformed.ply"))
    drv3 = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", device=local_rank))
    drv3.set_max_iterations(100)
    drv3.set_transform_tolerance(1e-4, 1e-4)
    for _ in range(40):
        r = drv3.converge(qt, q)
    t0 = time.perf_counter()
    for _ in range(20):
        r = drv3.converge(qt, q)
    ex["c3_queen_vanilla3d_converge_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    ex["c3_queen_iterations"] = r.iteration_count
    ex["c3_queen_queries_per_s"] = r.iteration_count * len(qt) / (ex["c3_queen_vanilla3d_converge_ms"] * 1e-3)
    ex["c3_queen_pairs_per_s"] = 1e3 / ex["c3_queen_vanilla3d_converge_ms"]
    # config 4: 1M x 1M grid NN, one correspondence step, inputs and outputs resident in HBM
    src, tgt, _, _ = synth.config4(1_000_000)
    d_q, d_t = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
    idx = torch.zeros(len(src), dtype=torch.int64, device=dev)
    d2 = torch.zeros(len(src), dtype=torch.float64, device=dev)
    icp_amd.nearest_neighbors(d_q, d_t, icp_amd.NN_GRID, device=local_rank, stream=stream, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        icp_amd.nearest_neighbors(d_q, d_t, icp_amd.NN_GRID, device=local_rank, stream=stream, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    ex["c4_grid_nn_1m_build_plus_query_ms"] = dt * 1e3
    ex["c4_grid_nn_1m_queries_per_s"] = len(src) / dt
    # SURVEY 8(d) roofline of the grid query (HBM/L2-bound): algorithmic bytes per query = 8*D (query)
    # + 32 * c_bar (32-byte target records examined, counted by the kernel) + 16 (int64 index + f64 d2)
    import ctypes as C
    prof = (C.c_double * 4)()
    lib = icp_amd._lib.load()
    best = None
    for _ in range(3):
        icp_amd._lib.check(lib.icpx_nn_grid_profile(3, d_q.data_ptr(), len(src), d_t.data_ptr(), len(tgt),
                                                    local_rank, stream, prof))
        if best is None or prof[1] < best[1]:
            best = list(prof)
    # two accountings of the algorithmic bytes per query: SURVEY 8(d)'s formula (24-B coordinates per examined target,
    # 24-B query, 12 B of result) and what the kernel's records really are (32-B padded records, 16 B of result); and
    # the compulsory floor (every query and every target read once, every result written once)
    c_bar = best[2]
    b24 = 8 * 3 * c_bar + 8 * 3 + 12
    b32 = 32 * c_bar + 8 * 3 + 16
    floor = 8 * 3 + 8 * 3 * len(tgt) / len(src) + 12
    sec = best[1] * 1e-3
    ex["c4_grid"] = {"build_ms": best[0], "query_ms": best[1], "candidates_per_query": c_bar, "cells": int(best[3]),
                     "algorithmic_bytes_per_query": b24, "record_bytes_per_query": b32, "compulsory_bytes_per_query": floor,
                     "roofline": {"bound": "hbm", "achieved": b24 * len(src) / sec / 1e9, "peak": 8000.0, "unit": "GB/s",
                                  "frac": b24 * len(src) / sec / 1e9 / 8000.0,
                                  "frac_32B_records": b32 * len(src) / sec / 1e9 / 8000.0,
                                  "frac_of_compulsory_floor": floor * len(src) / sec / 1e9 / 8000.0,
                                  "note": "examined records mostly come from L2 (neighbouring queries share cells): the "
                                          "algorithmic figure counts them all; measured HBM bytes are in profiles/ "
                                          "(rNN_counters.json, grid_nn_kernel)"}}
    # f3: the OPT-IN f32 correspondence mode on the same clouds (16-byte f32 records, candidates evaluated in f32;
    # not reference arithmetic, not part of `value`): one correspondence step, build + query, and how often it returns
    # the f64 search's target
    idx32 = torch.zeros_like(idx)
    icp_amd.nearest_neighbors(d_q, d_t, icp_amd.NN_GRID_F32, device=local_rank, stream=stream, out_idx=idx32, out_d2=d2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        icp_amd.nearest_neighbors(d_q, d_t, icp_amd.NN_GRID_F32, device=local_rank, stream=stream, out_idx=idx32, out_d2=d2)
    torch.cuda.synchronize()
    dt32 = (time.perf_counter() - t0) / 5
    ex["c4_f32_grid_nn_1m_build_plus_query_ms"] = dt32 * 1e3
    ex["c4_f32_grid_nn_1m_queries_per_s"] = len(src) / dt32
    ex["c4_f32_same_target_as_f64"] = float((idx32 == idx).double().mean().item())
    del idx32
    # config 4, clustered variant (points on 64 planes: most cells empty) — one correspondence step, build + query
    srcc, tgtc, _, _ = synth.config4(1_000_000, clustered=True)
    d_qc, d_tc = torch.from_numpy(srcc).to(dev), torch.from_numpy(tgtc).to(dev)
    icp_amd.nearest_neighbors(d_qc, d_tc, icp_amd.NN_GRID, device=local_rank, stream=stream, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        icp_amd.nearest_neighbors(d_qc, d_tc, icp_amd.NN_GRID, device=local_rank, stream=stream, out_idx=idx, out_d2=d2)
    torch.cuda.synchronize()
    dtc = (time.perf_counter() - t0) / 5
    ex["c4_clustered_grid_nn_1m_build_plus_query_ms"] = dtc * 1e3
    ex["c4_clustered_grid_nn_1m_queries_per_s"] = len(srcc) / dtc
    # f3: the linear BVH (icpx_bvh.h) beside the grid on the same clouds (same results; the grid wins on both —
    # the BVH is for clouds that collapse into a few grid cells: tools/degenerate_probe.py, 1.5 ms against 29 ms)
    for tag, qq, tt in (("c4", d_q, d_t), ("c4_clustered", d_qc, d_tc)):
        icp_amd.nearest_neighbors(qq, tt, icp_amd.NN_BVH, device=local_rank, stream=stream, out_idx=idx, out_d2=d2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            icp_amd.nearest_neighbors(qq, tt, icp_amd.NN_BVH, device=local_rank, stream=stream, out_idx=idx, out_d2=d2)
        torch.cuda.synchronize()
        ex[tag + "_bvh_nn_1m_build_plus_query_ms"] = (time.perf_counter() - t0) / 3 * 1e3
    del d_qc, d_tc
    # config 4, M2: full converge of the 1 M x 1 M pair (vanilla ICP3, max 50 iterations, tol 1e-4)
    drv4 = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", device=local_rank))
    drv4.set_max_iterations(MAX_ITER)
    drv4.set_transform_tolerance(*TOL)
    r4 = drv4.converge(d_q, d_t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r4 = drv4.converge(d_q, d_t)
    torch.cuda.synchronize()
    dt4 = time.perf_counter() - t0
    ex["c4_converge_ms"] = dt4 * 1e3
    ex["c4_converge_iterations"] = r4.iteration_count
    ex["c4_converge_queries_per_s"] = r4.iteration_count * len(src) / dt4
    # said plainly: on this synthetic pair the transform tolerance is not met within max_iterations, so this is the
    # time of 50 iterations, not of a converged registration
    ex["c4_converge_stopped_by"] = ("max_iterations (%d): not converged" % MAX_ITER) if r4.iteration_count >= MAX_ITER \
        else "transform tolerance"
    # the same converge with the opt-in f32 correspondence mode (Config nn = "grid_f32"; sums, Kabsch, termination f64)
    cfg32 = icp_amd.Config()
    cfg32.set("nn", "grid_f32")
    drv32 = icp_amd.ICPDriver(icp_amd.ICP3.from_method("vanilla", cfg32, device=local_rank))
    drv32.set_max_iterations(MAX_ITER)
    drv32.set_transform_tolerance(*TOL)
    r32 = drv32.converge(d_q, d_t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r32 = drv32.converge(d_q, d_t)
    torch.cuda.synchronize()
    ex["c4_f32_converge_ms"] = (time.perf_counter() - t0) * 1e3
    ex["c4_f32_converge_iterations"] = r32.iteration_count
    ex["c4_f32_converge_max_abs_transform_diff_vs_f64"] = float(np.abs(np.asarray(r32.transform) - np.asarray(r4.transform)).max())
    del drv32
    # roofline of one iteration of this path (HBM): what an iteration must move at least — the running cloud read and
    # written (48 B per point), the matched target's coordinates and index written for the sums (32 B per point), the
    # target cloud read once (24 B per target); the examined grid records (L2 mostly) are on top and not counted
    it_bytes = len(src) * (48 + 32) + len(tgt) * 24
    it_s = dt4 / max(1, r4.iteration_count)
    ex["c4_converge"] = {"ms_per_iteration": it_s * 1e3, "algorithmic_bytes_per_iteration": it_bytes,
                         "roofline": {"bound": "hbm", "achieved": it_bytes / it_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                                      "frac": it_bytes / it_s / 1e9 / 8000.0,
                                      "from_profile": newest_profile("_c4_converge_kernel_stats.csv"),
                                      "note": "per iteration: lg_nn_sd_kernel (seeded grid walk in the sorted domain through f32 filter "
                                              "records with exact f64 refinement, a chain of dependent L2 round trips) ~2/3 of the "
                                              "time, the two sum kernels the rest"}}
    # configs 2 and 3 (one pair each): the correspondence work of the path is brute force in the SURVEY 8(d) model,
    # 8 flops per (query, target) and correspondence step, against the f64 FMA peak — a latency figure, not a
    # throughput one: one scan-sized pair occupies one compute unit of 256
    f2 = 8.0 * len(a) * len(b) * (ex["c2_scan1_iterations"] + 1)
    ex["c2_scan1_roofline"] = {"bound": "valu_fp64", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS,
                               "achieved": f2 / (ex["c2_scan1_trimmed_converge_ms"] * 1e-3) / 1e12,
                               "frac": f2 / (ex["c2_scan1_trimmed_converge_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                               "model": "8 flops x n x m x (iterations + 1) correspondence steps"}
    f3 = 8.0 * len(qt) * len(q) * ex["c3_queen_iterations"]
    ex["c3_queen_roofline"] = {"bound": "valu_fp64", "unit": "TFLOP/s", "peak": FP64_PEAK_TFLOPS,
                               "achieved": f3 / (ex["c3_queen_vanilla3d_converge_ms"] * 1e-3) / 1e12,
                               "frac": f3 / (ex["c3_queen_vanilla3d_converge_ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                               "from_profile": newest_profile("_queen_kernel_stats.csv"),
                               "model": "8 flops x n x m x iterations (tiled brute force with the f32 candidate filter)"}
    return ex


def newest_profile(suffix):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)))
    return ("profiles/" + os.path.basename(files[-1])) if files else None


def extras_config5(torch, icp_amd, workload, sh, local_rank):
    """Config 5 beyond the headline shard, on this one GPU: the WHOLE 8192-pair batch in one step (the strong-scaling
    figure at N = 1), and the two method / dimension combinations the headline step leaves out (config 5 names
    "vanilla and trimmed" for both dimensions): 512 2D vanilla pairs and 512 3D trimmed pairs."""
    ex = {}
    from icp_amd import api
    crit = sh.crit
    dev = sh.dev

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps * 1e3

    cur = torch.cuda.current_stream(dev).cuda_stream
    out = torch.zeros((max(sh.p2, sh.p3), 18), dtype=torch.float64, device=dev)
    if sh.p2:
        ms = timed(lambda: api.converge_batch(2, "vanilla", sh.d_s2, sh.off2, sh.d_t2, sh.off2, crit,
                                              out_states=out[:sh.p2], device=local_rank, stream=cur))
        ex["c5_2d_vanilla_batch_ms"] = ms
        ex["c5_2d_vanilla_pairs_per_s"] = sh.p2 / (ms * 1e-3)
    if sh.p3:
        cfg = icp_amd.Config()
        ms = timed(lambda: api.converge_batch(3, "trimmed", sh.d_s3, sh.off3, sh.d_t3, sh.off3, crit, cfg,
                                              out_states=out[:sh.p3], device=local_rank, stream=cur))
        ex["c5_3d_trimmed_batch_ms"] = ms
        ex["c5_3d_trimmed_pairs_per_s"] = sh.p3 / (ms * 1e-3)
        it = out[:sh.p3].cpu().numpy().reshape(-1).view(workload.STATE_DT)["it"]
        ex["c5_3d_trimmed_mean_iterations"] = float(it.mean())
    del out
    # the whole config-5 batch (4096 + 4096 pairs, ~1 GB of clouds) on this GPU: strong scaling at N = 1
    full = workload.Config5Shard(torch, 0, 1, 4096, 4096, local_rank)
    ms = timed(lambda: full.step(), reps=3)
    st2, st3 = full.states()
    q = int(((st2["it"].astype(np.int64) + 1) * N_2D).sum()) + int((st3["it"].astype(np.int64) * N_3D).sum())
    ex["c5_full_8192_ms"] = ms
    ex["c5_full_8192_pairs_per_s"] = 8192 / (ms * 1e-3)
    ex["c5_full_8192_queries_per_s"] = q / (ms * 1e-3)
    del full
    return ex


def cpu_baseline(s2, t2, off2, s3, t3, off3, sample):
    """The oracle in reference order on one host core, on the first `sample` pairs of each kind."""
    from oracle import binding as ob
    crit = ob.make_criteria(max_iterations=MAX_ITER, transform_tol=TOL)
    k2, k3 = min(sample, PAIRS_2D), min(sample, PAIRS_3D)
    queries = 0
    # the kd-tree legs run through the reference's OWN include/algo/kdtree.h when oracle/_ref/liboracle_ref.so was
    # built from the reference checkout (oracle/Makefile); the restated tree otherwise
    ref_tree = ob.lib_ref_tree() is not None
    t0 = time.perf_counter()
    o2 = ob.OracleICP(2, "trimmed")  # brute-force NN + std::sort, as trimmed.cpp
    for p in range(k2):
        r = o2.converge(s2[off2[p]:off2[p + 1]], t2[off2[p]:off2[p + 1]], None, crit)
        queries += (r.iteration_count + 1) * N_2D
    o3 = ob.OracleICP(3, "vanilla", reference_tree=ref_tree)  # kd-tree NN, as vanilla_3d.cpp:30-45,75-80
    for p in range(k3):
        r = o3.converge(s3[off3[p]:off3[p + 1]], t3[off3[p]:off3[p + 1]], None, crit)
        queries += r.iteration_count * N_3D
    dt = time.perf_counter() - t0
    return {
        "value": queries / dt,
        "unit": "queries/s",
        "pairs_per_s": (k2 + k3) / dt,
        "cores": 1,
        "kind": "reference" if ref_tree else "port",
        "sample": f"first {k2} 2D pairs + first {k3} 3D pairs of rank 0's shard, {dt:.1f} s of CPU time; "
                  + ("kd-tree build and search of vanilla ICP3 = the reference's own include/algo/kdtree.h compiled "
                     "from its sources (oracle/_ref); " if ref_tree else "kd-tree restated (oracle/_ref not built); ")
                  + "everything that needs Eigen in the reference (brute-force NN + std::sort of trimmed ICP2, sums, "
                    "Kabsch SVD, driver) is the oracle's restatement in reference order — Eigen is absent here, the "
                    "reference's method classes cannot be built; g++ -O3 -DNDEBUG",
    }


def cpu_baseline_threads(s2, t2, off2, s3, t3, off3, sample):
    """the same oracle run over the same pairs, pairs spread over the host cores this
    process may use (one oracle instance per worker thread; ctypes releases the GIL).  The reference
    itself is single-threaded: this is the embarrassingly parallel upper bound a user could build."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    crit = ob.make_criteria(max_iterations=MAX_ITER, transform_tol=TOL)
    k2, k3 = min(sample, PAIRS_2D), min(sample, PAIRS_3D)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)), 64))

    ref_tree = ob.lib_ref_tree() is not None

    def work(w):
        q = 0
        o2, o3 = ob.OracleICP(2, "trimmed"), ob.OracleICP(3, "vanilla", reference_tree=ref_tree)
        for p in range(w, k3, cores):  # the expensive pairs first
            r = o3.converge(s3[off3[p]:off3[p + 1]], t3[off3[p]:off3[p + 1]], None, crit)
            q += r.iteration_count * N_3D
        for p in range(w, k2, cores):
            r = o2.converge(s2[off2[p]:off2[p + 1]], t2[off2[p]:off2[p + 1]], None, crit)
            q += (r.iteration_count + 1) * N_2D
        return q

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        queries = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    return {
        "value": queries / dt,
        "unit": "queries/s",
        "pairs_per_s": (k2 + k3) / dt,
        "cores": cores,
        "kind": "reference" if ref_tree else "port",
        "sample": f"the same {k2} + {k3} pairs spread over {cores} worker threads, {dt:.1f} s wall",
    }


if __name__ == "__main__":
    main()
