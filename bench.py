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

roofline: dominant kernel = the fused 3D pair kernel; algorithmic work = 8 flops per
(query, target) pair evaluated ((3*Dim - 1)), measured with events on the launch stream.
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
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--pairs-total", type=int, default=0,
                    help="strong scaling: this many pairs in all (half 2D, half 3D; config 5 names 8192) split over the "
                         "ranks; default 0 = weak scaling, 512 + 512 pairs per GPU")
    ap.add_argument("--verify-pairs", type=int, default=2, help="pairs per kind checked against the oracle after the timed region")
    ap.add_argument("--cpu-sample", type=int, default=512, help="pairs per dim timed on the CPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import icp_amd
    from icp_amd import shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

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

    def step(timed_kernel=False):
        sh.step(timed=timed_kernel)
        if world > 1:
            gathered = shard.gather_states(sh.out, total2 + total3, rank, world, counts=(total2, total3))
        if timed_kernel:
            a, b = sh.launch_ms()
            k3_ms.append(a)
            k2_ms.append(b)

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
        flops3 = float((it3 * (8.0 * N_3D * N_3D)).sum())
        achieved = flops3 / (k3 * 1e-3) / 1e12
        traffic = measured_traffic()
        valu = measured_traffic("valu_instructions_per_launch")
        line = {
            "metric": "nn_correspondence_queries_per_s",
            "value": total_q / (elapsed / args.steps),
            "unit": "queries/s",
            "n_gpus": world,
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
            "launch_ms": {"3d": k3, "2d": k2},
            "config": {
                "workload": "BASELINE config 5 shard per GPU: 512 2D pairs (n=m=1024, trimmed ICP2, "
                            "overlap 0.9) + 512 3D pairs (n=m=4096, vanilla ICP3), ICPDriver "
                            "max_iterations=50, transform tolerance (1e-4, 1e-4); results gathered with RCCL",
                "pairs_per_gpu": sh.p2 + sh.p3,
                "mean_iterations_2d": float(it2.mean()),
                "mean_iterations_3d": float(it3.mean()),
                "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
            },
            "roofline": {
                "bound": "valu_fp64",
                "kernel": "pair_kernel<3,vanilla,8,512,pruned> (fused 3D converge; persistent workgroups, one or two "
                          "pair-iterations per task; grid-pruned exact NN search, icpx_prune.h)",
                "note": "achieved = ALGORITHMIC f64 flops (SURVEY 8d: 8 per query-target pair of every "
                        "correspondence step, i.e. the brute-force count) / launch time.  Since r01_f the batch "
                        "kernel reaches the same bit-exact (d2, index) result through a per-pair uniform grid in "
                        "LDS and only evaluates the targets of the cells within each query's bound (a few per "
                        "cent of the pairs), so the figure exceeds the VALU peak: frac > 1 says the kernel beats "
                        "the brute-force roofline by skipping work, not that it issues f64 faster than the "
                        "hardware can.  The brute-force kernel it replaces (ICPX_PRUNE=0) reaches frac 0.72",
                "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS,
                "frac_of_unfused_peak": achieved / (FP64_PEAK_TFLOPS / 2),
                "launch_ms": k3,
                "flops_per_launch": flops3,
                "traffic": traffic,
                # executed work: wave-level VALU instructions of the launch (PMC SQ_INSTS_VALU, profiles/) x 4 issue
                # cycles over the cycles 1024 SIMDs have in the launch at 2.4 GHz — how busy the pruned kernel
                # keeps the vector units (it is bound by the LDS round trips of its per-lane cell walks)
                "valu_issue_frac": None if valu is None else valu * 4.0 / (1024 * 2.4e9 * k3 * 1e-3),
                # the other roofline of the same launch: HBM bytes (PMC, profiles/) over its duration
                "hbm": None if traffic is None else {"achieved": traffic / (k3 * 1e-3) / 1e9, "peak": 8000.0,
                                                     "unit": "GB/s", "frac": traffic / (k3 * 1e-3) / 1e9 / 8000.0},
            },
        }
        if not args.no_extras and world == 1:
            line["extras"] = extras(torch, icp_amd, synth, dev, local_rank, stream)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(s2, t2, off2, s3, t3, off3, args.cpu_sample)
            line["cpu_baseline_all_cores"] = cpu_baseline_threads(s2, t2, off2, s3, t3, off3, args.cpu_sample)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def measured_traffic(field="hbm_bytes_per_launch_raw"):
    """HBM bytes (or another summarized counter) per launch of the dominant kernel from the newest committed PMC passes
    (profiles/*_counters.json, tools/collect_profiles.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate runs of this same command).  PMC counters cannot be read from inside this process;
    None when absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), reverse=True) + \
            [os.path.join(ROOT, "profiles", "r01_b_traffic.json")]:
        try:
            with open(path) as f:
                k = json.load(f)["kernels"]
            for name, v in k.items():
                if "pair_kernel<3, 0" in name and field in v:
                    return v[field]
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
    r = drv3.converge(qt, q)
    t0 = time.perf_counter()
    for _ in range(10):
        r = drv3.converge(qt, q)
    ex["c3_queen_vanilla3d_converge_ms"] = (time.perf_counter() - t0) / 10 * 1e3
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
    bytes_per_query = 8 * 3 + 32 * best[2] + 16
    gbps = bytes_per_query * len(src) / (best[1] * 1e-3) / 1e9
    ex["c4_grid"] = {"build_ms": best[0], "query_ms": best[1], "candidates_per_query": best[2],
                     "cells": int(best[3]), "algorithmic_bytes_per_query": bytes_per_query,
                     "roofline": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s",
                                  "frac": gbps / 8000.0}}
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
    return ex


def cpu_baseline(s2, t2, off2, s3, t3, off3, sample):
    """The oracle in reference order on one host core, on the first `sample` pairs of each kind."""
    from oracle import binding as ob
    crit = ob.make_criteria(max_iterations=MAX_ITER, transform_tol=TOL)
    k2, k3 = min(sample, PAIRS_2D), min(sample, PAIRS_3D)
    queries = 0
    t0 = time.perf_counter()
    o2 = ob.OracleICP(2, "trimmed")  # brute-force NN + std::sort, as trimmed.cpp
    for p in range(k2):
        r = o2.converge(s2[off2[p]:off2[p + 1]], t2[off2[p]:off2[p + 1]], None, crit)
        queries += (r.iteration_count + 1) * N_2D
    o3 = ob.OracleICP(3, "vanilla")  # kd-tree NN, as vanilla_3d.cpp
    for p in range(k3):
        r = o3.converge(s3[off3[p]:off3[p + 1]], t3[off3[p]:off3[p + 1]], None, crit)
        queries += r.iteration_count * N_3D
    dt = time.perf_counter() - t0
    return {
        "value": queries / dt,
        "unit": "queries/s",
        "pairs_per_s": (k2 + k3) / dt,
        "cores": 1,
        "kind": "port",
        "sample": f"first {k2} 2D pairs + first {k3} 3D pairs of rank 0's shard, {dt:.1f} s of CPU time; "
                  "oracle in reference order (brute-force NN + std::sort for trimmed ICP2, kd-tree NN for "
                  "vanilla ICP3, sequential sums), g++ -O3 -DNDEBUG",
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

    def work(w):
        q = 0
        o2, o3 = ob.OracleICP(2, "trimmed"), ob.OracleICP(3, "vanilla")
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
        "kind": "port",
        "sample": f"the same {k2} + {k3} pairs spread over {cores} worker threads, {dt:.1f} s wall",
    }


if __name__ == "__main__":
    main()
