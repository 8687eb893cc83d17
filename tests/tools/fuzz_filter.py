"""Test infrastructure (lives under tests/ because it calls the oracle): randomized differential test of the fused kernels (f32 filter + exact refinement,
radix select, scheduler) against the oracle in device order.  usage: python tests/tools/fuzz_filter.py [seed] [rounds]   (tests/test_gpu_fuzz.py runs a short one)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import icp_amd
from oracle import binding as ob

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
bad = 0
t_start = time.time()
for rnd in range(rounds):
    dim = int(rng.integers(2, 4))
    method = ["vanilla", "trimmed"][int(rng.integers(0, 2))]
    pairs = int(rng.integers(1, 40))
    nmax = int(rng.choice([40, 300, 1100, 2100]))
    scale = 10.0 ** rng.uniform(-6, 6)
    offset = scale * 10.0 ** rng.uniform(-2, 5) * rng.choice([0.0, 1.0])
    kind = rng.choice(["uniform", "lattice", "dup", "line"])
    S, T, so, to = [], [], [0], [0]
    for p in range(pairs):
        n, m = int(rng.integers(1, nmax + 1)), int(rng.integers(1, nmax + 1))
        if kind == "lattice":
            t = rng.integers(0, 12, (m, dim)).astype(float)
        elif kind == "line":
            t = np.zeros((m, dim)); t[:, 0] = rng.uniform(-5, 5, m)
        else:
            t = rng.uniform(-5, 5, (m, dim))
        if kind == "dup":
            t[m // 2:] = t[: m - m // 2]
        s = t[rng.integers(0, m, n)] + rng.normal(0, 0.1, (n, dim)) * rng.choice([0.0, 1.0])
        S.append(s * scale + offset); T.append(t * scale + offset)
        so.append(so[-1] + n); to.append(to[-1] + m)
    S, T = np.concatenate(S), np.concatenate(T)
    so, to = np.array(so, np.int64), np.array(to, np.int64)
    ov = float(rng.choice([0.9, 0.5, 1.0]))
    md = float(0.3 * scale)
    cfg = icp_amd.Config(); cfg.set("overlap_rate", ov); cfg.set("max_distance", md)
    iters = int(rng.integers(1, 6))
    crit = icp_amd.make_criteria(max_iterations=iters)
    sched = str(int(rng.integers(0, 2)))
    coop = str(int(rng.integers(0, 2)))
    icp_amd._lib.check(icp_amd._lib.load().icpx_debug_knob(b"ICPX_SCHED", sched.encode()))
    icp_amd._lib.check(icp_amd._lib.load().icpx_debug_knob(b"ICPX_COOP", coop.encode()))
    cost, it, tr = icp_amd.converge_batch(dim, method, S, so, T, to, crit, config=cfg, device=0)
    for p in range(pairs):
        o = ob.OracleICP(dim, method, overlap_rate=ov, max_distance=md, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64)
        r = o.converge(S[so[p]:so[p + 1]], T[to[p]:to[p + 1]], None, ob.make_criteria(max_iterations=iters))
        same = np.array_equal(tr[p], r.transform, equal_nan=True) and it[p] == r.iteration_count and \
            (cost[p] == r.cost or (np.isnan(cost[p]) and np.isnan(r.cost)))
        if not same:
            bad += 1
            print("MISMATCH", dict(round=rnd, pair=p, dim=dim, method=method, kind=str(kind), scale=scale, offset=offset,
                                   n=int(so[p + 1] - so[p]), m=int(to[p + 1] - to[p]), ov=ov, iters=iters), flush=True)
    print(f"round {rnd} t={time.time() - t_start:.1f}s n_total={so[-1]}: dim={dim} {method} pairs={pairs} nmax={nmax} kind={kind} scale={scale:.2e} offset={offset:.2e} "
          f"sched={sched} coop={coop} ok", flush=True)
print(f"{rounds} rounds, {bad} mismatches, {time.time() - t_start:.0f} s")
sys.exit(1 if bad else 0)
