"""Mints tests/golden/traces.json: per-iteration (cost, transform) traces and final
ConvergenceStates of the CPU oracle on the reference's example data.

The reference holds NO golden vectors for this path (its tests only assert 0.5 units / 0.01 rad),
and its method classes cannot be built here (Eigen is absent), so these vectors come from the
oracle restatement.  They pin the oracle against regressions and give the GPU tests a committed
target; they are not outputs of the reference itself.

Run:  python tests/tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icp_amd import io as iio  # noqa: E402
from oracle import binding as ob  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def run(dim, method, src, dst, crit_kw, nn_mode, sum_mode):
    o = ob.OracleICP(dim, method, nn_mode=nn_mode, sum_mode=sum_mode)
    r = o.converge(src, dst, None, ob.make_criteria(**crit_kw), trace_cap=128)
    return {
        "iterations": r.iteration_count,
        "cost": r.cost.hex(),
        "transform": [v.hex() for v in r.transform.flatten().tolist()],
        "trace_cost": [float(c).hex() for c in r.trace_cost],
        "trace_transform": [[v.hex() for v in t.flatten().tolist()] for t in r.trace_transform],
    }


def main():
    out = {"_doc": "hex floats; transforms row-major (dim+1)^2; modes: ref = kd-tree NN + "
                   "sequential sums (reference order), dev = kd-tree rule in canonical form (ORC_NN_KDCANON; brute force for trimmed ICP2) + TREE64 sums (device order)"}
    bench = dict(max_iterations=50, transform_tol=(0.1 * np.pi / 180, 0.01))  # bench/bench.cpp:109-113
    for k in (1, 2, 3):
        a = iio.parse_lidar_scan(f"{G}/ex_data/scan{k}/first.csv")
        b = iio.parse_lidar_scan(f"{G}/ex_data/scan{k}/second.csv")
        for method in ("vanilla", "trimmed"):
            out[f"scan{k}/{method}/ref"] = run(2, method, a, b, bench, ob.NN_KDTREE, ob.SUM_SEQUENTIAL)
            out[f"scan{k}/{method}/dev"] = run(2, method, a, b, bench, ob.NN_KDCANON, ob.SUM_TREE64)
    q = iio.read_ply_vertices(f"{G}/ex_data/ply/queen.ply")
    qt = iio.read_ply_vertices(f"{G}/ex_data/ply/queen_transformed.ply")
    ply = dict(max_iterations=100, transform_tol=(1e-4, 1e-4))  # tests/test_ply.cpp:22-23,104-105
    for method in ("vanilla", "trimmed"):
        out[f"queen/{method}/ref"] = run(3, method, qt, q, ply, ob.NN_KDTREE, ob.SUM_SEQUENTIAL)
        out[f"queen/{method}/dev"] = run(3, method, qt, q, ply, ob.NN_KDCANON, ob.SUM_TREE64)
    with open(os.path.join(G, "traces.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", os.path.join(G, "traces.json"))


if __name__ == "__main__":
    main()
