"""Test infrastructure: checks a sample of a Config5Shard's results (icp_amd/workload.py) against the CPU
oracle.  Used by tests/test_gpu_bench_shape.py and by bench.py AFTER its timed region (the oracle is the
checker there, never the thing measured)."""
import numpy as np

from icp_amd.workload import MAX_ITER, TOL, sample_pairs, transform_of


def verify_sample(ob, shard_obj, count_per_dim: int) -> int:
    """Checks `count_per_dim` pairs of each kind of the shard's last step against the CPU oracle in device
    order (brute-force (d2, index) NN, TREE64 sums), bit for bit.  Returns the number of pairs verified;
    raises AssertionError on the first mismatch.  Test infrastructure: called by tests and by bench.py
    after its timed region only."""
    st2, st3 = shard_obj.states()
    ocrit = ob.make_criteria(max_iterations=MAX_ITER, transform_tol=TOL)
    done = 0
    for dim, st, method in ((2, st2, "trimmed"), (3, st3, "vanilla")):
        if len(st) == 0:
            continue
        for k in sample_pairs(st["it"].astype(np.int64), count_per_dim):
            a, b = shard_obj.host_pair(dim, k)
            ro = ob.OracleICP(dim, method, nn_mode=ob.NN_KDCANON, sum_mode=ob.SUM_TREE64).converge(a, b, None, ocrit)
            assert int(st["it"][k]) == ro.iteration_count, (dim, k, int(st["it"][k]), ro.iteration_count)
            assert np.array_equal(transform_of(st[k], dim), ro.transform), (dim, k)
            assert st["cost"][k] == ro.cost, (dim, k)
            done += 1
    return done
