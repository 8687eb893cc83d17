"""CPU tests of the product's host side: the C-ABI library loads and exports every declared
symbol, fails loudly without a GPU, the Python mirror keeps the reference's interface semantics,
the readers parse the reference's data, and the N>1 sharding path works (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    from icp_amd import _lib
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "icpx", "icpx.h")) as f:
        text = f.read()
    declared = set(re.findall(r"\b(icpx_[a-z_0-9]+)\s*\(", text))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.icpx_version() == 10000


def test_struct_layouts_match_header():
    import ctypes as C
    from icp_amd import _lib
    assert C.sizeof(_lib.State) == 144
    assert C.sizeof(_lib.Criteria) == 72
    assert C.sizeof(_lib.Params) == 48


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_fails_loudly_without_device():
    import icp_amd
    with pytest.raises(icp_amd.IcpxError) as e:
        icp_amd.ICP2.from_method("vanilla")
    assert e.value.code == icp_amd._lib.ERR_NO_DEVICE
    with pytest.raises(icp_amd.IcpxError):
        icp_amd.nearest_neighbors(np.zeros((4, 2)), np.zeros((4, 2)))


def test_registry_and_config_semantics():
    import icp_amd
    assert sorted(icp_amd.ICP2.registered_methods()) == ["feature_aware", "trimmed", "vanilla"]
    assert sorted(icp_amd.ICP3.registered_methods()) == ["trimmed", "vanilla"]
    assert icp_amd.ICP2.is_method_registered("trimmed")
    assert not icp_amd.ICP2.is_method_registered("nope")
    assert icp_amd.ICP2.from_method("nope") is None  # empty optional, icp.h:97-99
    c = icp_amd.Config()
    c.set("overlap_rate", 0.7)
    c.set("symmetric_neighbors", 1)
    assert c.get("overlap_rate", 0.9) == 0.7
    assert c.get("missing", 3) == 3
    with pytest.raises(TypeError):  # std::bad_variant_access, config.h:30
        c.get("overlap_rate", 1)
    with pytest.raises(TypeError):
        c.get("symmetric_neighbors", 1.0)


def test_driver_setters_assert_like_reference():
    import icp_amd

    class Fake:
        dim = 2
    d = icp_amd.ICPDriver(Fake())
    d.set_min_iterations(5)
    with pytest.raises(AssertionError):  # driver.h:91
        d.set_max_iterations(3)
    d.set_max_iterations(10)
    with pytest.raises(AssertionError):  # driver.h:81
        d.set_min_iterations(11)
    d.set_transform_tolerance(0.1, 0.2)
    d.set_stop_cost(1.0)
    c = d.criteria()
    from icp_amd import _lib
    assert c.has == (_lib.HAS_MIN_ITER | _lib.HAS_MAX_ITER | _lib.HAS_TRANSFORM_TOL | _lib.HAS_STOP_COST)
    assert (c.min_iterations, c.max_iterations) == (5, 10)


def test_parse_lidar_scan_matches_reference_semantics(tmp_path):
    from icp_amd import io as iio
    sizes = {1: (953, 959), 2: (766, 768), 3: (562, 661)}  # +-inf rows dropped
    for k, (n, m) in sizes.items():
        a = iio.parse_lidar_scan(f"{GOLDEN}/ex_data/scan{k}/first.csv")
        b = iio.parse_lidar_scan(f"{GOLDEN}/ex_data/scan{k}/second.csv")
        assert a.shape == (n, 2) and b.shape == (m, 2)
        assert np.isfinite(a).all() and np.isfinite(b).all()
    p = tmp_path / "s.csv"
    p.write_text("1.5,2.5\n-inf,3\n4,inf\n7e-1,-2\n")
    assert iio.parse_lidar_scan(str(p)).tolist() == [[1.5, 2.5], [0.7, -2.0]]
    with pytest.raises(RuntimeError):
        iio.parse_lidar_scan(str(tmp_path / "missing.csv"))


def test_ply_reader():
    from icp_amd import io as iio
    q = iio.read_ply_vertices(f"{GOLDEN}/ex_data/ply/queen.ply")
    qt = iio.read_ply_vertices(f"{GOLDEN}/ex_data/ply/queen_transformed.ply")
    assert q.shape == qt.shape == (10720, 3)
    assert len(np.unique(q, axis=0)) == 8764  # duplicate vertices
    # same vertex order, rigid motion of 15 degrees about z up to float32 rounding
    ca, cb = qt.mean(0), q.mean(0)
    U, _, Vt = np.linalg.svd((qt - ca).T @ (q - cb))
    R = (U @ Vt).T
    assert np.abs((qt - ca) @ R.T + cb - q).max() < 1e-4
    assert abs(np.degrees(np.arccos((np.trace(R) - 1) / 2)) - 15.0) < 1e-3


def test_synth_is_deterministic():
    from icp_amd import synth
    s1, t1, off, T = synth.config5_pairs([3, 9], 2, 64)
    s2, t2, _, _ = synth.config5_pairs([9], 2, 64)
    assert np.array_equal(s1[64:], s2) and np.array_equal(t1[64:], t2)
    assert off.tolist() == [0, 64, 128]
    src, tgt, R, t = synth.config4(1000)
    assert np.abs(src - (tgt @ R.T + t)).max() < 0.08
    assert tgt.min() >= -10 and tgt.max() < 10
    # splitmix64 known answer (seed 0 -> first output of the reference splitmix64 stream)
    assert int(synth.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF


def test_shard_plumbing_gloo_world2(tmp_path):
    """N>1 path: pairs sharded over ranks with no data-path collective, results gathered once.
    Two CPU ranks over gloo; the per-pair solver here is the oracle (test infrastructure)."""
    script = tmp_path / "w.py"
    script.write_text(f"""
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch, torch.distributed as dist
from icp_amd import shard, synth
from oracle import binding as ob
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
P = 6
lo, hi = shard.shard_range(P, rank, world)
src, dst, off, _ = synth.config5_pairs(range(lo, hi), 2, 48)
crit = ob.make_criteria(max_iterations=20, transform_tol=(1e-4, 1e-4))
states = np.zeros((hi - lo, 18))
for k in range(hi - lo):
    r = ob.OracleICP(2, 'trimmed').converge(src[off[k]:off[k+1]], dst[off[k]:off[k+1]], None, crit)
    states[k, 0] = r.cost; states[k, 1] = r.iteration_count; states[k, 2:11] = r.transform.flatten(order='F')
allst = shard.gather_states(torch.from_numpy(states), P, rank, world)
if rank == 0:
    src, dst, off, _ = synth.config5_pairs(range(P), 2, 48)
    for k in range(P):
        r = ob.OracleICP(2, 'trimmed').converge(src[off[k]:off[k+1]], dst[off[k]:off[k+1]], None, crit)
        assert allst[k, 1] == r.iteration_count and allst[k, 0] == r.cost, k
        assert np.array_equal(allst[k, 2:11].numpy(), r.transform.flatten(order='F'))
# ragged shards and several independently sharded groups, as bench.py gathers them (2D pairs, then 3D pairs):
# row r of a rank's block carries (group, global pair id)
for groups in ((7,), (5, 4), (3, 3), (1, 6), (4, 2), (8,)):
    mine = []
    for gi, g in enumerate(groups):
        lo, hi = shard.shard_range(g, rank, world)
        mine += [[gi, p] + [0.0] * 16 for p in range(lo, hi)]
    local = torch.tensor(mine, dtype=torch.float64).reshape(-1, 18)
    scratch = {{}}
    for rep in range(2):   # (the second call reuses the scratch buffer)
        tab = shard.gather_states(local, sum(groups), rank, world, counts=groups, scratch=scratch)
        want = []
        for r in range(world):
            for gi, g in enumerate(groups):
                lo, hi = shard.shard_range(g, r, world)
                want += [[gi, p] for p in range(lo, hi)]
        assert tab.shape[0] == sum(groups) and tab[:, :2].tolist() == [[float(a), float(b)] for a, b in want], (groups, tab[:, :2])
if rank == 0:
    print('OK')
dist.destroy_process_group()
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK" in out.stdout


def test_cmake_package_exports_reference_target(tmp_path):
    """find_package(CevIcp) + CEV::CevIcp, the names the reference exports (CMakeLists.txt:94-100,198-218), resolve to
    libicpx.so and include/; a consumer compiles and links against them (running it needs no GPU: registry only)."""
    import shutil
    if shutil.which("cmake") is None or shutil.which("make") is None:
        pytest.skip("cmake / make not on PATH")
    (tmp_path / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.16)\nproject(consumer CXX)\nfind_package(CevIcp 1.3 REQUIRED)\n"
        "add_executable(app main.cpp)\ntarget_link_libraries(app PRIVATE CEV::CevIcp)\n")
    (tmp_path / "main.cpp").write_text(
        '#include <cstdio>\n#include "icp/icp.h"\n#include "icp/driver.h"\n'
        'int main() { std::printf("%d %d\\n", (int)icp::ICP2::is_method_registered("trimmed"), '
        '(int)icp::ICP3::registered_methods().size()); return 0; }\n')
    cfg = subprocess.run(["cmake", "-S", str(tmp_path), "-B", str(tmp_path / "b"), "-G", "Unix Makefiles",
                          f"-DCevIcp_DIR={ROOT}/cmake"], capture_output=True, text=True, timeout=120)
    assert cfg.returncode == 0, cfg.stdout[-1500:] + cfg.stderr[-1500:]
    bld = subprocess.run(["cmake", "--build", str(tmp_path / "b")], capture_output=True, text=True, timeout=240)
    assert bld.returncode == 0, bld.stdout[-1500:] + bld.stderr[-1500:]
    run = subprocess.run([str(tmp_path / "b" / "app")], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0 and run.stdout.split() == ["1", "2"], run.stdout + run.stderr


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (before
    anything touches a GPU); --dry-ranks runs the rank plumbing on the CPU over gloo (shard ranges, result gather)
    with no kernels.  A WORLD_SIZE that disagrees with --gpus is an error, not a warning."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks", "--steps", "2"],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = lines[0]
    assert line["ranks"] == 2 and line["n_gpus"] == 2 and line["gather_ok"]
    assert line["pairs_total"] == 2048 and line["pairs_per_rank"] == 1024   # weak scaling: 512 + 512 per rank
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks"],
                         capture_output=True, text=True, env=dict(env, WORLD_SIZE="3"), timeout=60)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_batch_rejects_criteria_that_never_terminate():
    """the converge loop runs inside one kernel, so criteria under which it cannot end are an
    argument error (checked before any device is touched, hence testable here)."""
    import ctypes as C
    from icp_amd import _lib
    from icp_amd.api import make_criteria
    lib = _lib.load()
    pts = np.zeros((8, 2))
    off = np.array([0, 8], dtype=np.int64)
    st = (_lib.State * 1)()
    for kw in (dict(), dict(stop_cost=1.0), dict(relative=1e-3, absolute=1e-3), dict(min_iterations=5, max_iterations=2)):
        crit = make_criteria(**kw)
        for fn in (lib.icpx_converge_batch,):
            rc = fn(2, _lib.TRIMMED, None, 1, pts.ctypes.data, off.ctypes.data, pts.ctypes.data, off.ctypes.data, None,
                    C.byref(crit), C.addressof(st))
            assert rc == _lib.ERR_INVALID, kw
        dev = (C.c_int * 1)(0)
        rc = lib.icpx_converge_batch_multi(2, _lib.TRIMMED, None, 1, dev, 1, pts.ctypes.data, off.ctypes.data,
                                           pts.ctypes.data, off.ctypes.data, None, C.byref(crit), C.addressof(st))
        assert rc == _lib.ERR_INVALID, kw
    assert b"max_iterations" in lib.icpx_last_error()
    assert lib.icpx_debug_knob(b"ICPX_NOPE", b"1") == _lib.ERR_INVALID
    assert lib.icpx_debug_knob(b"ICPX_SCHED", b"1") == _lib.OK and lib.icpx_debug_knob(b"ICPX_SCHED", None) == _lib.OK
