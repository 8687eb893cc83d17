"""Runs tests/cpp/test_dropin (the reference's own integration tests re-typed against the drop-in
C++ headers include/icp/*.h) and the bench report programs on the GPU box.  The binaries are built
by __graft_entry__.build()."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_dropin_binary():
    exe = os.path.join(ROOT, "tests", "cpp", "test_dropin")
    assert os.path.exists(exe), "tests/cpp/test_dropin missing: run __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "0 failed" in out.stdout


def _report(exe, *args):
    out = subprocess.run([exe, *args], capture_output=True, text=True, timeout=240,
                         cwd=os.path.join(ROOT, "tests", "golden"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def _fields(text):
    """Report lines of the reference format, minus the two wall-clock lines."""
    return [ln for ln in text.splitlines() if not ln.startswith("+ ") and "time per" not in ln]


def test_bench_report_program(traces):
    """bench/icp_bench prints the reference's `make bench` report (bench.cpp:88-142): blocks for
    every registered ICP2 method x scan 1-3; iteration counts and costs equal the oracle's."""
    exe = os.path.join(ROOT, "bench", "icp_bench")
    assert os.path.exists(exe), "bench/icp_bench missing: run __graft_entry__.build()"
    text = _report(exe, ".", "3")
    assert text.startswith("ICP ALGORITHM BENCHMARKING\n")
    blocks = re.findall(r"\* Method: (\w+)\n\* Number of invocations: 3\n.*?\* Median cost: (\S+)\n.*?"
                        r"\* Min iterations: (\d+)\n\* Max iterations: (\d+)\n", text, flags=re.S)
    # three scans per registered ICP2 method (lib/icp/icp.cpp:17-21), in registry order
    assert len(blocks) == 9
    methods = [b[0] for b in blocks]
    assert sorted(set(methods)) == ["feature_aware", "trimmed", "vanilla"]
    assert all(methods[i] == methods[i - i % 3] for i in range(9))
    for i, (method, cost, lo, hi) in enumerate(blocks):
        assert int(lo) == int(hi)
        if method == "feature_aware":
            continue  # pinned against the oracle in test_gpu_parity.py
        tr = traces["scan%d/%s/dev" % (i % 3 + 1, method)]
        assert int(lo) == tr["iterations"], (method, i)
        assert abs(float(cost) - tr["cost"]) <= 1e-5 * max(1.0, abs(tr["cost"]))  # 6 printed digits
    assert "+ NN correspondence queries/s:" in text


def test_reference_bench_cpp_unchanged_against_dropin_headers():
    """The reference's own bench/bench.cpp, compiled unchanged against include/icp + libicpx.so
    (bench/Makefile, only where the reference checkout exists), prints the same report as
    icp_bench (wall-clock lines aside)."""
    ref = os.path.join(ROOT, "bench", "_ref", "bench_ref")
    if not os.path.exists(ref):
        pytest.skip("bench/_ref/bench_ref was not built (no reference checkout at build time)")
    ours = _fields(_report(os.path.join(ROOT, "bench", "icp_bench"), ".", "10"))
    theirs = _fields(_report(ref))
    assert ours == theirs
