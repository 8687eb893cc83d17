"""Runs tests/cpp/test_dropin (the reference's own integration tests re-typed against the drop-in
C++ headers include/icp/*.h) on the GPU box.  The binary is built by __graft_entry__.build()."""
import os
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
