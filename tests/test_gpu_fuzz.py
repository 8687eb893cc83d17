"""Randomized differential test of the fused kernels against the oracle (tests/tools/fuzz_filter.py):
random dimension, method, sizes, scales from 1e-6 to 1e6, offsets up to 1e5 x scale, lattices,
duplicated targets, collinear targets, with and without the iteration-granular scheduler."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,prune,chunk", [(3, "0", "1"), (11, "0", "0"), (3, "1", "1"), (17, "1", "1"), (23, "0", "1")])
def test_fuzz_fused_kernels_against_oracle(seed, prune, chunk):
    """prune = "1": handles and batches go through the grid-pruned kernels (icpx_prune.h);
    "0": through the brute-force kernels — trimmed ICP2 with the chunk-skipping exact scan (chunk = "1", the default)
    or, like the 3D methods, with the f32 filter + exact refinement (chunk = "0")."""
    env = dict(os.environ, ICPX_PRUNE=prune, ICPX_PRUNE_HANDLES=prune, ICPX_CHUNK_SKIP=chunk)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_filter.py"), str(seed), "60"],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 mismatches" in out.stdout



def test_grid_filter_random_shapes():
    """tests/tools/grid_filter_stress.py, twelve random cases: the 3D grid converge through f32 filter records and the
    lean sorted-domain kernel against the f64 records (ICPX_GRID_FILTER=0) and against unseeded queries
    (ICPX_GRID_SEED=0) — uniform, clustered, planar, duplicated, anisotropic and lattice clouds, far from the origin and
    at tiny scale, sources sticking out of the targets' box: iteration counts, transforms, every neighbour and distance
    bit-identical."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "grid_filter_stress.py"), "12", "11"],
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "all identical" in out.stdout
