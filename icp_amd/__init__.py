"""icp_amd — MI355X (gfx950) native ICP engine behind cornellev/icp's interface.

The product is the HIP library icp_amd/lib/libicpx.so (C ABI: include/icpx/icpx.h).  This package
is its Python host side: the mirror of the reference's ICP / ICPDriver / Config classes
(api.py), the data readers either side of the path (io.py) and the synthetic workloads the
benchmarks use (synth.py).
"""
from .api import (ICP2, ICP3, Config, ConvergenceState, ICPDriver, Match, converge_batch,
                  make_criteria, nearest_neighbors, nearest_neighbors_kdtree)
from ._lib import IcpxError, NN_AUTO, NN_BRUTE, NN_GRID, NN_BVH, NN_GRID_F32, TRIMMED, VANILLA

__all__ = ["ICP2", "ICP3", "Config", "ConvergenceState", "ICPDriver", "Match", "converge_batch",
           "make_criteria", "nearest_neighbors", "nearest_neighbors_kdtree", "IcpxError", "NN_AUTO", "NN_BRUTE", "NN_GRID", "NN_BVH", "NN_GRID_F32",
           "TRIMMED", "VANILLA"]
