#!/bin/bash
# development aid: shapes x {default, pruned only, brute force only} for the sparse-grid hand-over threshold
set -e
for sh in sphere curve circle clump2 clump3; do
  timeout -k 10 80 python -u tools/surface_batch_probe.py $sh
  ICPX_DEFER_SPARSE=0 timeout -k 10 80 python -u tools/surface_batch_probe.py $sh
  ICPX_PRUNE=0 timeout -k 10 80 python -u tools/surface_batch_probe.py $sh
done
