#!/bin/bash
# development aid: per-iteration latency of one pair per compute unit (256 pairs, no scheduler) from the slope over
# the iteration count, and throughput at 512 / 1024 pairs; coop on and off
for coop in 1 0; do
  export ICPX_COOP=$coop
  for it in 1 2 4 8; do
    echo -n "coop=$coop 3D 256 pairs iters=$it: "; timeout -k 10 100 python tools/perf_probe.py 3 4096 256 $it vanilla 3 2>&1 | grep dim= | cut -c1-110
  done
  for it in 1 2 4 8; do
    echo -n "coop=$coop 2D 256 pairs iters=$it: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 256 $it trimmed 3 2>&1 | grep dim= | cut -c1-110
  done
  echo -n "coop=$coop 3D 1024 pairs iters=4: "; timeout -k 10 100 python tools/perf_probe.py 3 4096 1024 4 vanilla 3 2>&1 | grep dim= | cut -c1-110
  echo -n "coop=$coop 2D 2048 pairs iters=8: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 2048 8 trimmed 3 2>&1 | grep dim= | cut -c1-110
done
