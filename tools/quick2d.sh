#!/bin/bash
# development aid: 2D trimmed batch timings (latency-bound shape and throughput shape)
for a in "512 16" "256 16" "2048 8" "4096 8"; do set -- $a
  echo -n "2D $1 pairs x $2 iters: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 $1 $2 trimmed 3 2>&1 | grep dim= | cut -c50-70 || exit 1
done
