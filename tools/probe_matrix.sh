#!/bin/bash
# development aid: equal-work probes over filter / scheduler settings
set -e
for f in 0 1; do for s in 0 1; do
  echo "== ICPX_FILTER=$f ICPX_SCHED=$s"
  ICPX_FILTER=$f ICPX_SCHED=$s timeout -k 10 120 python tools/perf_probe.py 3 4096 256 1 vanilla 3
  ICPX_FILTER=$f ICPX_SCHED=$s timeout -k 10 120 python tools/perf_probe.py 3 4096 256 2 vanilla 3
  ICPX_FILTER=$f ICPX_SCHED=$s timeout -k 10 120 python tools/perf_probe.py 3 4096 256 10 vanilla 3
done; done
echo "== 2D"
for f in 0 1; do
  ICPX_FILTER=$f timeout -k 10 120 python tools/perf_probe.py 2 1024 512 10 trimmed 3
done
