#!/bin/bash
# development aid: A/B of two library builds (icp_amd/lib vs icp_amd/lib_alt) on the latency and bench shapes
for lib in lib lib_alt; do
  export ICPX_LIB=$PWD/icp_amd/$lib/libicpx.so
  echo "== $lib"
  for it in 1 2 4 8; do
    echo -n "3D 256 x $it: "; timeout -k 10 100 python tools/perf_probe.py 3 4096 256 $it vanilla 3 2>&1 | grep dim= | cut -c50-70 || exit 1
  done
  for a in "512 16" "2048 8"; do set -- $a
    echo -n "2D $1 x $2: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 $1 $2 trimmed 3 2>&1 | grep dim= | cut -c50-70 || exit 1
  done
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --verify-pairs 0 --no-executed-work | cut -c130-170,420-520 || exit 1
done
