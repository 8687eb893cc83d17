#!/bin/bash
# development aid (needs `make -C icp_amd/csrc probe`): 2D trimmed batch, 512 pairs x 16 iterations, phases switched off
# masks: 16 search, 32 sums, 64 Kabsch, 128 selection
export ICPX_LIB=$PWD/icp_amd/lib_probe/libicpx.so
for mask in 0 16 32 64 128 240; do
  export ICPX_PROBE_MASK=$mask
  echo -n "mask=$mask 2D 512x16: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 512 16 trimmed 3 2>&1 | grep dim= | cut -c50-70 || exit 1
  echo -n "mask=$mask 2D 256x16: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 256 16 trimmed 3 2>&1 | grep dim= | cut -c50-70 || exit 1
done
