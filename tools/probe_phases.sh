#!/bin/bash
# development aid (needs `make -C icp_amd/csrc probe`): time of a fixed-iteration batch with phases switched off
export ICPX_LIB=$PWD/icp_amd/lib_probe/libicpx.so
for mask in 0 16 32 64 48 96 112; do
  export ICPX_PROBE_MASK=$mask
  echo -n "mask=$mask 3D 256x8: "; timeout -k 10 100 python tools/perf_probe.py 3 4096 256 8 vanilla 3 2>&1 | grep dim= | cut -c50-70
  echo -n "mask=$mask 2D 256x8: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 256 8 trimmed 3 2>&1 | grep dim= | cut -c50-70
done
