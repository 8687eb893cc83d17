#!/bin/bash
# development aid (needs `make -C icp_amd/csrc probe`): masks 0 = normal, 512 = every seed counts as certified (no search
# beyond the seed evaluation), 16 = the seeded searches skipped altogether
export ICPX_LIB=$PWD/icp_amd/lib_probe/libicpx.so
for mask in 0 512 16; do
  export ICPX_PROBE_MASK=$mask
  echo -n "mask=$mask 2D 512x16: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 512 16 trimmed 3 2>&1 | grep dim= | cut -c50-70 || exit 1
  echo -n "mask=$mask 3D 256x8: "; timeout -k 10 100 python tools/perf_probe.py 3 4096 256 8 vanilla 3 2>&1 | grep dim= | cut -c50-70 || exit 1
done
