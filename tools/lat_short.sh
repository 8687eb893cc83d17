export ICPX_COOP=1
for it in 1 2 4 8 16; do
  echo -n "3D 256 pairs iters=$it: "; timeout -k 10 100 python tools/perf_probe.py 3 4096 256 $it vanilla 3 2>&1 | grep dim= | cut -c1-110 || exit 1
done
for it in 1 2 4 8 16; do
  echo -n "2D 256 pairs iters=$it: "; timeout -k 10 100 python tools/perf_probe.py 2 1024 256 $it trimmed 3 2>&1 | grep dim= | cut -c1-110 || exit 1
done
