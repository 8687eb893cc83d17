export TMPDIR=/tmp
for g in 4,512 8,256; do echo "2D n=2048 geom $g"; ICPX_GEOM=$g timeout -k 10 100 python -u tools/perf_probe.py 2 2048 512 8 trimmed 3 2>&1 | grep dim= | cut -c1-150 || exit 1; done
for g in 4,512 8,256; do echo "3D n=2048 geom $g"; ICPX_GEOM=$g timeout -k 10 100 python -u tools/perf_probe.py 3 2048 512 4 vanilla 3 2>&1 | grep dim= | cut -c1-150 || exit 1; done
for g in 2,256 1,512 4,128; do echo "2D n=512 geom $g"; ICPX_GEOM=$g timeout -k 10 100 python -u tools/perf_probe.py 2 512 2048 8 trimmed 3 2>&1 | grep dim= | cut -c1-150 || exit 1; done
