#!/bin/bash
# development aid: per-kernel durations of config 4's correspondence step, cooperative query on / off
export TMPDIR=/tmp
for coop in 1 0; do
  export ICPX_GRID_COOP=$coop
  rm -rf gpurun_out/c4k_$coop
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4k_$coop -- python3 tools/c4_grid_probe.py 3 > gpurun_out/c4k_$coop.log 2>&1 || exit 1
  echo "== coop=$coop"
  python3 - gpurun_out/c4k_$coop <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 1.0:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>3s} avg={float(r['AverageNs'])/1e3:8.1f} us")
PY
done
