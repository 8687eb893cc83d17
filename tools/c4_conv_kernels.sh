#!/bin/bash
# development aid: per-kernel durations of config 4's converge — f32 filter records on / off, opt-in f32 mode
export TMPDIR=/tmp
run() {  # tag, env assignment, probe args
  rm -rf gpurun_out/c4c_$1
  env_kv=$2
  shift 2
  export $env_kv
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4c_$tag -- python3 tools/c4_converge_probe.py "$@" > gpurun_out/c4c_$tag.log 2>&1 || exit 1
  echo "== $tag: $(tail -1 gpurun_out/c4c_$tag.log)"
  python3 - gpurun_out/c4c_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 1.0:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>3s} avg={float(r['AverageNs'])/1e3:8.1f} us")
PY
}
tag=filter;   run $tag ICPX_GRID_FILTER=1 "$@"
tag=nofilter; run $tag ICPX_GRID_FILTER=0 "$@"
tag=f32;      run $tag ICPX_GRID_FILTER=1 f32 "$@"
