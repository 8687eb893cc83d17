#!/bin/bash
# development aid (GPU box): instruction counters of a fixed-iteration batch (tools/perf_probe.py arguments follow)
export TMPDIR=/tmp
out=gpurun_out/pmc_probe
rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out -- python3 tools/perf_probe.py "$@" > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('gpurun_out/pmc_probe/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_INSTS_VALU': cnt[k] += 1
for k, v in agg.items():
    n = cnt[k] or 1
    print(k, 'dispatches', n, {c: round(x / n) for c, x in v.items()})
PY
