#!/bin/bash
# development aid: the three probes used to judge a change of the fused kernels
timeout -k 10 100 python tools/perf_probe.py 2 1024 2048 20 trimmed 3 2>&1 | grep dim= | cut -c1-95
timeout -k 10 100 python tools/perf_probe.py 3 4096 1024 5 vanilla 3 2>&1 | grep dim= | cut -c1-95
timeout -k 10 100 python tools/perf_probe.py 3 1024 2048 10 vanilla 3 2>&1 | grep dim= | cut -c1-95
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('bench', d['value'],d['ms_per_step'],d['roofline']['launch_ms'])"
