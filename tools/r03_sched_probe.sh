#!/bin/bash
# ON THE GPU BOX: HBM traffic of the bench step for a scheduler task length (ICPX_SCHED_STEPS), and the whole config-5 batch
set -e
export TMPDIR=/tmp
out=gpurun_out
S=$1
export ICPX_SCHED_STEPS=$S
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --verify-pairs 0 --no-executed-work"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/sched${S}_fetch -- python3 $B > $out/sched${S}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/sched${S}_write -- python3 $B > $out/sched${S}_write.log 2>&1
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --pairs-total 8192 --no-cpu-baseline --no-extras --verify-pairs 0 --no-executed-work > $out/sched${S}_full.json 2>/dev/null
echo done $S
