#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench line, the rocprofv3 kernel-trace summary of the
# same command, and the PMC passes (one counter group per run, never combined with traces); then the
# same for config 4's grid correspondence step alone (tools/c4_grid_probe.py).
# usage: tools/collect_profiles.sh <tag>      -> files under gpurun_out/<tag>_*
set -e
tag=${1:-rXX}
out=gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py --steps 10 --warmup 2 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --verify-pairs 0 --no-executed-work"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 $B > $out/${tag}_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- python3 $B > $out/${tag}_pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- python3 $B > $out/${tag}_pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_valu -- python3 $B > $out/${tag}_pmc_valu.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/${tag}_pmc_lds -- python3 $B > $out/${tag}_pmc_lds.log 2>&1
G="tools/c4_grid_probe.py 3"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c4_trace -- python3 $G > $out/${tag}_c4_trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_c4_pmc_fetch -- python3 $G > $out/${tag}_c4_pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_c4_pmc_write -- python3 $G > $out/${tag}_c4_pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_c4conv_trace -- python3 tools/c4_converge_probe.py > $out/${tag}_c4conv_trace.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_queen_trace -- python3 tools/queen_probe.py > $out/${tag}_queen_trace.log 2>&1
echo collected
