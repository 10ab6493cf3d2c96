#!/bin/bash
# SQ issue / wait counters of one C3 step.  usage: tools/pmc_sq.sh <out dir under gpurun_out> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-clips 0 --streamed-steps 0 > $OUT.log 2>&1
