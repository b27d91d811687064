#!/bin/bash
# Kernel trace + stats of one bench.py workload on the GPU box: tools/prof_workload.sh <workload> <tag> [bench args...]
# (run through gpurun from the repo root).  The per-kernel summary lands in gpurun_out/<tag>/<workload>_kernel_stats.csv.
set -e
W=$1; TAG=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o trace -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu --no-host "$@" > $OUT/trace_$W.log 2>&1
f=$(find $OUT/trace_$W -name "*_kernel_stats.csv" | head -1)
cp "$f" $OUT/${W}_kernel_stats.csv
head -14 $OUT/${W}_kernel_stats.csv | cut -c1-150
