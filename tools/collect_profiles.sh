#!/bin/bash
# Profiles of the benchmark command on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of `bench.py` (per-kernel average durations)
#   2. two counter passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass) restricted to the per-pixel encode kernels
#   3. kernel trace + stats of the other workloads of bench.py
# Results land in gpurun_out/<tag>/; tools/summarize_profiles.py turns them into the files under profiles/.
set -e
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-host --no-verify --no-configs > $OUT/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_lz77_sparse|k_parse_mark|k_bitcount|k_emit" --output-format csv -d $OUT/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-host --no-verify --no-configs > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_lz77_sparse|k_parse_mark|k_bitcount|k_emit" --output-format csv -d $OUT/write -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-host --no-verify --no-configs > $OUT/write.log 2>&1
for W in decode1080p palette2k single16k predictor4k; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o trace -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu --no-host > $OUT/trace_$W.log 2>&1
done
find $OUT -name "*_kernel_stats.csv" | head -20
