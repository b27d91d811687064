#!/bin/bash
# HBM traffic of the decode kernels (BASELINE configs[3]: 512 streams of 1920x1080 per GPU) from two counter passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass).  On the GPU box, from the repo root:
#   tools/collect_decode_traffic.sh TAG      -> gpurun_out/TAG/{dfetch,dwrite}/...; then tools/summarize_decode_traffic.py TAG here
set -e
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KERNELS="k_par_|k_decode_headers|k_decode_entropy|k_inverse"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KERNELS" --output-format csv -d $OUT/dfetch -o dfetch -- python3 bench.py --workload decode1080p --steps 1 --warmup 0 --no-cpu > $OUT/dfetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KERNELS" --output-format csv -d $OUT/dwrite -o dwrite -- python3 bench.py --workload decode1080p --steps 1 --warmup 0 --no-cpu > $OUT/dwrite.log 2>&1
find $OUT -name "*counter_collection.csv" | head
