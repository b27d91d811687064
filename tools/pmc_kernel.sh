#!/bin/bash
# Development aid: SQ counters of one kernel (two passes of 8 counters) over tools/stage_probe.py.
# usage (on the GPU box, from the repo root): bash tools/pmc_kernel.sh TAG KERNEL_REGEX MODE BATCH [W H]
TAG=${1:-pmc}; KERN=${2:-k_lz77_sparse}; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --kernel-include-regex "$KERN" --output-format csv -d $OUT/p1 -o p1 -- python3 tools/stage_probe.py "$@" > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES --kernel-include-regex "$KERN" --output-format csv -d $OUT/p2 -o p2 -- python3 tools/stage_probe.py "$@" > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob,collections
for p in ("p1","p2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%p, recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k in acc: print(k, acc[k]/n[k], "per launch over", n[k], "launches")
PY
