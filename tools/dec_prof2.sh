# Development aid: per-kernel decode times for one content kind (rocprofv3 kernel trace of tools/stage_probe.py)
# usage (GPU box, repo root): bash tools/dec_prof2.sh "MODE BATCH [W H]" [lib.so]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
what=$1; lib=${2:-libvp8l_amd.so}
export VP8L_AMD_LIB=$GRAFT_REPO_ROOT/webp_amd/csrc/$lib
tag=$(echo "$what $lib" | tr ' ' '_')
rm -rf gpurun_out/dq_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dq_$tag -o t -- python3 tools/stage_probe.py $what > gpurun_out/dq_$tag.log 2>&1 || exit 1
echo "== $what $lib"
f=$(find gpurun_out/dq_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "k_par" in n or "k_decode" in n or "k_inverse" in n:
        print("  %-28s calls %4s total %8.2f ms avg %8.3f ms" % (n.split("(")[0].replace("vp8l::",""), r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
