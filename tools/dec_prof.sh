# Development aid: per-kernel times of the decode1080p batch for library variants (rocprofv3 kernel trace)
# usage (GPU box, repo root): bash tools/dec_prof.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  export VP8L_AMD_LIB=$GRAFT_REPO_ROOT/webp_amd/csrc/$lib
  rm -rf gpurun_out/dp_$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dp_$lib -o t -- python3 bench.py --workload decode1080p --steps 3 --warmup 1 --no-cpu --no-host > gpurun_out/dp_$lib.log 2>&1 || exit 1
  echo "== $lib"
  f=$(find gpurun_out/dp_$lib -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "k_par" in n or "k_decode" in n or "k_inverse" in n:
        print("  %-28s calls %4s total %8.2f ms avg %8.3f ms" % (n.split("(")[0].replace("vp8l::",""), r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
done
