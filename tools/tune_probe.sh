#!/bin/bash
# usage: tune_probe.sh OUT "per_value,budget,max_bucket" ...   -- runs three workloads per setting
out=$1; shift
for t in "$@"; do
  for wl in palette2k predictor4k; do
    VP8L_TUNE=$t timeout -k 10 200 python bench.py --workload $wl --no-cpu --steps 2 > /tmp/tp.json 2>/dev/null || exit 1
    python - "$t" $wl <<'PY' >> $out
import json, sys
j = json.load(open("/tmp/tp.json"))
print(sys.argv[1], sys.argv[2], "%.0f MP/s" % j["value"], "match %.2f ms fallback %.2f ms" % (j["stage_ms_per_step"]["match"], j["stage_ms_per_step"]["match_fallback"]), j["exhaustive_search_tiles"], "ok" if j["roundtrip_verified"] else "BAD")
PY
  done
done
