# Development aid: one bench workload under several VP8L_TUNE settings
# usage: bash tools/wl_sweep.sh WORKLOAD "tune1" "tune2" ...   ("" = defaults)
wl=$1; shift
for t in "$@"; do
  echo -n "$wl VP8L_TUNE=$t : "
  VP8L_TUNE=$t timeout -k 10 250 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --no-host 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f ms %.2f' % (d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d.get('stage_ms_per_step',{}).items() if k in ('match','match_fallback','parse')}, d.get('dense_tiles'))"
done
