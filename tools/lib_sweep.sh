# Development aid: bench workloads under several builds of the library, interleaved on one box
# usage: bash tools/lib_sweep.sh "workload1 workload2" lib1.so lib2.so ...
wls=$1; shift
for wl in $wls; do
for lib in "$@"; do
  echo -n "$wl $lib : "
  VP8L_AMD_LIB=$PWD/webp_amd/csrc/$lib timeout -k 10 250 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu --no-host 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f ms %.2f' % (d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d.get('stage_ms_per_step',{}).items() if k in ('match','match_fallback','parse')}, d.get('roundtrip_verified'))" || exit 1
done
done
