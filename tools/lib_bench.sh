# Development aid: the headline bench under several builds of the library, interleaved on one box
# usage: bash tools/lib_bench.sh lib1.so lib2.so ...
for rep in 1 2; do
for lib in "$@"; do
  echo -n "bench $lib : "
  VP8L_AMD_LIB=$PWD/webp_amd/csrc/$lib timeout -k 10 250 python bench.py --steps 3 --warmup 1 --no-cpu --no-host --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f ms %.2f' % (d['value'], d['ms_per_step']), {k:round(v,2) for k,v in d.get('stage_ms_per_step',{}).items() if k in ('match','match_fallback','parse')})" || exit 1
done
done
