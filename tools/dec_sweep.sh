# Development aid: decode1080p batch time of several library variants, interleaved on one box
# usage: bash tools/dec_sweep.sh lib1.so lib2.so ...
for rep in 1 2; do
for lib in "$@"; do
  echo -n "$lib : "
  VP8L_AMD_LIB=$PWD/webp_amd/csrc/$lib timeout -k 10 200 python bench.py --workload decode1080p --steps 3 --warmup 1 --no-cpu --no-host 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f ms %.2f' % (d['value'], d['ms_per_step']))"
done
done
