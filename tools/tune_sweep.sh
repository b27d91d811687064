# Development aid: match-stage time of the benchmark batch under several libraries / VP8L_TUNE settings, interleaved on one box
# usage: bash tools/tune_sweep.sh "lib.so|tune" "lib.so|tune" ...   (tune may be empty)
for rep in 1 2; do
for spec in "$@"; do
  lib=${spec%%|*}; t=${spec#*|}
  echo -n "$lib VP8L_TUNE=$t : "
  VP8L_AMD_LIB=$PWD/webp_amd/csrc/$lib VP8L_TUNE=$t timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-host --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f match %.2f parse %.2f ms step %.2f ms' % (d['value'], d['stage_ms_per_step']['match'], d['stage_ms_per_step']['parse'], d['ms_per_step']))"
done
done
