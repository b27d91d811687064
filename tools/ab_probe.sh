# Development aid: per-stage encoder times of library variants, interleaved on one box
# usage: bash tools/ab_probe.sh "MODE BATCH" lib1.so lib2.so ...
set -e
what=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib"
  VP8L_AMD_LIB=$PWD/webp_amd/csrc/$lib timeout -k 10 120 python tools/stage_probe.py $what 2>/dev/null | head -2
done
done
