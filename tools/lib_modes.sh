# Development aid: stage times per content kind under several builds of the library
# usage: bash tools/lib_modes.sh "3 4" lib1.so lib2.so ...
modes=$1; shift
for m in $modes; do
for lib in "$@"; do
  echo -n "mode $m $lib : "
  VP8L_AMD_LIB=$PWD/webp_amd/csrc/$lib timeout -k 10 100 python tools/stage_probe.py $m 32 3840 2160 2>&1 | grep -E "match " | head -1
done
done
