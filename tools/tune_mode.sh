# Development aid: stage times of one content kind with the predictor on under several VP8L_TUNE settings
# usage: bash tools/tune_mode.sh MODE "tune1" "tune2" ...
m=$1; shift
for t in "$@"; do
  echo "mode $m VP8L_TUNE=$t"
  VP8L_TUNE=$t timeout -k 10 100 python tools/stage_probe.py $m 16 3840 2160 --predictor=5 2>&1 | grep -E "dense tiles|match" | head -2
done
