#!/bin/bash
# Builds the library in-tree; prints the errors and fails if the build fails.
cd "$(dirname "$0")/../webp_amd/csrc" || exit 1
if ! make "$@" > /tmp/vp8l_make.log 2>&1; then
  grep -E "error|Error" -A6 /tmp/vp8l_make.log | head -60
  exit 1
fi
echo "build ok"
