#!/bin/bash
# tools/f16_ab.sh out.txt lib ...   ("-" = in-tree), two interleaved rounds
out=$1; shift
: > "$out"
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset EFFOCR_HIP_LIB; else export EFFOCR_HIP_LIB=$PWD/$lib; fi
    python tools/f16_ab.py >> "$out" 2>&1
  done
done
unset EFFOCR_HIP_LIB
