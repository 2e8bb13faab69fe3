#!/bin/bash
# compile-time ablation sweep (scratch helper for gpurun): $1 = source file to touch, rest = EXP values
F=$1; shift
for e in "$@"; do
  touch effocr_amd/csrc/$F
  make -s -C effocr_amd/csrc EXP=$e 2>&1 | grep -E "error" | head -3
  echo "== EXP $e"; python bench.py --steps 5 --warmup 2 --breakdown --no-cpu-baseline 2>&1 | grep -E "panel_|fc2|attention|fc1|qkv|proj|mlp" | cut -c1-110
done
touch effocr_amd/csrc/$F; make -s -C effocr_amd/csrc EXP=0 2>&1 | grep error
