#!/bin/bash
# compile-time ablation sweep of the fused MLP kernel on the GPU box: args = EFFOCR_EXP values (2000+bits)
for e in "$@"; do
  touch effocr_amd/csrc/mlp_kernel.hpp
  make -s -j4 -C effocr_amd/csrc EXP=$e 2>&1 | grep -E "error" | head -3
  echo "== EXP $e"; python bench.py --steps 5 --warmup 2 --breakdown --no-cpu-baseline 2>&1 | grep -E "qkv_attn|mlp_fused|proj_resid|sum of" | cut -c1-110
done
touch effocr_amd/csrc/mlp_kernel.hpp; make -s -j4 -C effocr_amd/csrc EXP=0 2>&1 | grep error
