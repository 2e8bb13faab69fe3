#!/bin/bash
# compile-time ablation sweep of the panel kernel (scratch helper for gpurun)
for e in 0 1 2 3 4 5; do
  touch effocr_amd/csrc/panel.hip
  make -s -C effocr_amd/csrc EXP=$e 2>&1 | grep -E "error" | head -3
  echo "== EXP $e"; python bench.py --steps 5 --warmup 2 --breakdown --no-cpu-baseline 2>&1 | grep -E "panel_" | cut -c1-100
done
touch effocr_amd/csrc/panel.hip; make -s -C effocr_amd/csrc EXP=0 2>&1 | grep error
