#!/bin/bash
# same-box A/B of library builds on BASELINE configs[3] (ViT-B/16 + 1M x 768 index): tools/ab_c4.sh out.txt libA.so libB.so ... ("-" = in-tree)
out=$1; shift
: > "$out"
for lib in "$@"; do
  echo "=== lib $lib" >> "$out"
  if [ "$lib" = "-" ]; then unset EFFOCR_HIP_LIB; else export EFFOCR_HIP_LIB=$PWD/$lib; fi
  python bench.py --arch vit_base_patch16_224 --index-rows 1000000 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --breakdown 2>> "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], 'e2e', d.get('encoder_mfma_frac_end_to_end'))" >> "$out"
done
unset EFFOCR_HIP_LIB
