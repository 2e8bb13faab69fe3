#!/bin/bash
# Same-box A/B of library builds: tools/ab_bench.sh out.txt libA.so libB.so ...   (interleaved, 2 rounds; "-" = the in-tree build)
out=$1; shift
: > "$out"
for round in 1 2; do
  for lib in "$@"; do
    echo "=== round $round lib $lib" >> "$out"
    if [ "$lib" = "-" ]; then unset EFFOCR_HIP_LIB; else export EFFOCR_HIP_LIB=$PWD/$lib; fi
    python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --breakdown 2>> "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], 'e2e', d.get('encoder_mfma_frac_end_to_end'))" >> "$out"
  done
done
unset EFFOCR_HIP_LIB
