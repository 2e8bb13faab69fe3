#!/bin/bash
# quick chunk-size sweep (scratch helper for gpurun)
for c in 0 512 256 128 64; do echo "== chunk $c"; python bench.py --steps 8 --warmup 2 --no-cpu-baseline --breakdown --chunk $c 2>&1 | grep -vE "amdgpu.ids" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], 'ms', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])
    else: print(l.rstrip())
"; done
