#!/bin/bash
# scripts/variants.sh <out-file> <name...>: the default bench (no CPU leg), twice per library variant (hiphase_amd/libhiphase_gpu_<name>.so; "main" = the product build)
OUT=$1; shift
for r in 1 2; do
  for v in "$@"; do
    if [ "$v" = main ]; then L=$(python bench.py --no-cpu --steps 20 --warmup 6 2>/dev/null | tail -1)
    else L=$(HP_LIB=hiphase_amd/libhiphase_gpu_$v.so python bench.py --no-cpu --steps 20 --warmup 6 2>/dev/null | tail -1); fi
    echo "$L" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', 'streamed ms/step', round(d['ms_per_step'],1), 'wfa span', round(d['roofline']['kernel_ms'],1), 'resident', round(d['resident']['ms_per_step'],1))" >> $OUT
  done
done
