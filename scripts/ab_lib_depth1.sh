#!/bin/bash
# A/B of library builds, kernels undisturbed: ab_lib_depth1.sh <steps> <variant|base> ... runs the streamed bench one set at a time
# (--depth 1: no other stage's kernels beside the launch set) and then at the default depth; prints the launch-set span and the step
STEPS=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset HP_LIB; else export HP_LIB=hiphase_amd/libhiphase_gpu_$v.so; fi
  for D in 1 0; do
    if [ $D = 1 ]; then X="--depth 1"; else X=""; fi
    timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --steps $STEPS $X 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
s = d['stage_ms']
print('$v', 'depth', d['config']['depth'], round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'wfa span', round(d['kernels'][0]['kernel_ms'], 2), 'left', d['kernels'][0].get('reads_left_compact_path'), 'stage2', round(s['stage2_wall'], 1), 'lat', round(s['latency_submit_to_done']))"
  done
done
