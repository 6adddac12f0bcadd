#!/bin/bash
# scripts/sweep.sh <out-file> <ENV_NAME|--depth> <values...> : the default bench (no CPU leg) twice per value; one summary line per run
OUT=$1; KEY=$2; shift 2
for v in "$@"; do
  for r in 1 2; do
    if [ "$KEY" = "--depth" ]; then L=$(python bench.py --no-cpu --no-resident --steps 20 --warmup 6 --depth $v 2>/dev/null | tail -1)
    else L=$(env $KEY=$v python bench.py --no-cpu --no-resident --steps 20 --warmup 6 2>/dev/null | tail -1); fi
    echo "$L" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms']
print('$KEY', '$v', 'ms/step', round(d['ms_per_step'],1), 'latency', round(s['latency_submit_to_done'],1), 'wfa span', round(d['roofline']['kernel_ms'],1), 'walls', [round(s[k],1) for k in ('stage1_wall','stage2_wall','stage3_wall','stage4_wall')])" >> $OUT
  done
done
