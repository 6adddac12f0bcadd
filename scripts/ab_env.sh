#!/bin/bash
# A/B of environment switches on the default bench: ab_env.sh <steps> "VAR=a" "VAR=b VAR2=c" ... ("-" = nothing set;
# BENCH_ARGS=--depth=7 adds bench.py arguments, no spaces inside)
STEPS=$1; shift
for e in "$@"; do
  if [ "$e" = "-" ]; then E=""; else E="$e"; fi
  env $E STEPS=$STEPS bash -c 'timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --steps $STEPS $BENCH_ARGS 2>/dev/null' | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
s = d['stage_ms']
print('[$e]', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'kernels', [(k['kernel'][:26], round(k['kernel_ms'], 2), k.get('reads_left_compact_path')) for k in d['kernels']],
      'walls', [round(s[k], 1) for k in ('stage1_wall', 'stage2_wall', 'stage3_wall', 'stage4_wall')], 'lat', round(s['latency_submit_to_done']), 'cpu', round(d['host_cpu']['process_cpu_s_per_wall_s'], 2), 'parity', (d.get('parity') or {}).get('bit_identical'))"
done
