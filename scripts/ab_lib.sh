#!/bin/bash
# A/B of library builds on the default bench: ab_lib.sh <steps> <variant|base> ... (variant = hiphase_amd/libhiphase_gpu_<variant>.so)
STEPS=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset HP_LIB; else export HP_LIB=hiphase_amd/libhiphase_gpu_$v.so; fi
  timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps $STEPS $AB_ARGS 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
s = d['stage_ms']
print('$v', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'period', round(d.get('period_ms') or 0, 2), 'kernels', [(k['kernel'][:26], round(k['kernel_ms'], 2), k.get('reads_left_compact_path')) for k in d['kernels']],
      'walls', [round(s[k], 1) for k in ('stage1_wall', 'stage2_wall', 'stage3_wall', 'stage4_wall')], 'lat', round(s['latency_submit_to_done']), 'cpu', round(d['host_cpu']['process_cpu_s_per_wall_s'], 2), 'parity', (d.get('parity') or {}).get('bit_identical'))"
done
