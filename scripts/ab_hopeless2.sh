#!/bin/bash
for k in "$@"; do
  for en in 0.01 0.02; do
    HP_WFA2_HOPELESS=$k timeout 300 python bench.py --no-cpu --no-resident --no-drop-in --steps 8 --spec edit_noise=$en 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('K=$k noise $en', round(d['value']), 'ms/step', round(d['ms_per_step'], 1), 'span', round(d['stage_ms']['graph_wfa_kernels'], 1), 'left', d['kernels'][0].get('reads_left_compact_path'), (d.get('parity') or {}).get('bit_identical'))"
  done
done
