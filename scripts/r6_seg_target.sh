#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_segt; mkdir -p $O
show() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); s = d['stage_ms']
print('$1', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'period', round(d.get('period_ms') or 0, 2), 'astar_kernel', round(s['astar_kernel'], 1), 'wfa', round(s['graph_wfa_kernels'], 1), 'stage4', round(s['stage4_wall'], 1), 'lat', round(s['latency_submit_to_done']))"; }
for rep in 1 2; do for t in 0 64 96; do
  if [ $t = 0 ]; then unset HP_SEG_TARGET; else export HP_SEG_TARGET=$t; fi
  python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 2>/dev/null | tail -1 | show "default seg_target=$t"
done; done 2>&1 | tee $O/default.txt
