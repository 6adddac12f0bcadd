#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_solve; mkdir -p $O
show() { python -c "
import json, sys
d = json.loads(sys.stdin.read()); s = d['stage_ms']
print('$1', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'period', round(d.get('period_ms') or 0, 2), 'astar_kernel', round(s['astar_kernel'], 1), 'stage4', round(s['stage4_wall'], 1), 'lat', round(s['latency_submit_to_done']), 'cpu', round(d['host_cpu']['process_cpu_s_per_wall_s'], 2), 'parity', (d.get('parity') or {}).get('bit_identical'))"; }
for rep in 1 2; do for n in 2 4 7; do
  HP_STREAM_SOLVE_THREADS=$n python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 2>/dev/null | tail -1 | show "default threads=$n"
done; done 2>&1 | tee $O/default.txt
for n in 2 4 7; do
  HP_STREAM_SOLVE_THREADS=$n python bench.py --deep60 --coverage 60 --total-hets 20000 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 3 --steps 14 2>/dev/null | tail -1 | show "deep60 threads=$n"
done 2>&1 | tee $O/deep60.txt
for n in 2 4; do
  HP_STREAM_SOLVE_THREADS=$n python bench.py --hifi --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 2>/dev/null | tail -1 | show "hifi threads=$n"
done 2>&1 | tee $O/hifi.txt
