#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_cring; mkdir -p $O
python -m pytest tests/test_astar_gpu.py tests/test_e2e_gpu.py -q -m gpu -x > $O/parity.log 2>&1; echo "astar parity rc=$?" | tee $O/rc.txt; tail -1 $O/parity.log
timeout 400 python scripts/long_stress.py > $O/long_stress.log 2>&1; tail -2 $O/long_stress.log
for rep in 1 2 3; do
  for v in 1 0; do
    HP_SEG_CRING=$v python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); s = d['stage_ms']
print('cring=$v', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'astar_kernel', round(s['astar_kernel'], 2), 'astar_solve', round(s['astar_solve'], 2), 'stage4', round(s['stage4_wall'], 1), 'lat', round(s['latency_submit_to_done']))"
  done
done 2>&1 | tee $O/ab.txt
for v in 1 0; do
  HP_SEG_CRING=$v python bench.py --workload c2 --no-cpu --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('c2 cring=$v', round(d['value']), 'ms/step', round(d['ms_per_step'], 2))"
done 2>&1 | tee -a $O/ab.txt
