#!/bin/bash
# A* after the round-6 changes to the main search's heap, the node-pool sizing and the segment take-over: random stress against the
# oracle (several seeds, also with tiny first pools and short warm-ups so that retries and take-overs happen everywhere), deep60 sets
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_astar_stress; mkdir -p $O
{
timeout 400 python scripts/long_stress.py 11 10 2>&1 | tail -1
timeout 400 python scripts/long_stress.py 12 8 40 2>&1 | tail -1
HP_ASTAR_CAP0=2 timeout 400 python scripts/long_stress.py 13 8 2>&1 | tail -1 | sed 's/^/cap0=2 /'
HP_SEG_WARM=16 HP_SEG_WARM2=64 timeout 400 python scripts/long_stress.py 14 8 2>&1 | tail -1 | sed 's/^/warm 16,64 /'
HP_SEG_WARM=16 HP_SEG_WARM2=64 HP_ASTAR_CAP0=3 timeout 400 python scripts/long_stress.py 15 6 2>&1 | tail -1 | sed 's/^/warm 16,64 cap0=3 /'
for sd in 71 72; do timeout 400 python bench.py --deep60 --coverage 60 --total-hets 12000 --seed $sd --steps 6 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deep60 seed $sd', round(d['value']), round(d['ms_per_step'],2), d.get('parity'), 'pruned', d.get('pruned_solutions'))"; done
} 2>&1 | tee $O/stress.txt
