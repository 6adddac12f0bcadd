#!/bin/bash
# A* kernels after a change: parity tests, the main search by phase (HP_MAIN_PROF build), one C2 block with and without segments,
# a throughput batch, and the default stream's A* stage
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
timeout 900 python -m pytest tests/test_astar_gpu.py -x -q 2>&1 | tail -2
[ -f build/variants/libhp_mainprof.so ] && HP_LIB=build/variants/libhp_mainprof.so timeout 300 python scripts/r6_mainprof.py 2>&1 | tail -4 | cut -c1-330
for i in 1 2; do timeout 100 python scripts/r6_astar_one.py 5000 30 0.01; HP_NO_SEGMENTS=1 timeout 100 python scripts/r6_astar_one.py 5000 30 0.01; done
for i in 1 2; do python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); s = d['stage_ms']; print('default', round(d['value']), 'astar_kernel', round(s['astar_kernel'], 2), 'stage4', round(s['stage4_wall'], 2))"; done
timeout 300 python bench.py --no-cpu --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 5 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('resident', d.get('resident'))" | cut -c1-400
