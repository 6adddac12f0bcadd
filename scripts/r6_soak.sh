#!/bin/bash
# soak on the shipped build: long streams (default, HiFi-shaped, deep60), the per-block entries with three pipelines, everything under
# timeouts; prints one line per run (value, parity) - a hang shows as a missing line + rc 124
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_soak; mkdir -p $O
show() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],2), (d.get('parity') or {}).get('bit_identical'), 'pruned', d.get('pruned_solutions'))
except Exception as e: print('$1 NO RESULT', repr(e))"; }
{
timeout 400 python bench.py --steps 400 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 1 2>/dev/null | tail -1 | show "default x400"; echo "rc=${PIPESTATUS[0]}"
timeout 400 python bench.py --hifi --steps 300 --seed 81 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 1 2>/dev/null | tail -1 | show "hifi x300"; echo "rc=${PIPESTATUS[0]}"
timeout 600 python bench.py --deep60 --coverage 60 --total-hets 20000 --steps 160 --seed 82 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 1 2>/dev/null | tail -1 | show "deep60 x160"; echo "rc=${PIPESTATUS[0]}"
HP_DEV_CACHE_POISON=1 timeout 400 python bench.py --steps 120 --seed 83 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 1 2>/dev/null | tail -1 | show "default x120 poisoned"; echo "rc=${PIPESTATUS[0]}"
HP_DEV_CACHE_POISON=1 timeout 600 python bench.py --deep60 --coverage 60 --total-hets 20000 --steps 60 --seed 84 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 1 2>/dev/null | tail -1 | show "deep60 x60 poisoned"; echo "rc=${PIPESTATUS[0]}"
for i in 1 2 3; do HP_QUEUE_WORKERS=3 timeout 200 tests/cpp/dispatch_test 64 60000 4165 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dispatch 3 pipelines', d['pool_hets_per_s'], d['async_hets_per_s'], 'mismatching', d['mismatching_blocks'], 'failed', d['failed_calls'])"; done
HP_DEV_CACHE_POISON=1 timeout 200 tests/cpp/dispatch_test 64 60000 4165 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dispatch poisoned', d['pool_hets_per_s'], d['async_hets_per_s'], 'mismatching', d['mismatching_blocks'], 'failed', d['failed_calls'])"
} 2>&1 | tee $O/soak.txt
