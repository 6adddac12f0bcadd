#!/bin/bash
# the hand-over threshold (HP_WFA2_HOPELESS, per cent of max_edit_distance): rate + the late results' chain
for k in "$@"; do
  HP_WFA2_HOPELESS=$k HP_STREAM_TRACE=1 timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --steps 20 2> gpurun_out/hopeless_$k.txt | tail -1 | python -c "
import json, sys, re, statistics
d = json.loads(sys.stdin.read()); s = d['stage_ms']
L = [l for l in open('gpurun_out/hopeless_$k.txt') if l.startswith('[hp] late')][-16:]
fin = [float(re.search(r'dense-band pass after ([\d.]+)', l).group(1)) for l in L]
n1 = [int(re.search(r\"collection's (\d+) leftovers\", l).group(1)) for l in L]
n2 = [int(re.search(r'their (\d+) leftovers', l).group(1)) for l in L]
print('K=$k', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'lat', round(s['latency_submit_to_done']), 'walls', [round(s[x], 1) for x in ('stage1_wall', 'stage2_wall', 'stage3_wall', 'stage4_wall')], 'span', round(s['graph_wfa_kernels'], 1),
      'chain mean', round(statistics.mean(fin), 1), 'max', max(fin), 'first', round(statistics.mean(n1)), 'second', round(statistics.mean(n2)), 'parity', (d.get('parity') or {}).get('bit_identical'))"
done
