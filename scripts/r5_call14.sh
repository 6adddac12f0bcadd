#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do HP_STREAM_TRACE=1 timeout 240 python bench.py --hifi --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 --warmup 5 2> gpurun_out/c14_hifi_trace_$i.txt | tail -1 > gpurun_out/c14_hifi_$i.json
python - <<PY
import json
d=json.loads(open('gpurun_out/c14_hifi_$i.json').read())
s=d['stage_ms']
print('hifi', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'period', round(d['period_ms'],2), 'first', round(d['first_completion_ms']), 'wfa', round(d['kernels'][0]['kernel_ms'],2), 'astar', round(d['kernels'][1]['kernel_ms'],2), 'walls', [round(s[k],1) for k in ('stage1_wall','stage2_wall','stage3_wall','stage4_wall')], 'lat', round(s['latency_submit_to_done']), 'intervals', d['completion_intervals_ms'], 'left', d['kernels'][0].get('reads_left_compact_path'))
PY
done
grep "^\[hp\] set" gpurun_out/c14_hifi_trace_2.txt | tail -8 | cut -c1-360
grep "late:" gpurun_out/c14_hifi_trace_2.txt | tail -4 | cut -c1-300
