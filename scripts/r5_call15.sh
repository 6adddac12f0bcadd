#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_stream_gpu.py tests/test_wfa_gpu.py -m gpu -x -q -k "hifi or noisy or around_max or leftover or band" > gpurun_out/c15_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c15_pytest.txt
tail -3 gpurun_out/c15_pytest.txt
for i in 1 2 3; do HP_STREAM_TRACE=1 timeout 240 python bench.py --hifi --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 --warmup 5 2> gpurun_out/c15_hifi_trace_$i.txt | tail -1 > gpurun_out/c15_hifi_$i.json
python - <<PY
import json
d=json.loads(open('gpurun_out/c15_hifi_$i.json').read())
s=d['stage_ms']
print('hifi', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'period', round(d['period_ms'],2), 'first', round(d['first_completion_ms']), 'wfa', round(d['kernels'][0]['kernel_ms'],2), 'astar', round(d['kernels'][1]['kernel_ms'],2), 'walls', [round(s[k],1) for k in ('stage1_wall','stage2_wall','stage3_wall','stage4_wall')], 'lat', round(s['latency_submit_to_done']), 'intervals', d['completion_intervals_ms'])
PY
done
grep "late:" gpurun_out/c15_hifi_trace_3.txt | awk -F'dense-band pass after ' '{print $2+0}' | sort -n | tr '\n' ' '; echo
grep "device-wide" gpurun_out/c15_hifi_trace_3.txt
bash scripts/ab_env5.sh 20 "uniform||" 2>&1 | tail -4 | cut -c1-200
