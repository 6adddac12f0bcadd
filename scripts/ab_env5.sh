#!/bin/bash
# round 5: environment / depth A/B on the default bench, three runs a side, interleaved. usage: ab_env5.sh <steps> "<label>|<env assignments>|<bench args>" ...
STEPS=$1; shift
TMP=$(mktemp)
for rep in 1 2 3; do
  for spec in "$@"; do
    IFS='|' read -r label envs bargs <<< "$spec"
    env $envs timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps $STEPS $bargs 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
s = d['stage_ms']
print('$label', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'period', round(d.get('period_ms') or 0, 2), 'first', round(d.get('first_completion_ms') or 0), 'wfa', round(d['kernels'][0]['kernel_ms'], 2), 'astar', round(d['kernels'][1]['kernel_ms'], 2),
      'walls', [round(s[k], 1) for k in ('stage1_wall', 'stage2_wall', 'stage3_wall', 'stage4_wall')], 'lat', round(s['latency_submit_to_done']), 'wait', round(s['waiting_between_stages']), 'cpu', round(d['host_cpu']['process_cpu_s_per_wall_s'], 2))" | tee -a $TMP
  done
done
# the spread per label (three runs a side: differences inside it are noise)
python - $TMP <<'PY'
import sys
rows = {}
for l in open(sys.argv[1]):
    f = l.split()
    if len(f) > 3 and f[2] == 'ms/step':
        rows.setdefault(f[0], []).append((float(f[1]), float(f[3])))
for k, v in rows.items():
    print(f"spread {k}: {min(a for a, _ in v) / 1e6:.2f}-{max(a for a, _ in v) / 1e6:.2f} M hets/s, {min(b for _, b in v):.2f}-{max(b for _, b in v):.2f} ms per step over {len(v)} runs")
PY
rm -f $TMP
