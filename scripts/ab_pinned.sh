#!/bin/bash
# A/B of the reads' hand-over: ordinary host memory (staged) vs hp_host_alloc memory (read in place). usage: ab_pinned.sh <tag>
T=${1:-ab}
O=gpurun_out
timeout 600 python -m pytest tests/test_stream_gpu.py -k "device_readable" -x -q > $O/${T}_test.log 2>&1; tail -3 $O/${T}_test.log
for rep in 1 2; do
  for hm in pageable pinned; do
    timeout 600 python bench.py --host-memory $hm --no-cpu --no-resident --no-drop-in --steps 24 2> $O/${T}_${hm}_$rep.err | tail -1 > $O/${T}_${hm}_$rep.json
    python - $O/${T}_${hm}_$rep.json $hm <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "stages", d.get("stage_ms"), "cpu", d.get("host_cpu"), "parity", d.get("parity"))
PY
  done
done
HP_STREAM_TRACE=1 timeout 600 python bench.py --host-memory pinned --no-cpu --no-resident --no-drop-in --steps 12 2>&1 | grep "^\[hp\] set" | tail -8
