#!/bin/bash
# deep60 headline, A/B over an environment switch: scripts/r6_deep60_ab.sh TAG [ENV=VAL ...]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
T=$1; shift
for kv in "$@"; do export "$kv"; done
timeout 600 python bench.py --deep60 --coverage 60 --total-hets 20000 --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 40 $DEEP60_EXTRA 2>/dev/null | tail -1 > gpurun_out/${T}.json
python - <<EOP
import json
d = json.loads(open("gpurun_out/${T}.json").read())
s = d["stage_ms"]
print("${T}", {k: d.get(k) for k in ("value", "ms_per_step", "period_ms")}, {k: round(s[k], 1) for k in ("astar_kernel", "stage4_wall", "stage3_wall", "graph_wfa_kernels", "latency_submit_to_done")}, "parity", d.get("parity"))
EOP
