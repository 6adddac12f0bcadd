#!/bin/bash
mkdir -p gpurun_out
run() { label=$1; shift; env "$@" timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$label', round(d['value']), 'ms/step', round(d['ms_per_step'], 2), 'cpu', round(d['host_cpu']['process_cpu_s_per_wall_s'], 2), {k[:12]: v for k, v in list(d['host_cpu']['by_thread_name_cpu_s_per_wall_s'].items())[:4]})"; }
run base X=1
run nodirect AMD_DIRECT_DISPATCH=0
run activewait ROC_ACTIVE_WAIT_TIMEOUT=50
run mwaitx HSA_ENABLE_MWAITX=1
run batch DEBUG_CLR_MAX_BATCH_SIZE=1024
run base2 X=1
