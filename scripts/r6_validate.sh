#!/bin/bash
# scripts/r6_validate.sh: whole GPU suite, the same under the device cache's poison switch, wfa_stress under it, default bench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_validate; mkdir -p $O
python -m pytest tests -q -m gpu > $O/suite.log 2>&1; echo "suite rc=$?" | tee $O/rc.txt
HP_DEV_CACHE_POISON=1 python -m pytest tests -q -m gpu > $O/suite_poison.log 2>&1; echo "poison suite rc=$?" | tee -a $O/rc.txt
if [ -z "$SKIP_EXTRA" ]; then
HP_DEV_CACHE_POISON=1 timeout 600 python scripts/wfa_stress.py 3 90 > $O/wfa_stress_poison.log 2>&1; echo "poison wfa_stress rc=$?" | tee -a $O/rc.txt
AMD_LOG_LEVEL=1 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
fi
tail -4 $O/suite.log; tail -4 $O/suite_poison.log; tail -2 $O/wfa_stress_poison.log; grep -c "failed to set" $O/bench.err; python - <<'EOP'
import json
d=json.loads(open('gpurun_out/r6_validate/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','period_ms','wait_mode')}, d.get('roofline',{}).get('frac'), d.get('parity'))
EOP
