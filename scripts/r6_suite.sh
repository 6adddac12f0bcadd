#!/bin/bash
# whole GPU suite on the current build + smoke + a short bench line (is roofline.traffic there?)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_suite; mkdir -p $O
python -m pytest tests -q -m gpu > $O/suite.log 2>&1; echo "suite rc=$?" | tee $O/rc.txt; tail -4 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print(round(d['value']), d['roofline'])"
