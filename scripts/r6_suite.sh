#!/bin/bash
# whole GPU suite on the current build + the ids variant's WFA parity (for the diary) + smoke
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_suite; mkdir -p $O
python -m pytest tests -q -m gpu > $O/suite.log 2>&1; echo "suite rc=$?" | tee $O/rc.txt; tail -4 $O/suite.log
HP_LIB=$PWD/hiphase_amd/libhiphase_gpu_ids.so python -m pytest tests/test_wfa_gpu.py -q -m gpu > $O/ids_parity.log 2>&1; echo "ids variant parity rc=$?" | tee -a $O/rc.txt; tail -1 $O/ids_parity.log
HP_LIB=$PWD/hiphase_amd/libhiphase_gpu_ids.so timeout 200 python scripts/wfa_stress.py 63 60 > $O/ids_stress.log 2>&1; tail -1 $O/ids_stress.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
