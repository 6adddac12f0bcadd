#!/bin/bash
# bring-up helper: one tiny solve with the HIP runtime log enabled
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HP_DEBUG=1 HP_DEBUG_STAGE=${1:-1} AMD_LOG_LEVEL=4
timeout -s KILL 30 python -c "
import sys; sys.path.insert(0,'.')
from hiphase_amd import *
b,_=synth_block(1,5,2,0,0,6); r=astar_solver(0,b); print(r.haplotype_1, r.statistics.as_tuple())
" > gpurun_out/amdlog.txt 2>&1
echo "rc=$?"
grep -n "hp\]" gpurun_out/amdlog.txt | head
tail -n 60 gpurun_out/amdlog.txt | cut -c1-300
