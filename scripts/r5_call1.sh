#!/bin/bash
# round 5, early-pass call: GPU suite, A/B (records routed past the compact kernels vs not), one traced run
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c1_pytest.txt
tail -3 gpurun_out/c1_pytest.txt
bash scripts/ab_env5.sh 20 "routed||" "off|HP_WFA2_SUSPECT_OPS=0|" > gpurun_out/c1_ab.txt 2>&1
cat gpurun_out/c1_ab.txt
HP_STREAM_TRACE=1 timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 > gpurun_out/c1_trace_bench.json 2> gpurun_out/c1_trace.txt
grep -c "early:" gpurun_out/c1_trace.txt; grep "early:" gpurun_out/c1_trace.txt | tail -5; grep "late:" gpurun_out/c1_trace.txt | tail -5; grep "streams created" gpurun_out/c1_trace.txt | tail -1
