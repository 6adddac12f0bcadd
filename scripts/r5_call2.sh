#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_stream_gpu.py -m gpu -x -q -k "routed or around_max or generated_sets" > gpurun_out/c2_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c2_pytest.txt
tail -3 gpurun_out/c2_pytest.txt
bash scripts/ab_env5.sh 20 "routed||" "off|HP_WFA2_SUSPECT_OPS=0|" "routed1|HP_EARLY_WORKERS=1|" > gpurun_out/c2_ab.txt 2>&1
cat gpurun_out/c2_ab.txt
HP_STREAM_TRACE=1 timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 > gpurun_out/c2_trace_bench.json 2> gpurun_out/c2_trace.txt
grep "early:" gpurun_out/c2_trace.txt | sed -e 's/.*routed past the compact kernels; their pass started/started/' -e 's/the reference-window test had settled/settled/' -e 's/the dense-band pass of the other/dense/' | tail -22; grep "late:" gpurun_out/c2_trace.txt | tail -3; grep "streams created" gpurun_out/c2_trace.txt | tail -1
