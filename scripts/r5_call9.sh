#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stream_gpu.py tests/test_coalesce_gpu.py -m gpu -x -q > gpurun_out/c9_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c9_pytest.txt
tail -4 gpurun_out/c9_pytest.txt
bash scripts/ab_env5.sh 20 "split||" "whole|HP_STREAM_SPLIT=0|" > gpurun_out/c9_ab.txt 2>&1
cat gpurun_out/c9_ab.txt
HP_STREAM_TRACE=1 timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 > gpurun_out/c9_bench.json 2> gpurun_out/c9_trace.txt
grep "^\[hp\] set" gpurun_out/c9_trace.txt | tail -23 | head -5 | cut -c1-420
