#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do timeout 200 python -m pytest tests/test_stream_gpu.py -m gpu -x -q -k "device_readable" 2>&1 | tail -2; done > gpurun_out/c12_pytest.txt 2>&1
cat gpurun_out/c12_pytest.txt
