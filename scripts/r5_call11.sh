#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wfa_gpu.py tests/test_stream_gpu.py -m gpu -x -q -k "not eight_pipelines and not routed" > gpurun_out/c11_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c11_pytest.txt
tail -3 gpurun_out/c11_pytest.txt
for r in 1 2 3; do bash scripts/ab_lib.sh 20 base nospec; done > gpurun_out/c11_ab.txt 2>&1
cut -c1-330 gpurun_out/c11_ab.txt
AB_ARGS=--depth=1 bash scripts/ab_lib.sh 6 base nospec base nospec > gpurun_out/c11_ab_d1.txt 2>&1
cut -c1-330 gpurun_out/c11_ab_d1.txt
