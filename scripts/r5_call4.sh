#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_stream_gpu.py -m gpu -x -q -k "routed" > gpurun_out/c4_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c4_pytest.txt
tail -3 gpurun_out/c4_pytest.txt
bash scripts/ab_env5.sh 20 "idle-only||" "never|HP_WFA2_ROUTE=0|" > gpurun_out/c4_ab.txt 2>&1
cat gpurun_out/c4_ab.txt
for m in 1 0 1 0; do echo "HP_WFA2_ROUTE=$m"; HP_WFA2_ROUTE=$m timeout 200 tests/cpp/dispatch_test 64 60000 4165 8 2>&1 | tail -4; done > gpurun_out/c4_dispatch.txt 2>&1
cat gpurun_out/c4_dispatch.txt
