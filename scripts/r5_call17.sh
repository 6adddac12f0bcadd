#!/bin/bash
mkdir -p gpurun_out
timeout 700 python -m pytest tests/test_coalesce_gpu.py tests/test_blocks_gpu.py tests/test_stream_gpu.py -m gpu -x -q -k "not eight_pipelines" > gpurun_out/c17_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c17_pytest.txt
tail -3 gpurun_out/c17_pytest.txt
for m in 1 0 1 0; do echo "HP_STREAM_SMALL_ASYNC=$m"; HP_STREAM_SMALL_ASYNC=$m timeout 200 tests/cpp/dispatch_test 64 60000 4165 8 2>/dev/null | tail -1 | cut -c1-420; done > gpurun_out/c17_dispatch.txt 2>&1
cat gpurun_out/c17_dispatch.txt
