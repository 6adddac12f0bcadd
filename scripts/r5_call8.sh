#!/bin/bash
mkdir -p gpurun_out
HP_BENCH_THREAD_DUMP=1 HP_STREAM_TRACE=1 timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 > gpurun_out/c8_bench.json 2> gpurun_out/c8_trace.txt
grep "\[bench\] thread" gpurun_out/c8_trace.txt | cut -c1-700
grep "^\[hp\] set" gpurun_out/c8_trace.txt | tail -22 | head -4 | cut -c1-600
grep "streams created" gpurun_out/c8_trace.txt | tail -1
