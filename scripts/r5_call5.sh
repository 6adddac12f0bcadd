#!/bin/bash
mkdir -p gpurun_out
bash scripts/ab_env5.sh 20 "div64||" "div256|HP_WFA2_ESC_DIV=256|" "div1024|HP_WFA2_ESC_DIV=1024|" > gpurun_out/c5_ab.txt 2>&1
cat gpurun_out/c5_ab.txt
HP_DEBUG=1 timeout 200 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 3 --warmup 1 2>&1 | grep "wfa2: [0-9]* jobs" | head -3 | cut -c1-300
