#!/bin/bash
mkdir -p gpurun_out
bash scripts/ab_env5.sh 20 "never||" "always|HP_WFA2_ROUTE=2|" "always12|HP_WFA2_ROUTE=2 HP_STREAM_RESERVE_PCT=12|" > gpurun_out/c7_ab.txt 2>&1
cat gpurun_out/c7_ab.txt
