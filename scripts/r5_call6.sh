#!/bin/bash
mkdir -p gpurun_out
export HP_WFA2_ESC_DIV=1024
bash scripts/ab_env5.sh 20 "res8||" "res4|HP_STREAM_RESERVE_PCT=4|" "res12|HP_STREAM_RESERVE_PCT=12|" "res8d6||--depth=6" > gpurun_out/c6_ab.txt 2>&1
cat gpurun_out/c6_ab.txt
