#!/bin/bash
mkdir -p gpurun_out
bash scripts/ab_env5.sh 20 "split110||" "split100|HP_WFA2_SPLIT=100|" "split125|HP_WFA2_SPLIT=125|" > gpurun_out/c13_ab.txt 2>&1
cut -c1-260 gpurun_out/c13_ab.txt
