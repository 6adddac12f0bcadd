#!/bin/bash
bash scripts/prof_path.sh c3 --steps 8 --warmup 3 > gpurun_out/c3_prof.txt 2>&1
python scripts/early_timeline.py gpurun_out/prof_path_c3/trace > gpurun_out/c3_timeline.txt 2>&1
rm -rf gpurun_out/prof_path_c3/trace/*/*kernel_trace.csv gpurun_out/prof_path_c3/trace/*/*memory_copy_trace.csv gpurun_out/prof_path_c3/trace/*kernel_trace.csv gpurun_out/prof_path_c3/trace/*memory_copy_trace.csv 2>/dev/null
wc -l gpurun_out/c3_timeline.txt; tail -5 gpurun_out/c3_prof.txt | cut -c1-300
