#!/bin/bash
# scripts/final_profiles.sh <tag>: on the GPU box - kernel/copy trace + PMC passes of the default bench, the traffic file for THIS
# build, then the default bench line (with roofline.traffic) and the phase breakdown of the instrumented build. Afterwards, here:
# scripts/final_profiles.sh --collect <tag> copies the summaries into profiles/round3/.
cd "$(dirname "$0")/.."
if [ "$1" = "--collect" ]; then
  P=gpurun_out/prof_path_$2
  cp $P/trace/trace_kernel_stats.csv profiles/round3/path_kernel_stats.csv
  cp $P/trace/trace_memory_copy_stats.csv profiles/round3/path_memory_copy_stats.csv
  cp $P/pmc_summary.txt profiles/round3/path_pmc_summary.txt
  cp $P/traffic.json profiles/round3/traffic.json
  python scripts/overlap.py $P/trace > profiles/round3/path_overlap.txt
  tail -1 $P/bench.json > profiles/round3/path_bench_under_rocprof.json
  tail -1 gpurun_out/$2_bench_default.json > profiles/round3/bench_default.json
  python scripts/prof_sum.py gpurun_out/$2_phases.txt > gpurun_out/$2_phases_sum.txt
  exit 0
fi
T=$1
PMC=1 timeout 1500 bash scripts/prof_path.sh $T --steps 5 --warmup 2 > gpurun_out/${T}_prof.log 2>&1
cp gpurun_out/prof_path_$T/traffic.json profiles/round3/traffic.json
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
HP_LIB=hiphase_amd/libhiphase_gpu_prof.so timeout 300 python bench.py --no-cpu --no-resident --steps 1 --warmup 1 --depth 1 > gpurun_out/${T}_phases.txt 2>&1
