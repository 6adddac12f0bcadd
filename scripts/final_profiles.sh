#!/bin/bash
# scripts/final_profiles.sh <tag>: on the GPU box - kernel/copy trace + PMC passes of the default bench, the traffic file for THIS
# build, then the default bench line (with roofline.traffic), the phase breakdown and the step counters of the instrumented builds
# (scripts/build_variant.sh prof -DW3_PROF=1; stats -DW3_STATS=1), a stream trace and the per-block dispatch test. Afterwards, here:
# scripts/final_profiles.sh --collect <tag> copies the summaries into profiles/round5/.
cd "$(dirname "$0")/.."
if [ "$1" = "--collect" ]; then
  P=gpurun_out/prof_path_$2
  cp $P/trace/trace_kernel_stats.csv profiles/round5/path_kernel_stats.csv
  cp $P/trace/trace_memory_copy_stats.csv profiles/round5/path_memory_copy_stats.csv
  cp $P/pmc_summary.txt profiles/round5/path_pmc_summary.txt
  cp $P/traffic.json profiles/round5/traffic.json
  python scripts/overlap.py $P/trace > profiles/round5/path_overlap.txt
  tail -1 $P/bench.json > profiles/round5/path_bench_under_rocprof.json
  tail -1 gpurun_out/$2_bench_default.json > profiles/round5/bench_default.json
  python scripts/w3_prof_sum.py gpurun_out/$2_phases.txt > gpurun_out/$2_phases_sum.txt
  python scripts/w3_stats_sum.py gpurun_out/$2_stats.txt > gpurun_out/$2_stats_sum.txt
  exit 0
fi
T=$1
PMC=1 timeout 1500 bash scripts/prof_path.sh $T --steps 5 --warmup 2 > gpurun_out/${T}_prof.log 2>&1
cp gpurun_out/prof_path_$T/traffic.json profiles/round5/traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
HP_LIB=hiphase_amd/libhiphase_gpu_prof.so timeout 300 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 1 --warmup 1 --depth 1 > gpurun_out/${T}_phases.txt 2>&1
HP_LIB=hiphase_amd/libhiphase_gpu_stats.so timeout 300 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 1 --warmup 1 --depth 1 > gpurun_out/${T}_stats.txt 2>&1
HP_STREAM_TRACE=1 timeout 300 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 > /dev/null 2> gpurun_out/${T}_stream_trace.txt
timeout 300 tests/cpp/dispatch_test 64 60000 4165 8 > gpurun_out/${T}_dispatch.json 2>/dev/null
# the HiFi-shaped workload as the headline run (three times), and the uniform one three times more (spread)
for i in 1 2 3; do timeout 240 python bench.py --hifi --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${T}_hifi_headline.jsonl; timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${T}_uniform_headline.jsonl; done
# the class kernels with the device to themselves: one set at a time
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 8 --depth 1 2>/dev/null | tail -1 > gpurun_out/${T}_depth1.json
# side figures for DESIGN.md 4: reads staged by the library, two host threads, more edit noise, ASCII hand-over
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 --host-memory pageable 2>/dev/null | tail -1 > gpurun_out/${T}_pageable.json
HP_HOST_THREADS=2 timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 2>/dev/null | tail -1 > gpurun_out/${T}_ht2.json
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 10 --spec edit_noise=0.01 2>/dev/null | tail -1 > gpurun_out/${T}_noise1.json
# (2 % edit noise: measured on earlier builds of the round)
timeout 300 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 12 --seq-format ascii 2>/dev/null | tail -1 > gpurun_out/${T}_ascii.json
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 --depth 6 2>/dev/null | tail -1 > gpurun_out/${T}_depth6.json
