#!/bin/bash
# scripts/final_profiles_r6.sh <tag>: on the GPU box - kernel/copy trace + PMC passes of the default bench, the traffic file for THIS build,
# then the driver-style bench line (with roofline.traffic and the hifi_mix / deep60 / drop_in legs), a stream trace, the per-block dispatch
# test and a few side figures. Afterwards, here: scripts/final_profiles_r6.sh --collect <tag> copies the summaries into profiles/round6/.
cd "$(dirname "$0")/.."
R=profiles/round6
if [ "$1" = "--collect" ]; then
  P=gpurun_out/prof_path_$2
  mkdir -p $R
  cp $P/trace/trace_kernel_stats.csv $R/path_kernel_stats.csv
  cp $P/trace/trace_memory_copy_stats.csv $R/path_memory_copy_stats.csv
  cp $P/pmc_summary.txt $R/path_pmc_summary.txt
  cp $P/traffic.json $R/traffic.json
  python scripts/overlap.py $P/trace > $R/path_overlap.txt
  tail -1 $P/bench.json > $R/path_bench_under_rocprof.json
  tail -1 gpurun_out/$2_bench_default.json > $R/bench_default.json
  tail -1 gpurun_out/$2_dispatch.json > $R/dispatch_test.json
  tail -40 gpurun_out/$2_stream_trace.txt > $R/stream_trace.txt
  for f in depth1 noise1 deep60_headline; do [ -s gpurun_out/$2_$f.json ] && tail -1 gpurun_out/$2_$f.json > $R/side_$f.json; done
  exit 0
fi
T=$1
mkdir -p $R
PMC=1 timeout 1500 bash scripts/prof_path.sh $T --steps 5 --warmup 2 > gpurun_out/${T}_prof.log 2>&1
cp gpurun_out/prof_path_$T/traffic.json $R/traffic.json
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
HP_STREAM_TRACE=1 timeout 300 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 > /dev/null 2> gpurun_out/${T}_stream_trace.txt
timeout 300 tests/cpp/dispatch_test 64 60000 4165 8 > gpurun_out/${T}_dispatch.json 2>/dev/null
for i in 1 2 3; do timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${T}_uniform_headline.jsonl; done
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 8 --depth 1 2>/dev/null | tail -1 > gpurun_out/${T}_depth1.json
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 10 --spec edit_noise=0.01 2>/dev/null | tail -1 > gpurun_out/${T}_noise1.json
timeout 600 python bench.py --deep60 --coverage 60 --total-hets 20000 --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --steps 40 2>/dev/null | tail -1 > gpurun_out/${T}_deep60_headline.json
python - <<EOP
import json
d = json.loads(open("gpurun_out/${T}_bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "period_ms", "wait_mode")}, "roofline", {k: d["roofline"].get(k) for k in ("frac", "traffic", "kernel_ms")})
print("deep60", {k: (d.get("deep60") or {}).get(k) for k in ("value", "ms_per_step", "period_ms", "pruned_solutions", "parity")})
print("hifi", (d.get("hifi_mix") or {}).get("value"), "drop_in", {k: (d.get("drop_in") or {}).get(k) for k in ("async_hets_per_s", "blocking_hets_per_s", "one_call_hets_per_s")})
print("parity", d.get("parity"))
EOP
