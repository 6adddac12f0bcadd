#!/bin/bash
# scripts/final_profiles_lite.sh <tag>: the end-of-round measurements that fit a quarter of an hour of GPU time - the full GPU suite, kernel /
# copy trace + PMC passes of the default bench with the traffic file for THIS build, the default bench line, a stream trace, the per-block
# dispatch test, the HiFi-shaped and the 1 %-noise headline runs, one set at a time, two more seeds with parity, one stress seed.
# Afterwards, here: scripts/final_profiles.sh --collect <tag> copies the summaries into profiles/round5/.
cd "$(dirname "$0")/.."
T=$1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/${T}_pytest.txt
tail -3 gpurun_out/${T}_pytest.txt
PMC=1 timeout 900 bash scripts/prof_path.sh $T --steps 5 --warmup 2 > gpurun_out/${T}_prof.log 2>&1
cp gpurun_out/prof_path_$T/traffic.json profiles/round5/traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
tail -1 gpurun_out/${T}_bench_default.json | cut -c1-400
HP_STREAM_TRACE=1 timeout 300 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 > /dev/null 2> gpurun_out/${T}_stream_trace.txt
timeout 300 tests/cpp/dispatch_test 64 60000 4165 8 > gpurun_out/${T}_dispatch.json 2>/dev/null
for i in 1 2; do timeout 240 python bench.py --hifi --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/${T}_hifi_headline.jsonl; done
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 8 --depth 1 2>/dev/null | tail -1 > gpurun_out/${T}_depth1.json
timeout 240 python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-pcie-probe --steps 10 --spec edit_noise=0.01 2>/dev/null | tail -1 > gpurun_out/${T}_noise1.json
for sd in 51 52; do timeout 200 python bench.py --seed $sd --steps 8 --no-resident --no-drop-in --no-hifi --cpu-seconds 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seed $sd', round(d['value']), round(d['ms_per_step'],2), d['parity'])"; done > gpurun_out/${T}_seeds.txt 2>&1
cat gpurun_out/${T}_seeds.txt
timeout 200 python scripts/wfa_stress.py 31 60 2>&1 | tail -2 > gpurun_out/${T}_stress.txt; cat gpurun_out/${T}_stress.txt
find gpurun_out/prof_path_$T -name "*kernel_trace.csv" -delete; find gpurun_out/prof_path_$T -name "*memory_copy_trace.csv" -size +20M -delete
