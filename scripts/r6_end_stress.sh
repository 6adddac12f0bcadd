#!/bin/bash
# end-of-round stress on the last build (more whole sets against the oracle, deep60 seeds, WFA / A* random stress) + the host side at
# several times a pipeline's set rate: eight pipelines on the one GPU over sets with the default's record count and a fifth of its bases
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_end; mkdir -p $O
show() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],2), d.get('parity'), 'pruned', d.get('pruned_solutions'))"; }
for sd in 61 62; do timeout 300 python bench.py --seed $sd --steps 8 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 2 2>/dev/null | tail -1 | show "seed $sd"; done 2>&1 | tee $O/sets.txt
timeout 300 python bench.py --seed 63 --steps 6 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 2 --spec edit_noise=0.01 2>/dev/null | tail -1 | show "seed 63 noise 1%" | tee -a $O/sets.txt
timeout 300 python bench.py --seed 64 --steps 6 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 2 --seq-format ascii 2>/dev/null | tail -1 | show "seed 64 ascii" | tee -a $O/sets.txt
for sd in 65 66; do timeout 400 python bench.py --deep60 --coverage 60 --total-hets 12000 --seed $sd --steps 6 --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --cpu-seconds 2 2>/dev/null | tail -1 | show "deep60 seed $sd"; done 2>&1 | tee -a $O/sets.txt
timeout 200 python scripts/wfa_stress.py 67 90 2>&1 | tail -1 | tee -a $O/sets.txt
timeout 200 python scripts/wfa_stress.py 68 90 2>&1 | tail -1 | tee -a $O/sets.txt
timeout 400 python scripts/long_stress.py 7 10 2>&1 | tail -1 | tee -a $O/sets.txt
# host side: same records per set, a fifth of the bases (3-kb reads at 6x), eight pipelines in one process on the one GPU
for n in 1 4 8; do
  HP_STREAM_DEVICES=$n timeout 400 python bench.py --inproc --depth 3 --steps $((40 * n)) --coverage 6 --spec read_mean=3000 --spec read_sd=300 --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['host_cpu']
print('pipelines $n', 'sets/s', round(1e3/d['ms_per_step'],1), 'hets/s', round(d['value']), 'records/set', d['config']['records'], 'bases/set', d['config']['read_bases'], 'host cpu-s/s', round(h['process_cpu_s_per_wall_s'],2), 'throttled ms', h.get('cgroup_throttled_ms'))"
done 2>&1 | tee $O/host_scaling.txt
