#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r6_ids; mkdir -p $O
python -m pytest tests/test_wfa_gpu.py -q -m gpu -x > $O/parity.log 2>&1; echo "parity rc=$?" | tee $O/rc.txt; tail -1 $O/parity.log
timeout 300 python scripts/wfa_stress.py 62 40 > $O/stress.log 2>&1; tail -1 $O/stress.log
for rep in 1 2 3; do AB_ARGS="--depth 1" scripts/ab_lib.sh 10 base noids; done 2>&1 | tee $O/ab_depth1.txt
for rep in 1 2; do scripts/ab_lib.sh 20 base noids; done 2>&1 | tee $O/ab_stream.txt
for rep in 1 2; do AB_ARGS="--spec edit_noise=0.01" scripts/ab_lib.sh 20 base noids; done 2>&1 | tee $O/ab_noise1.txt
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc -o pmc -- python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --depth 1 --steps 3 --warmup 2 > /dev/null 2> $O/pmc.err
python - <<'EOP'
import csv, glob, collections
tot = collections.Counter()
for f in glob.glob('gpurun_out/r6_ids/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'hp_wfa3_kernel' in r['Kernel_Name']: tot[r['Counter_Name']] += float(r['Counter_Value'])
print({k: round(v) for k, v in tot.items()}, 'sum', round(sum(tot.values())))
EOP
