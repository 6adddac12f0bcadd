#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_wfa_$1; shift
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- python scripts/bench_wfa.py "$@" > $OUT/bench.json 2> $OUT/trace.err
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc1 -o pmc1 -- python scripts/bench_wfa.py "$@" > /dev/null 2> $OUT/pmc1.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python scripts/bench_wfa.py "$@" > /dev/null 2> $OUT/pmc3.err
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- python scripts/bench_wfa.py "$@" > /dev/null 2> $OUT/pmc4.err
cat $OUT/bench.json
for f in $(find $OUT -name "*kernel_stats.csv"); do head -4 $f; done
for f in $(find $OUT -name "*counter_collection.csv"); do python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    if 'wfa' in row.get('Kernel_Name',''):
        agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
for k in sorted(agg): print(f"{k:28s} total={agg[k]:.6g} dispatches={n[k]}")
PY
done
