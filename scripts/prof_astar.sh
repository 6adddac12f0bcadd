#!/bin/bash
# profile helper: kernel-trace stats + PMC counters for the A* kernel (separate passes, see MI355X guide)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1; shift
mkdir -p $OUT
BENCH="python bench.py --no-cpu $@"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/bench_pmc1.json 2> $OUT/pmc1.err
rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc2 -- $BENCH > $OUT/bench_pmc2.json 2> $OUT/pmc2.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- $BENCH > $OUT/bench_pmc3.json 2> $OUT/pmc3.err
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- $BENCH > $OUT/bench_pmc4.json 2> $OUT/pmc4.err
find $OUT -name "*.csv" | head -30
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -5 $f; done
for f in $(find $OUT -name "*counter_collection.csv"); do echo "== $f"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(float); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    if 'astar' in row.get('Kernel_Name',''):
        agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
for k in agg: print(f"{k:28s} total={agg[k]:.4g} dispatches={n[k]}")
PY
done
