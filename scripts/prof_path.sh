#!/bin/bash
# rocprofv3 kernel + memory-copy trace (+ optional PMC passes) of the default whole-path bench (streamed); summaries go to
# gpurun_out/prof_path_<tag>. The trace pass runs the pipeline as the bench does; the counter passes (PMC=1, one --pmc pass per
# counter group, nothing but counters in them) run it one set at a time (--depth 1): counter collection serialises kernels.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_path_$1; shift
mkdir -p $OUT
[ -n "$PMC_ONLY" ] || rocprofv3 --output-format csv --kernel-trace --memory-copy-trace --stats -d $OUT/trace -o trace -- python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe "$@" > $OUT/bench.json 2> $OUT/trace.err
[ -n "$PMC_ONLY" ] && python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe "$@" > $OUT/bench.json 2> $OUT/trace.err
if [ -n "$PMC" ]; then
  rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc1 -o pmc1 -- python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --depth 1 "$@" > /dev/null 2> $OUT/pmc1.err
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --depth 1 "$@" > /dev/null 2> $OUT/pmc3.err
  rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- python bench.py --no-cpu --no-resident --no-drop-in --no-hifi --no-deep60 --no-pcie-probe --depth 1 "$@" > /dev/null 2> $OUT/pmc4.err
fi
tail -c 1200 $OUT/bench.json; echo
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cat $f; done
if [ -n "$PMC" ]; then python scripts/make_traffic_json.py $OUT; fi
