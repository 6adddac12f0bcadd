#!/bin/bash
# A/B of two builds of the library on the default bench: scripts/ab.sh <out-prefix> [bench args]; the alternative build is
# hiphase_amd/libhiphase_gpu_alt.so (HP_LIB). Alternates the two, three runs each.
P=$1; shift
for i in 1 2 3; do
  python bench.py --no-cpu "$@" >> ${P}_main.jsonl 2>/dev/null
  HP_LIB=hiphase_amd/libhiphase_gpu_alt.so python bench.py --no-cpu "$@" >> ${P}_alt.jsonl 2>/dev/null
done
