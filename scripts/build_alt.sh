#!/bin/bash
# scripts/build_alt.sh [sed-expression ...]: a copy of the sources with the given sed edits applied, built as
# hiphase_amd/libhiphase_gpu_alt.so (the second library of scripts/ab.sh). Extra hipcc flags: ALT_FLAGS.
set -e
cd "$(dirname "$0")/.."
T=/tmp/alt_tree; rm -rf $T; mkdir -p $T/hiphase_amd $T/include
cp -r hiphase_amd/csrc $T/hiphase_amd/; cp include/*.h include/*.hpp $T/include/ 2>/dev/null || true
for e in "$@"; do sed -i "$e" $T/hiphase_amd/csrc/*.hip $T/hiphase_amd/csrc/*.h; done
S=$T/hiphase_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-result $ALT_FLAGS \
  -o hiphase_amd/libhiphase_gpu_alt.so $S/hp_api.hip $S/hp_astar.hip $S/hp_wfa.hip $S/hp_wfa2.hip $S/hp_edit.hip $S/hp_local.hip $S/hp_block.hip $S/hp_stream.hip $S/hp_synth.cpp $S/hp_synth_reads.cpp $S/hp_capture.cpp $S/hp_abi_layout.cpp
