#!/bin/bash
# scripts/build_variant.sh <name> <hipcc flags...>: the library built with extra flags as hiphase_amd/libhiphase_gpu_<name>.so
set -e
cd "$(dirname "$0")/.."
N=$1; shift
S=hiphase_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-result "$@" \
  -o hiphase_amd/libhiphase_gpu_$N.so $S/hp_api.hip $S/hp_astar.hip $S/hp_wfa.hip $S/hp_wfa2.hip $S/hp_edit.hip $S/hp_local.hip $S/hp_block.hip $S/hp_stream.hip $S/hp_synth.cpp $S/hp_synth_reads.cpp $S/hp_capture.cpp $S/hp_abi_layout.cpp
