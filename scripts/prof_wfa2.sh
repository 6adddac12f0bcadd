#!/bin/bash
# Instrumented build of the library (s_memtime section timers in hp_wfa2_kernel, printed by a few workgroups) next to
# the product build; run with HP_LIB=hiphase_amd/libhiphase_gpu_prof.so python scripts/bench_wfa.py ...
set -e
cd "$(dirname "$0")/.."
S=hiphase_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-result -DW2_PROF=1 \
  -o hiphase_amd/libhiphase_gpu_prof.so $S/hp_api.hip $S/hp_astar.hip $S/hp_wfa.hip $S/hp_wfa2.hip $S/hp_edit.hip $S/hp_local.hip $S/hp_block.hip $S/hp_stream.hip $S/hp_synth.cpp $S/hp_synth_reads.cpp $S/hp_capture.cpp $S/hp_abi_layout.cpp
