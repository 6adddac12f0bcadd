"""hiphase_amd — MI355X-native phasing core behind HiPhase's per-block API.

Host-side mirror (Python, ctypes over the C ABI of ``libhiphase_gpu.so``) of the reference's
hot-path interface:

* ``astar_phaser.astar_solver``        <- reference src/astar_phaser.rs:426-633
* ``read_segments.ReadSegment``        <- reference src/data_types/read_segments.rs:19-207
* ``wfa_graph.wfa_assign_batch``       <- reference src/wfa_graph.rs:119-284,350-650 + src/read_parsing.rs:790-800
* ``sequence_alignment.edit_distance`` <- reference src/sequence_alignment.rs:7-38

There is no CPU fallback: every compute entry point raises if the HIP library is missing.
"""
from ._ffi import (  # noqa: F401
    HpError,
    lib,
    BlockView,
    AstarParams,
    PhaseStats,
    WorkCounters,
    SynthSpec,
)
from .read_segments import AlleleType, ReadSegment, BlockMatrix, synth_block  # noqa: F401
from .astar_phaser import AstarResult, astar_solver, astar_solve_batch, ResidentBatch  # noqa: F401

__version__ = "0.1.0"
