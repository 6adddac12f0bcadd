"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/hiphase_gpu.h declares, validates its inputs on the host, and fails loudly without a GPU
(there is no CPU fallback on the product path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from hiphase_amd import _ffi, BlockMatrix, synth_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(hp_lib):
    hdr = open(os.path.join(ROOT, "include", "hiphase_gpu.h")).read()
    declared = set(re.findall(r"\b(hp_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"hp_batch"}
    assert declared == set(_ffi.EXPORTS), declared ^ set(_ffi.EXPORTS)
    for sym in declared:
        assert hasattr(hp_lib, sym), sym


def test_version_and_error_slot(hp_lib):
    assert b"gfx950" in hp_lib.hp_version()
    assert hp_lib.hp_last_error() is not None


def test_synth_matches_oracle_copy(hp_lib, oracle_lib):
    a, ta = synth_block(200, 30, 20, 0.01, 0.02, 42, dll=hp_lib)
    b, tb = synth_block(200, 30, 20, 0.01, 0.02, 42, dll=oracle_lib)
    for f in ("read_start", "read_end", "row_off", "quals", "var_flags"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    assert np.array_equal(a.alleles_2bit[: (a.n_cells + 3) // 4], b.alleles_2bit[: (b.n_cells + 3) // 4])
    assert np.array_equal(ta, tb)
    assert a.n_reads == 300  # R = ceil(N*C/S)


def _solve_rc(hp_lib, blk):
    p = _ffi.AstarParams(1000, 3, 40, 0)
    h1 = np.zeros(max(1, blk.n_variants), np.uint8)
    h2 = np.zeros(max(1, blk.n_variants), np.uint8)
    st = _ffi.PhaseStats()
    v = blk.view()
    return hp_lib.hp_astar_solve(C.byref(v), C.byref(p), h1.ctypes.data, h2.ctypes.data, C.byref(st))


def test_arg_validation_on_host(hp_lib):
    blk, _ = synth_block(20, 8, 6, 0.0, 0.0, 1, dll=hp_lib)
    bad = BlockMatrix(blk.n_variants, blk.read_start, blk.read_end.copy(), blk.row_off, blk.alleles_2bit, blk.quals,
                      blk.var_flags)
    bad.read_end[0] = bad.read_start[0] + 1000  # region beyond N / row_off mismatch
    assert _solve_rc(hp_lib, bad) == -4  # HP_ERR_ARG
    assert b"inconsistent" in hp_lib.hp_last_error()


def test_ignored_variant_invariant_on_host(hp_lib):
    """astar_phaser.rs:435-442: every row must be NoOverlap at an ignored variant."""
    blk, _ = synth_block(30, 8, 6, 0.0, 0.0, 3, dll=hp_lib)
    flags = blk.var_flags.copy()
    covered = int(blk.read_start[0])
    flags[covered] |= 1
    bad = BlockMatrix(blk.n_variants, blk.read_start, blk.read_end, blk.row_off, blk.alleles_2bit, blk.quals, flags)
    assert _solve_rc(hp_lib, bad) == -3  # HP_ERR_INVARIANT
    assert b"ignored variant" in hp_lib.hp_last_error()


def test_fails_loudly_without_gpu(hp_lib):
    if hp_lib.hp_device_count() > 0:
        pytest.skip("a GPU is visible")
    blk, _ = synth_block(20, 8, 6, 0.0, 0.0, 1, dll=hp_lib)
    assert _solve_rc(hp_lib, blk) == -1  # HP_ERR_HIP
    assert b"no CPU fallback" in hp_lib.hp_last_error()
