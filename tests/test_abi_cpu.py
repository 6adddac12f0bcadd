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


# ---- the other entry points: host-side validation runs before the device is touched, and without a GPU every one of
# ---- them fails loudly (there is no CPU fallback anywhere on the product path) ----------------------------------------
def _wfa_rc(hp_lib, specs, prune=500, max_ed=500):
    from hiphase_amd.wfa_graph import make_jobs
    jobs, keep = make_jobs(specs)
    out = (_ffi.WfaResult * len(specs))()
    als = [np.full(max(1, len(s.hets)), 3, np.uint8) for s in specs]
    ptrs = (C.c_void_p * len(specs))(*[a.ctypes.data for a in als])
    rc = hp_lib.hp_wfa_assign_batch(jobs, len(specs), prune, max_ed, out, ptrs, 0)
    _wfa_rc.last_status = [o.status for o in out]
    return rc, jobs


def test_wfa_host_validation_and_no_fallback(hp_lib):
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from wfa_util import synth_wfa_job
    spec = synth_wfa_job(5, ref_len=900, n_vars=6)[0]
    rc, _ = _wfa_rc(hp_lib, [spec], max_ed=70000)
    assert rc == 0 and _wfa_rc.last_status == [3]    # beyond the kernels' diagonal range: soft, per job (HP_WFA_UNSUPPORTED), no device needed to say so
    bad = synth_wfa_job(6, ref_len=900, n_vars=6)[0]
    bad.ref_start, bad.ref_end = 500, 100                                          # window turned inside out
    rc, _ = _wfa_rc(hp_lib, [bad])
    assert rc == -4 and b"reference window" in hp_lib.hp_last_error()            # HP_ERR_ARG
    if hp_lib.hp_device_count() == 0:
        rc, _ = _wfa_rc(hp_lib, [spec])                                            # valid job: the graph is built, then ...
        assert rc == -1 and b"no CPU fallback" in hp_lib.hp_last_error()         # HP_ERR_HIP


def test_edit_distance_and_local_batches_fail_loudly(hp_lib):
    a = np.frombuffer(b"ACGT", np.uint8)
    pr = (_ffi.EdPair * 1)()
    pr[0].a = a.ctypes.data_as(C.POINTER(C.c_uint8)); pr[0].b = pr[0].a; pr[0].a_len = pr[0].b_len = 4
    out = np.zeros(1, np.uint64)
    assert hp_lib.hp_edit_distance_batch(pr, 1, None, 0) == -4                      # null output
    if hp_lib.hp_device_count() == 0:
        assert hp_lib.hp_edit_distance_batch(pr, 1, out.ctypes.data_as(C.POINTER(C.c_uint64)), 0) == -1
        assert b"no CPU fallback" in hp_lib.hp_last_error()
    # a CIGAR with a Pad op is rejected on the host (rust-htslib's aligned_pairs panics on it, read_parsing.rs:165)
    rd = (_ffi.LocalRead * 1)()
    cg = np.array([(4 << 4) | 6], np.uint32)                                        # 4P
    sq = np.frombuffer(b"ACGT", np.uint8); ql = np.full(4, 30, np.uint8)
    rd[0].pos, rd[0].cigar, rd[0].n_cigar = 10, cg.ctypes.data_as(C.POINTER(C.c_uint32)), 1
    rd[0].seq_len, rd[0].seq, rd[0].qual = 4, sq.ctypes.data_as(C.POINTER(C.c_uint8)), ql.ctypes.data_as(C.POINTER(C.c_uint8))
    vs = (_ffi.LocalVariant * 1)()
    al0 = np.frombuffer(b"A", np.uint8); al1 = np.frombuffer(b"C", np.uint8)
    vs[0].position, vs[0].ref_len, vs[0].variant_type = 11, 1, 0
    vs[0].allele0, vs[0].allele1 = al0.ctypes.data_as(C.POINTER(C.c_uint8)), al1.ctypes.data_as(C.POINTER(C.c_uint8))
    vs[0].allele0_len = vs[0].allele1_len = 1
    alleles = np.zeros(1, np.uint8); quals = np.zeros(1, np.uint8)
    st = (_ffi.ReadStats * 1)()
    rc = hp_lib.hp_local_realign_batch(rd, 1, vs, 1, alleles.ctypes.data, quals.ctypes.data, st, 0)
    assert rc == -5, hp_lib.hp_last_error()                                         # HP_ERR_UNSUPPORTED


def test_abi_layout_matches_the_ctypes_bindings(hp_lib):
    """hp_abi_layout(): sizeof / offsetof of every struct of the C ABI as the library was compiled, diffed against the
    ctypes structs (the same diff a #[repr(C)] binding does once at start-up)."""
    import ctypes as C
    import json
    from hiphase_amd import _ffi
    layout = json.loads(hp_lib.hp_abi_layout().decode())
    pairs = {"hp_block_view": _ffi.BlockView, "hp_astar_params": _ffi.AstarParams, "hp_phase_stats": _ffi.PhaseStats,
             "hp_work_counters": _ffi.WorkCounters, "hp_wfa_variant": _ffi.WfaVariant, "hp_wfa_job": _ffi.WfaJob,
             "hp_wfa_result": _ffi.WfaResult, "hp_graph_node": _ffi.GraphNode, "hp_graph_job": _ffi.GraphJob,
             "hp_graph_result": _ffi.GraphResult, "hp_ed_pair": _ffi.EdPair, "hp_local_variant": _ffi.LocalVariant,
             "hp_local_read": _ffi.LocalRead, "hp_read_stats": _ffi.ReadStats, "hp_block_record": _ffi.BlockRecord,
             "hp_block_input": _ffi.BlockInput, "hp_block_params": _ffi.BlockParams, "hp_block_output": _ffi.BlockOutput,
             "hp_synth_spec": _ffi.SynthSpec, "hp_synth_reads_spec": _ffi.SynthReadsSpec}
    assert set(layout) == set(pairs)
    for name, cls in pairs.items():
        assert layout[name]["sizeof"] == C.sizeof(cls), name
        fields = layout[name]["fields"]
        assert list(fields) == [f for f, _ in cls._fields_], name
        for f, _ in cls._fields_:
            assert fields[f] == getattr(cls, f).offset, (name, f)


def test_block_tickets_are_checked_on_host(hp_lib):
    """hp_block_wait looks its ticket up in a table before anything else (round 5): a ticket that was never issued is HP_ERR_ARG
    with a message, whatever its value - the round-4 entry cast it to a pointer and freed it. No GPU needed for that."""
    import ctypes as C
    for t in (0, 1, 7, 0xDEADBEEF, 2 ** 64 - 1):
        assert hp_lib.hp_block_wait(C.c_uint64(t)) == -4
        assert b"ticket" in hp_lib.hp_last_error()
