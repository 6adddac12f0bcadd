"""The parity plumbing itself (no GPU): hp_block_output_equal - product code, compiled into libhiphase_gpu.so, behind every
whole-path parity claim of the suite, of bench.py and of tests/cpp/dispatch_test - must say NO when any one field class of an
output is perturbed, and output buffers that nobody wrote must not compare equal. A pure-Python comparator (tests/e2e_util.py
output_diff) is held to it on the same cases."""
import ctypes as C

import pytest

from hiphase_amd import _ffi
from hiphase_amd.blocks import _params
from hiphase_amd.synth_sets import SynthSet, default_spec
from e2e_util import output_diff, outputs_diff
from oracle_ffi import oracle

KW = dict(total_hets=300, max_block_hets=90, seed=19, noisy_fraction=0.03, supplementary_fraction=0.06, frac_snv=0.70, frac_indel=0.15, frac_sv=0.05)


@pytest.fixture(scope="module")
def solved():
    d = oracle()
    prod = _ffi.lib()   # (host code of the product library: loads without a GPU)
    s = SynthSet(default_spec(d, seq_format=_ffi.SEQ_ASCII, **KW), d)
    prm = _params(2, 1000, 3, None, True)
    a, b = s.outputs().poison(0x77), s.outputs().poison(0xEE)
    for o in (a, b):
        for k in range(s.n):
            assert d.hpo_solve_block(C.byref(s.inputs[k]), C.byref(prm), C.byref(o.arr[k])) == 0
    return prod, s, a, b


def product_equal(prod, s, a, b, k):
    return bool(prod.hp_block_output_equal(C.byref(s.inputs[k]), C.byref(a.arr[k]), C.byref(b.arr[k])))


def test_two_solves_into_differently_poisoned_buffers_are_equal(solved):
    prod, s, a, b = solved
    assert all(product_equal(prod, s, a, b, k) for k in range(s.n))
    assert outputs_diff(s, a, b) == []
    assert sum(a.arr[k].n_segments for k in range(s.n)) > 50 and sum(a.arr[k].n_edit_distances for k in range(s.n)) > 50


def test_unwritten_buffers_do_not_compare_equal(solved):
    prod, s, a, _ = solved
    fresh = s.outputs().poison(0x55)
    other = s.outputs().poison(0xAA)
    for k in range(s.n):
        assert not product_equal(prod, s, fresh, other, k) and output_diff(s.inputs[k], fresh.arr[k], other.arr[k]) is not None
        assert not product_equal(prod, s, a, fresh, k) and output_diff(s.inputs[k], a.arr[k], fresh.arr[k]) is not None


def _block_with(s, a, pred):
    return next(k for k in range(s.n) if pred(s.inputs[k], a.arr[k]))


PERTURBATIONS = {
    # field class -> (which block qualifies, how to flip one element, how to undo)
    "h1": (lambda i, o: i.n_hets > 1, lambda o: o.h1.__setitem__(0, o.h1[0] ^ 1)),
    "h2": (lambda i, o: i.n_hets > 1, lambda o: o.h2.__setitem__(0, o.h2[0] ^ 1)),
    "stats": (lambda i, o: True, lambda o: setattr(o.stats, "actual_cost", o.stats.actual_cost + 1)),
    "stats.pruned": (lambda i, o: True, lambda o: setattr(o.stats, "pruned_solutions", o.stats.pruned_solutions + 1)),
    "span_counts": (lambda i, o: i.n_hets > 2, lambda o: o.span_counts.__setitem__(1, o.span_counts[1] + 1)),
    "n_segments": (lambda i, o: o.n_segments > 1, lambda o: setattr(o, "n_segments", o.n_segments - 1)),
    "n_solver": (lambda i, o: True, lambda o: setattr(o, "n_solver", o.n_solver + 1)),
    "seg_qname": (lambda i, o: o.n_segments > 0, lambda o: o.seg_qname.__setitem__(0, o.seg_qname[0] + 1)),
    "seg_start": (lambda i, o: o.n_segments > 0, lambda o: o.seg_start.__setitem__(0, o.seg_start[0] + 1)),
    "seg_end": (lambda i, o: o.n_segments > 1, lambda o: o.seg_end.__setitem__(o.n_segments - 1, o.seg_end[o.n_segments - 1] + 1)),
    "seg_solver": (lambda i, o: o.n_segments > 0, lambda o: o.seg_solver.__setitem__(0, o.seg_solver[0] ^ 1)),
    "seg_haplotag": (lambda i, o: o.n_segments > 0, lambda o: o.seg_haplotag.__setitem__(0, (o.seg_haplotag[0] + 1) % 3)),
    "seg_first_het": (lambda i, o: o.n_segments > 0, lambda o: o.seg_first_het.__setitem__(0, (o.seg_first_het[0] + 1) & 0xFFFFFFFF)),
    "seg_row_off": (lambda i, o: o.n_segments > 1, lambda o: o.seg_row_off.__setitem__(1, o.seg_row_off[1] + 1)),
    "seg_alleles (one cell)": (lambda i, o: o.n_segments > 0 and o.seg_row_off[o.n_segments] > 3,
                               lambda o: o.seg_alleles.__setitem__(o.seg_row_off[o.n_segments] - 1, o.seg_alleles[o.seg_row_off[o.n_segments] - 1] ^ 1)),
    "seg_quals (one cell)": (lambda i, o: o.n_segments > 0 and o.seg_row_off[o.n_segments] > 3, lambda o: o.seg_quals.__setitem__(2, o.seg_quals[2] ^ 4)),
    "num_reads": (lambda i, o: True, lambda o: setattr(o, "num_reads", o.num_reads + 1)),
    "skipped_reads": (lambda i, o: True, lambda o: setattr(o, "skipped_reads", o.skipped_reads + 1)),
    "global_aligned": (lambda i, o: True, lambda o: setattr(o, "global_aligned", o.global_aligned + 1)),
    "local_aligned": (lambda i, o: True, lambda o: setattr(o, "local_aligned", o.local_aligned + 1)),
    "num_alleles": (lambda i, o: True, lambda o: setattr(o, "num_alleles", o.num_alleles + 1)),
    "inexact_matches[0]": (lambda i, o: True, lambda o: o.inexact_matches.__setitem__(0, o.inexact_matches[0] + 1)),
    "failed_matches[9]": (lambda i, o: True, lambda o: o.failed_matches.__setitem__(9, o.failed_matches[9] + 1)),
    "allele0_matches[1]": (lambda i, o: True, lambda o: o.allele0_matches.__setitem__(1, o.allele0_matches[1] + 1)),
    "allele1_matches[10]": (lambda i, o: True, lambda o: o.allele1_matches.__setitem__(10, o.allele1_matches[10] + 1)),
    "exact_matches[3]": (lambda i, o: True, lambda o: o.exact_matches.__setitem__(3, o.exact_matches[3] + 1)),
    "edit_distances (last)": (lambda i, o: o.n_edit_distances > 1, lambda o: o.edit_distances.__setitem__(o.n_edit_distances - 1, o.edit_distances[o.n_edit_distances - 1] + 1)),
    "n_edit_distances": (lambda i, o: o.n_edit_distances > 1, lambda o: setattr(o, "n_edit_distances", o.n_edit_distances - 1)),
    "status": (lambda i, o: True, lambda o: setattr(o, "status", 2)),
}


@pytest.mark.parametrize("name", sorted(PERTURBATIONS))
def test_one_perturbed_field_is_a_mismatch(solved, name):
    prod, s, a, b = solved
    pred, flip = PERTURBATIONS[name]
    k = _block_with(s, a, pred)
    assert product_equal(prod, s, a, b, k) and output_diff(s.inputs[k], a.arr[k], b.arr[k]) is None
    o = a.arr[k]
    saved = bytes(C.string_at(C.addressof(o), C.sizeof(o)))
    cells = o.seg_row_off[o.n_segments] if o.n_segments else 0
    arrays = {f: [getattr(o, f)[j] for j in range(n)] for f, n in (
        ("h1", s.inputs[k].n_hets), ("h2", s.inputs[k].n_hets), ("span_counts", max(0, s.inputs[k].n_hets - 1)), ("seg_qname", o.n_segments), ("seg_start", o.n_segments),
        ("seg_end", o.n_segments), ("seg_solver", o.n_segments), ("seg_haplotag", o.n_segments), ("seg_first_het", o.n_segments), ("seg_row_off", o.n_segments + 1),
        ("seg_alleles", cells), ("seg_quals", cells), ("edit_distances", o.n_edit_distances))}
    try:
        flip(o)
        assert not product_equal(prod, s, a, b, k), f"hp_block_output_equal accepts a perturbed {name}"
        assert not product_equal(prod, s, b, a, k)
        assert output_diff(s.inputs[k], a.arr[k], b.arr[k]) is not None, f"output_diff accepts a perturbed {name}"
    finally:
        C.memmove(C.addressof(o), saved, len(saved))
        for f, vals in arrays.items():
            for j, v in enumerate(vals):
                getattr(o, f)[j] = v
    assert product_equal(prod, s, a, b, k) and output_diff(s.inputs[k], a.arr[k], b.arr[k]) is None
