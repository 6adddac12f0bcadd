"""CPU tests of the local re-alignment oracle (oracle/hp_oracle_local.cpp): the reference's known-answer vectors for
Variant::match_allele / closest_allele (variants.rs:668-846) and hand-derived cases of `local_realignment`
(read_parsing.rs:121-503; the reference holds no test for it — parity unpinned upstream)."""
import ctypes as C
import json
import os

import numpy as np

from hiphase_amd.phaser import add_reference_buffer, ignore_tandem_repeat_contained
from hiphase_amd.read_parsing import LocalRecord
from hiphase_amd.wfa_graph import Variant, VariantType
from local_util import make_local_block, oracle_local, pack_variants

GOLD = os.path.join(os.path.dirname(__file__), "golden", "sequence_alignment.json")

KIND = {
    "snv": lambda p, rl, a0, a1, i0, i1: Variant.new_snv(0, p, a0, a1, i0, i1),
    "deletion": lambda p, rl, a0, a1, i0, i1: Variant.new_deletion(0, p, rl, a0, a1, i0, i1),
    "insertion": lambda p, rl, a0, a1, i0, i1: Variant.new_insertion(0, p, a0, a1, i0, i1),
    "indel": lambda p, rl, a0, a1, i0, i1: Variant.new_indel(0, p, rl, a0, a1, i0, i1),
    "sv_insertion": lambda p, rl, a0, a1, i0, i1: Variant.new_sv_insertion(0, p, rl, a0, a1, i0, i1),
    "sv_deletion": lambda p, rl, a0, a1, i0, i1: Variant.new_sv_deletion(0, p, rl, a0, a1, i0, i1),
    "tandem_repeat": lambda p, rl, a0, a1, i0, i1: Variant.new_tandem_repeat(0, p, rl, a0, a1, i0, i1),
}


def u8(b):
    return np.frombuffer(bytes(b), np.uint8) if len(b) else np.zeros(1, np.uint8)


def test_match_allele_golden(oracle_lib):
    g = json.load(open(GOLD))
    for kind, pos, ref_len, a0, a1, i0, i1, cases in g["match_allele"]:
        v = KIND[kind](pos, ref_len, a0.encode(), a1.encode(), i0, i1)
        keep = []
        vs = pack_variants([v], keep)
        for obs, exp in cases:
            o = u8(obs.encode())
            assert oracle_lib.hpo_match_allele(vs, o.ctypes.data, len(obs)) == exp, (kind, obs)
            assert v.match_allele(obs.encode()) == exp   # host mirror used when packing


def test_reference_adjustment_golden(oracle_lib):
    """variants.rs:800-846 through the padded-variant path (prefix "AC", postfix "GGCC" truncated by one)."""
    g = json.load(open(GOLD))["closest_allele"]
    v = Variant.new_indel(0, 20, 2, b"A", b"AGT", 1, 2)
    assert (v.prefix_len, v.postfix_len) == (0, 0)
    v.add_reference_prefix(b"AC")
    v.add_reference_postfix(b"GGCC")
    v.truncate_reference_postfix(1)
    assert (v.prefix_len, v.postfix_len) == (g["prefix_len"], g["postfix_len"])
    assert v.get_allele0() == g["allele0"].encode() and v.get_allele1() == g["allele1"].encode()
    assert v.allele0 == g["truncated_allele0"].encode() and v.allele1 == g["truncated_allele1"].encode()
    keep = []
    vs = pack_variants([v], keep)
    for obs, exp in g["match_after_padding"]:
        o = u8(obs.encode())
        assert oracle_lib.hpo_match_allele(vs, o.ctypes.data, len(obs)) == exp
    for obs, exp_allele, dmin, dother in g["cases"]:
        o = u8(obs.encode())
        a, b = C.c_uint64(), C.c_uint64()
        assert oracle_lib.hpo_closest_allele_clip(vs, o.ctypes.data, len(obs), 0, 0, C.byref(a), C.byref(b)) == exp_allele
        assert (a.value, b.value) == (dmin, dother), obs
    # clip asserts of variants.rs:625-626
    o = u8(b"ACAG")
    a, b = C.c_uint64(), C.c_uint64()
    assert oracle_lib.hpo_closest_allele_clip(vs, o.ctypes.data, 4, 3, 0, C.byref(a), C.byref(b)) == -1
    assert oracle_lib.hpo_closest_allele_clip(vs, o.ctypes.data, 4, 0, 4, C.byref(a), C.byref(b)) == -1
    # head 2 / tail 3 strips the padding again: "A" vs "A" / "AGT"
    o = u8(b"A")
    assert oracle_lib.hpo_closest_allele_clip(vs, o.ctypes.data, 1, 2, 3, C.byref(a), C.byref(b)) == 0
    assert (a.value, b.value) == (0, 2)


REF = b"ACGTTGCAAGCTTAGGCTAACGTAGCTAGGATCCGATTACAGGCATTAGCCGATAGCTAGGCTTAAGCGCTAAGGCTAGCTAGGATATCGCGATTAGGC"


def _snv_at(pos, buffer):
    v = Variant.new_snv(0, pos, REF[pos:pos + 1], b"A" if REF[pos:pos + 1] != b"A" else b"C", 0, 1)
    add_reference_buffer([v], REF, buffer)
    return v


def test_local_hand_cases(oracle_lib):
    """Hand-derived expectations, following read_parsing.rs line by line."""
    v = _snv_at(40, 2)
    assert v.get_allele0() == REF[38:43]
    # exact reference match; all base qualities 40 -> harmonic 40 -> factor 1 -> SNV_QUAL 80 (:293-327)
    rec = LocalRecord("r", 10, [("M", 60)], REF[10:70], bytes([40]) * 60)
    al, ql, st, rc = oracle_local(oracle_lib, [rec], [v])
    assert rc == [0] and al[0, 0] == 0 and ql[0, 0] == 80
    assert st[0][0] == 0 and st[0][1] == 1 and st[0][2][0] == 1 and st[0][5][0] == 1 and st[0][7] == 1
    # qualities 20 -> harmonic 20 -> factor 0.5 -> 40; ALT base -> exact allele 1
    seq = bytearray(REF[10:70]); seq[30] = v.allele1[0]
    rec = LocalRecord("r", 10, [("M", 60)], bytes(seq), bytes([20]) * 60)
    al, ql, st, _ = oracle_local(oracle_lib, [rec], [v])
    assert al[0, 0] == 1 and ql[0, 0] == 40 and st[0][6][0] == 1
    # a third base at the site: d0 == d1 == 1 -> Ambiguous -> failed match, but the quality is still assigned (:281-327)
    third = [c for c in b"ACGT" if c not in (v.allele0[0], v.allele1[0])][0]
    seq[30] = third
    rec = LocalRecord("r", 10, [("M", 60)], bytes(seq), bytes([20]) * 60)
    al, ql, st, _ = oracle_local(oracle_lib, [rec], [v])
    assert al[0, 0] == 2 and ql[0, 0] == 40 and st[0][4][0] == 1 and st[0][0] == 1 and st[0][7] == 0
    # a zero base quality: 1/0 = inf -> harmonic 0 -> factor 0 -> .max(1.0) -> qual 1
    rec = LocalRecord("r", 10, [("M", 60)], REF[10:70], bytes([40] * 29 + [0] + [40] * 30))
    al, ql, _, _ = oracle_local(oracle_lib, [rec], [v])
    assert al[0, 0] == 0 and ql[0, 0] == 1
    # qualities mixing 10 and 40 over the 5-base window [38, 43): harmonic = 5 / (3/40 + 2/10) = 18.18.. -> 80*0.4545 = 36.36 -> 36
    q = bytearray([40] * 60); q[28] = 10; q[32] = 10
    rec = LocalRecord("r", 10, [("M", 60)], REF[10:70], bytes(q))
    _, ql, _, _ = oracle_local(oracle_lib, [rec], [v])
    assert ql[0, 0] == 36
    # the read stops before the variant's postfix window is reachable: start found, no end -> Ambiguous, overlaps (:331-337)
    rec = LocalRecord("r", 10, [("M", 31)], REF[10:41], bytes([40]) * 31)    # last aligned base = 40 = variant_pos
    al, ql, st, _ = oracle_local(oracle_lib, [rec], [v])
    assert al[0, 0] == 2 and ql[0, 0] == 0 and st[0][4][0] == 1
    # no overlap at all -> NoOverlap, read skipped (:344-349, :489)
    rec = LocalRecord("r", 50, [("M", 30)], REF[50:80], bytes([40]) * 30)
    al, _, st, _ = oracle_local(oracle_lib, [rec], [v])
    assert al[0, 0] == 3 and st[0][0] == 1 and st[0][7] == 0
    # ignored variant -> NoOverlap regardless (:180-186)
    v.is_ignored = True
    rec = LocalRecord("r", 10, [("M", 60)], REF[10:70], bytes([40]) * 60)
    al, _, st, _ = oracle_local(oracle_lib, [rec], [v])
    assert al[0, 0] == 3 and st[0][0] == 1


def test_local_sv_deletion_hand_cases(oracle_lib):
    sv = Variant.new_sv_deletion(0, 30, 21, REF[30:51], REF[30:31])     # deletes 20 bases: 31..50
    snv_inside = Variant.new_snv(0, 40, REF[40:41], b"A" if REF[40:41] != b"A" else b"C", 0, 1)
    vs = [sv, snv_inside]
    # full deletion: ratio 1.0 -> Alternate, qual 20, exact; the SNV inside the deleted span -> Ambiguous (:187-195)
    seq = REF[10:31] + REF[51:90]
    rec = LocalRecord("r", 10, [("M", 21), ("D", 20), ("M", 39)], seq, bytes([30]) * len(seq))
    al, ql, st, _ = oracle_local(oracle_lib, [rec], vs)
    assert list(al[0]) == [1, 2] and list(ql[0]) == [20, 0]
    assert st[0][2][5] == 1 and st[0][4][0] == 1 and st[0][6][5] == 1      # exact SvDeletion, failed Snv, allele1
    # no deletion: ratio 0 -> Reference, qual 20, exact; the SNV is then evaluated normally
    rec = LocalRecord("r", 10, [("M", 80)], REF[10:90], bytes([40]) * 80)
    al, ql, st, _ = oracle_local(oracle_lib, [rec], vs)
    assert list(al[0]) == [0, 0] and list(ql[0]) == [20, 80]
    # 15 of 20 deleted: ratio .75 -> |1-.75| < .33 -> Alternate, qual = 20 * 0.75 = 15, inexact
    seq = REF[10:31] + REF[46:90]
    rec = LocalRecord("r", 10, [("M", 21), ("D", 15), ("M", 44)], seq, bytes([30]) * len(seq))
    al, ql, st, _ = oracle_local(oracle_lib, [rec], vs)
    assert al[0, 0] == 1 and ql[0, 0] == 15 and st[0][3][5] == 1
    # 10 of 20 deleted: ratio .5 -> neither window -> Ambiguous
    seq = REF[10:31] + REF[41:90]
    rec = LocalRecord("r", 10, [("M", 21), ("D", 10), ("M", 49)], seq, bytes([30]) * len(seq))
    al, ql, _, _ = oracle_local(oracle_lib, [rec], vs)
    assert al[0, 0] == 2 and ql[0, 0] == 0
    # 4 of 20 deleted: ratio .2 -> Reference, qual = 20 * .8 = 16, inexact
    seq = REF[10:31] + REF[35:90]
    rec = LocalRecord("r", 10, [("M", 21), ("D", 4), ("M", 55)], seq, bytes([30]) * len(seq))
    al, ql, st, _ = oracle_local(oracle_lib, [rec], vs)
    assert al[0, 0] == 0 and ql[0, 0] == 16 and st[0][3][5] == 1
    # the read ends inside the deletion: start overlaps, far end does not -> Ambiguous with overlap (:441-447)
    rec = LocalRecord("r", 10, [("M", 30)], REF[10:40], bytes([30]) * 30)
    al, _, st, _ = oracle_local(oracle_lib, [rec], [sv])
    assert al[0, 0] == 2 and st[0][4][5] == 1
    # CIGAR Pad: rust-htslib aligned_pairs panics
    rec = LocalRecord("r", 10, [("M", 10), ("P", 2), ("M", 10)], REF[10:30], bytes([30]) * 20)
    assert oracle_local(oracle_lib, [rec], [sv])[3] == [-3]


def test_reference_buffer_and_tr_containment():
    """phaser.rs:236-294 (prefix/postfix with truncation against the previous het) and :448-513."""
    a = Variant.new_snv(0, 30, REF[30:31], b"A" if REF[30:31] != b"A" else b"C", 0, 1)
    b = Variant.new_snv(0, 36, REF[36:37], b"A" if REF[36:37] != b"A" else b"C", 0, 1)
    add_reference_buffer([a, b], REF, 15)
    assert a.prefix == REF[15:30]
    # b's prefix would start at 21 < previous_het_end 31: a's postfix [31, 46) is cut back to [31, 36), b's prefix = [31, 36)
    assert a.postfix == REF[31:36] and b.prefix == REF[31:36] and b.postfix == REF[37:52]
    first = Variant.new_snv(0, 3, REF[3:4], b"A" if REF[3:4] != b"A" else b"C", 0, 1)
    add_reference_buffer([first], REF, 15)
    assert first.prefix == REF[0:3]                      # clamps at coordinate 0
    tr = Variant.new_tandem_repeat(0, 50, 12, REF[50:62], REF[50:62] + b"ACAC", 0, 1)
    inside = Variant.new_snv(0, 55, REF[55:56], b"A" if REF[55:56] != b"A" else b"C", 0, 1)
    outside = Variant.new_snv(0, 62, REF[62:63], b"A" if REF[62:63] != b"A" else b"C", 0, 1)
    ignore_tandem_repeat_contained([inside, outside], [tr])
    assert inside.is_ignored and not outside.is_ignored and not tr.is_ignored


def test_oracle_local_synthetic_invariants(oracle_lib):
    """ReadStats::new sanity checks (phase_stats.rs:49-51) hold on noisy synthetic records of every variant type."""
    seen_types, seen_inexact = set(), 0
    for seed in range(6):
        ref, variants, truth, records = make_local_block(100 + seed)
        al, ql, stats, rcs = oracle_local(oracle_lib, records, variants)
        assert all(rc == 0 for rc in rcs)
        for i in range(len(records)):
            skipped, n_al, exact, inexact, failed, a0, a1, local = stats[i]
            assert n_al == sum(exact) + sum(inexact) == sum(a0) + sum(a1)
            assert n_al == int(((al[i] == 0) | (al[i] == 1)).sum())
            assert (skipped == 1) == (n_al == 0) and local == 1 - skipped
            assert ((ql[i] > 0) <= (al[i] < 3)).all()
            seen_inexact += sum(inexact)
        seen_types |= {int(v.variant_type) for v in variants}
    assert {0, 1, 2, 3, 4, 5, 9} <= seen_types and seen_inexact > 50
