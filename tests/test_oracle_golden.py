"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §8c): read_segments.rs:214-308, astar_phaser.rs:663-798, phaser.rs:757-804,
sequence_alignment.rs:45-76, variants.rs:838-845. (wfa_graph.rs vectors: test_oracle_wfa.py.)"""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from hiphase_amd.read_segments import BlockMatrix, ReadSegment


def u8(x):
    return np.asarray(list(x), dtype=np.uint8)


def test_constructor(oracle_lib):
    g = load_golden("read_segments.json")["constructor"]
    a = u8(g["alleles"])
    s, e = C.c_size_t(), C.c_size_t()
    oracle_lib.hpo_read_segment_new(a.ctypes.data, a.size, C.byref(s), C.byref(e))
    assert [s.value, e.value] == g["expect_region"]
    assert a[s.value:e.value].tolist() == g["expect_alleles"]
    assert g["quals"][s.value:e.value] == g["expect_quals"]
    # the host mirror clips identically
    rs = ReadSegment("read_name", g["alleles"], g["quals"])
    assert (rs.start, rs.end, rs.alleles, rs.quals) == (1, 6, g["expect_alleles"], g["expect_quals"])


@pytest.mark.parametrize("key", ["score_haplotype", "score_partial_haplotype"])
def test_score(oracle_lib, key):
    g = load_golden("read_segments.json")[key]
    rs = ReadSegment("read_name", g["alleles"], g["quals"])
    if "expect_region" in g:
        assert [rs.start, rs.end] == g["expect_region"]
        assert rs.get_num_set() == g["expect_num_set"]
    ra, rq = u8(rs.alleles), u8(rs.quals)
    for case in g["cases"]:
        h = u8(case["haplotype"])
        got = oracle_lib.hpo_score_partial_haplotype(ra.ctypes.data, rq.ctypes.data, rs.start, rs.end,
                                                     h.ctypes.data, h.size, case["offset"])
        assert got == case["expect"], case


def test_collapse(oracle_lib):
    g = load_golden("read_segments.json")["collapse"]
    rows = g["rows"]
    n = len(rows[0]["alleles"])
    A = u8(sum((r["alleles"] for r in rows), []))
    Q = u8(sum((r["quals"] for r in rows), []))
    oa, oq = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    s, e = C.c_size_t(), C.c_size_t()
    assert oracle_lib.hpo_read_segment_collapse(A.ctypes.data, Q.ctypes.data, len(rows), n, oa.ctypes.data,
                                                oq.ctypes.data, C.byref(s), C.byref(e)) == 0
    assert [s.value, e.value] == g["expect_region"]
    exp = ReadSegment("read_name", g["expect_alleles"], g["expect_quals"])
    assert oa[s.value:e.value].tolist() == exp.alleles and oq[s.value:e.value].tolist() == exp.quals
    # host mirror
    segs = [ReadSegment("read_name", r["alleles"], r["quals"]) for r in rows]
    col = ReadSegment.collapse(segs)
    assert col == exp
    assert ReadSegment.collapse(segs[:1]) == segs[0]
    h = u8(g["haplotype"])
    ra, rq = u8(col.alleles), u8(col.quals)
    assert oracle_lib.hpo_score_partial_haplotype(ra.ctypes.data, rq.ctypes.data, col.start, col.end, h.ctypes.data,
                                                  h.size, 0) == g["expect_score"]


def test_astarnode(oracle_lib):
    g = load_golden("astar_phaser.json")["astarnode"]
    blk = BlockMatrix.from_rows([(r["alleles"], r["quals"]) for r in g["reads"]])
    v = blk.view()
    heur = np.asarray(g["heuristic_costs"], np.uint64)
    for w in g["walks"]:
        p1, p2 = u8(w["path1"]), u8(w["path2"])
        n = p1.size
        frozen, total, hets = (np.zeros(n, np.uint64) for _ in range(3))
        assert oracle_lib.hpo_astar_node_walk(C.byref(v), p1.ctypes.data, p2.ctypes.data, n, heur.ctypes.data,
                                              g["hap_offset"], frozen.ctypes.data, total.ctypes.data,
                                              hets.ctypes.data) == 0
        assert total.tolist() == w["expect_total"], w["name"]
        assert frozen.tolist() == w["expect_frozen"], w["name"]
        assert hets.tolist() == w["expect_hets"], w["name"]


def test_pqueuehaptracker(oracle_lib):
    g = load_golden("astar_phaser.json")["pqueuehaptracker"]
    ops = np.asarray([s[0] for s in g["script"]], np.int32)
    vals = np.asarray([s[1] for s in g["script"]], np.uint64)
    out = np.zeros(len(ops), np.uint64)
    assert oracle_lib.hpo_hap_tracker_script(g["max_hap_length"], ops.ctypes.data, vals.ctypes.data, len(ops),
                                             out.ctypes.data) == 0
    assert out.tolist() == [s[2] for s in g["script"]]


def test_span_counts(oracle_lib):
    g = load_golden("phaser.json")["span_counts"]
    blk = BlockMatrix.from_rows([(r["alleles"], r["quals"]) for r in g["reads"]])
    v = blk.view()
    h1, h2 = u8(g["h1"]), u8(g["h2"])
    out = np.zeros(len(g["h1"]) - 1, np.uint64)
    assert oracle_lib.hpo_solution_span_counts(C.byref(v), h1.ctypes.data, h2.ctypes.data, out.ctypes.data) == 0
    assert out.tolist() == g["expect"]


def test_haplotag(oracle_lib):
    g = load_golden("phaser.json")["haplotag"]
    blk = BlockMatrix.from_rows([(r["alleles"], r["quals"]) for r in g["reads"]])
    v = blk.view()
    h1, h2 = u8(g["h1"]), u8(g["h2"])
    tags = np.asarray(g["block_tags"], np.uint64)
    ht = np.zeros(blk.n_reads, np.uint8)
    pb = np.zeros(blk.n_reads, np.uint64)
    assert oracle_lib.hpo_haplotag_reads(C.byref(v), h1.ctypes.data, h2.ctypes.data, tags.ctypes.data, ht.ctypes.data,
                                         pb.ctypes.data) == 0
    for i, r in enumerate(g["reads"]):
        exp = g["expect"][r["name"]]
        if exp is None:
            assert ht[i] == 2
        else:
            assert [int(pb[i]), int(ht[i])] == exp, r["name"]


def test_edit_distance(oracle_lib):
    g = load_golden("sequence_alignment.json")
    for a, b, exp in g["edit_distance"]:
        A, B = u8(a), u8(b)
        assert oracle_lib.hpo_edit_distance(A.ctypes.data, A.size, B.ctypes.data, B.size) == exp
    ca = g["closest_allele"]
    a0, a1 = u8(ca["allele0"].encode()), u8(ca["allele1"].encode())
    for obs, exp_allele, dmin, dother in ca["cases"]:
        o = u8(obs.encode())
        d0 = oracle_lib.hpo_edit_distance(o.ctypes.data, o.size, a0.ctypes.data, a0.size)
        d1 = oracle_lib.hpo_edit_distance(o.ctypes.data, o.size, a1.ctypes.data, a1.size)
        got = (0, d0, d1) if d0 < d1 else ((1, d1, d0) if d0 > d1 else (2, d0, d1))  # variants.rs:633-640
        assert got == (exp_allele, dmin, dother), obs
