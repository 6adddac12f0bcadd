"""Pins the WFA oracle against the reference's 19 wfa_graph tests (wfa_graph.rs:677-1208): exact
(score, traversed_nodes) and exact node_to_alleles, plus hash-iteration-order independence."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from hiphase_amd import _ffi
from hiphase_amd.wfa_graph import make_jobs
from oracle_ffi import OracleGraph, oracle
from wfa_util import spec_from_golden, synth_wfa_job

G = load_golden("wfa_graph.json")


@pytest.mark.parametrize("case", G["hand_built"], ids=lambda c: c["name"])
def test_hand_built(case):
    g = OracleGraph()
    for i, n in enumerate(case["nodes"]):
        assert g.add_node(n["seq"], n["parents"]) == i
    for q in case["queries"]:
        for seed in (0, 1, 7, 12345):
            st, score, nodes = g.edit_distance(q["seq"], shuffle_seed=seed)
            assert st == 0 and score == q["score"], (case["name"], q)
            if q["nodes"] is not None:
                assert nodes == q["nodes"], (case["name"], q)


def test_add_node_errors():
    g = OracleGraph()
    assert g.add_node([1], [0]) < 0          # first node must have no parents (wfa_graph.rs:302-306)
    assert g.add_node([1], []) == 0
    assert g.add_node([2], []) < 0           # later nodes need a parent (:309-311)
    assert g.add_node([2], [1]) < 0          # parent must precede (:313-317)
    assert g.add_node([2], [0]) == 1


@pytest.mark.parametrize("case", G["variant_built"], ids=lambda c: c["name"])
def test_variant_built(case):
    d = oracle()
    spec = spec_from_golden(case)
    jobs, keep = make_jobs([spec])
    status = C.c_int(0)
    h = d.hpo_graph_from_job(C.byref(jobs[0]), 1000, C.byref(status))
    assert h and status.value == 0
    g = OracleGraph(handle=h)
    assert g.num_nodes() == case["num_nodes"]
    for node, exp in case["node_to_alleles"].items():
        assert g.node_alleles(int(node)) == [tuple(e) for e in exp], (case["name"], node)
    for q in case["queries"]:
        for seed in (0, 3, 99):
            st, score, nodes = g.edit_distance(q["seq"], shuffle_seed=seed)
            assert st == 0 and score == q["score"] and nodes == q["nodes"], (case["name"], q, score, nodes)


def test_overlapping_variants_structure():
    """wfa_graph.rs:919-923 picture: REF 0->2->4->5->6, ALT 1 rejoins at 5, ALT 3 rejoins at 6."""
    case = next(c for c in G["variant_built"] if c["name"] == "test_overlapping_variants")
    d = oracle()
    jobs, keep = make_jobs([spec_from_golden(case)])
    st = C.c_int(0)
    g = OracleGraph(handle=d.hpo_graph_from_job(C.byref(jobs[0]), 1000, C.byref(st)))
    assert [g.node_seq(i) for i in range(7)] == [b"A", b"C", b"C", b"G", b"G", b"T", b"A"]
    assert [g.node_parents(i) for i in range(7)] == [[], [0], [0], [2], [2], [1, 4], [3, 5]]
    assert g.node_edges(0) == [1, 2] and g.node_edges(2) == [3, 4]


def test_max_edit_distance_error():
    """wfa_graph.rs:645-648: Err(MaxEditDistance) once edit_distance > max."""
    g = OracleGraph(max_edit_distance=2)
    g.add_node(list(b"AAAAAAAA"), [])
    st, score, _ = g.edit_distance(list(b"CCCCCCCC"))
    assert st == 1 and score == 2
    st, score, _ = g.edit_distance(list(b"AACCAAAA"))
    assert st == 0 and score == 2     # largest returnable score == max_edit_distance


def test_pruning_changes_nothing_when_wide():
    spec, _ = synth_wfa_job(5, ref_len=1500, n_vars=6)
    d = oracle()
    jobs, keep = make_jobs([spec])
    out = _ffi.WfaResult()
    a1 = np.zeros(len(spec.hets), np.uint8)
    a2 = np.zeros(len(spec.hets), np.uint8)
    assert d.hpo_wfa_assign(C.byref(jobs[0]), 2 ** 64 - 1, 500, C.byref(out), a1.ctypes.data) == 0
    s1 = out.score
    assert d.hpo_wfa_assign(C.byref(jobs[0]), 500, 500, C.byref(out), a2.ctypes.data) == 0
    assert out.score == s1 and (a1 == a2).all()


@pytest.mark.parametrize("seed", range(1, 9))
def test_synthetic_recovers_haplotype(seed):
    """On low-noise synthetic reads the assigned allele equals the planted one or is Ambiguous/NoOverlap
    (never the opposite allele) for variants well inside the window."""
    spec, chosen = synth_wfa_job(seed, ref_len=3000, n_vars=10, noise=0.002)
    d = oracle()
    jobs, keep = make_jobs([spec])
    out = _ffi.WfaResult()
    al = np.zeros(max(1, len(spec.hets)), np.uint8)
    assert d.hpo_wfa_assign(C.byref(jobs[0]), 500, 500, C.byref(out), al.ctypes.data) == 0
    assert out.status == 0
    wrong = 0
    for i, (v, pick) in enumerate(chosen):
        if al[i] in (0, 1) and al[i] != pick:
            wrong += 1
    assert wrong <= 1, (al[:len(chosen)].tolist(), [p for _, p in chosen])


def _random_graph(r, max_nodes=12):
    """A random DAG in creation order the way add_node takes it (wfa_graph.rs:298-331): node 0 has no parents, every later node one
    to three earlier ones; sequences of 0-5 bases over a 2- or 4-letter alphabet (small alphabets make ties and repeats)."""
    g = OracleGraph(max_edit_distance=10 ** 6)
    n = r.randint(2, max_nodes)
    alpha = b"AC" if r.u01() < 0.4 else b"ACGT"
    seqs = []
    for i in range(n):
        ln = r.randint(0, 5) if r.u01() < 0.8 else r.randint(0, 12)
        s = bytes(alpha[r.randint(0, len(alpha) - 1)] for _ in range(ln))
        if i == 0:
            parents = []
        else:
            k = min(i, r.randint(1, 3))
            parents = sorted({r.randint(max(0, i - 4), i - 1) for _ in range(k)})
            if r.u01() < 0.7 and (i - 1) not in parents and r.u01() < 0.5:
                parents.append(i - 1)
        assert g.add_node(s, parents) == i
        seqs.append(s)
    return g, seqs, alpha


@pytest.mark.parametrize("chunk", range(10))
def test_score_equals_path_enumeration(chunk):
    """Independent of the wavefront code: the score of edit_distance_with_pruning (pruning off) is the minimum, over every path from
    node 0 to the last node, of the plain Levenshtein distance between the path's spelling and the query (hp_oracle_brute.cpp;
    reference src/wfa_graph.rs:350-650). The traversed set lies inside the union of the optimal paths' nodes and holds one of them
    whole. 10 x 1 100 random graphs of up to 12 nodes (the A* side has hpo_bruteforce_mec for the same purpose)."""
    from wfa_util import _Rng
    r = _Rng(9000 + chunk)
    n_cases = ties = 0
    for _ in range(1100):
        g, seqs, alpha = _random_graph(r)
        # the query: a walk through the graph with a few edits, or plain random
        if r.u01() < 0.7:
            node, q = 0, bytearray()
            while True:
                q += seqs[node]
                ch = g.node_edges(node)
                if not ch:
                    break
                node = ch[r.randint(0, len(ch) - 1)]
            q = bytearray(b for b in q if r.u01() > 0.08)
            for _k in range(r.randint(0, 3)):
                q.insert(r.randint(0, len(q)), alpha[r.randint(0, len(alpha) - 1)])
        else:
            q = bytearray(alpha[r.randint(0, len(alpha) - 1)] for _ in range(r.randint(0, 30)))
        st, score, nodes = g.edit_distance(bytes(q), shuffle_seed=r.randint(0, 3))
        assert st == 0
        best, n_paths, n_opt, union, inside = g.bruteforce(bytes(q), nodes)
        assert score == best, (chunk, seqs, bytes(q), score, best)
        assert set(nodes) <= set(union), (chunk, seqs, bytes(q), nodes, union)
        assert inside, (chunk, seqs, bytes(q), nodes)
        n_cases += 1
        ties += n_opt > 1
    assert n_cases == 1100 and ties > 50


def test_hull_statistics_count_what_the_alignment_visits():
    """hpo_wfa_hull_stats (oracle/hp_oracle_wfa.cpp): a measurement aid - per (round, node) visit the hull of the diagonals that kept a
    wave (how wide a one-read-per-wavefront kernel's steps are). It must count and must not change a result: a read equal
    to its reference visits one diagonal per node in round 0; a noisier read takes more rounds and wider hulls; the scores are the
    plain edit distances either way (reference src/wfa_graph.rs:350-650)."""
    import ctypes as C
    from oracle_ffi import oracle
    d = oracle()
    d.hpo_wfa_hull_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    d.hpo_wfa_hull_stats.restype = None
    d.hpo_graph_new.restype = C.c_void_p
    d.hpo_graph_new.argtypes = [C.c_uint64]
    d.hpo_graph_add_node.restype = C.c_int64
    d.hpo_graph_add_node.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t]
    d.hpo_graph_free.argtypes = [C.c_void_p]
    d.hpo_graph_edit_distance.restype = C.c_int
    g = d.hpo_graph_new(1000)
    try:
        assert d.hpo_graph_add_node(g, b"ACGTACGTAC", 10, None, 0) == 0
        par = (C.c_uint64 * 1)(0)
        assert d.hpo_graph_add_node(g, b"GGTTAACCGG", 10, par, 1) == 1
        st = (C.c_uint64 * 8)()
        score = C.c_uint64(0)
        trav = (C.c_uint64 * 4)()
        ntrav = C.c_size_t(0)

        def run(read):
            d.hpo_wfa_hull_stats(st, 1)
            d.hpo_graph_edit_distance.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
            ntrav.value = 4
            rc = d.hpo_graph_edit_distance(g, read, len(read), 2 ** 64 - 1, 0, C.byref(score), trav, C.byref(ntrav))
            d.hpo_wfa_hull_stats(st, 0)
            return rc, score.value, [int(x) for x in st]
        rc, sc, a = run(b"ACGTACGTACGGTTAACCGG")
        assert rc == 0 and sc == 0
        assert a[0] == 2 and a[1] == 2 and a[2] == 1 and a[7] == 0                  # two node visits, one diagonal each
        rc, sc, b = run(b"ACGTTCGTACGGTAACCGGA")
        assert rc == 0 and sc == 3                                                  # one substitution, one deletion, one insertion
        assert b[0] > a[0] and b[2] >= 2 and b[3] >= b[0] and b[4] <= b[3]          # more visits, wider hulls; chunk counts consistent
    finally:
        d.hpo_graph_free(g)
