"""Parity of the HIP graph-WFA allele assignment and the HIP Levenshtein kernel (through the C ABI) with
the CPU oracle and the reference's own known-answer vectors (wfa_graph.rs:842-1208, sequence_alignment.rs:45-76,
variants.rs:838-845)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden
from hiphase_amd import _ffi
from hiphase_amd.sequence_alignment import closest_allele_clip, edit_distance_batch
from hiphase_amd.wfa_graph import Variant, WfaJobSpec, make_jobs, wfa_assign_batch
from oracle_ffi import oracle
from wfa_util import spec_from_golden, synth_wfa_job, _Rng

pytestmark = pytest.mark.gpu
G = load_golden("wfa_graph.json")


@pytest.fixture(autouse=True, params=["compact", "compact-gen2", "dense-band"])
def wfa_kernel_path(request, monkeypatch):
    """hp_wfa_assign_batch picks its kernel by batch size (hp_wfa.hip); every test runs through both: the compact
    several-reads-per-wavefront kernel with the device graph builder (hp_wfa2*.hip) and the dense-band one."""
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0" if request.param.startswith("compact") else "1000000000")
    monkeypatch.setenv("HP_WFA_GEN", "2" if request.param == "compact-gen2" else "3")   # hp_wfa3_kernel (flat slot lists) / hp_wfa2_kernel (hull arenas)
    return request.param


def oracle_assign(spec, prune, max_ed):
    d = oracle()
    jobs, keep = make_jobs([spec])
    out = _ffi.WfaResult()
    al = np.full(max(1, len(spec.hets)), 3, np.uint8)
    rc = d.hpo_wfa_assign(C.byref(jobs[0]), (2 ** 64 - 1) if prune in (0, None) else prune, max_ed, C.byref(out), al.ctypes.data)
    assert rc == 0
    return out.status, out.score, out.n_nodes, al[:len(spec.hets)]


def check_specs(specs, prune=500, max_ed=500):
    got = wfa_assign_batch(specs, prune_distance=prune, max_edit_distance=max_ed)
    for i, (spec, g) in enumerate(zip(specs, got)):
        st, score, nn, al = oracle_assign(spec, prune, max_ed)
        assert (g[0], g[1], g[2]) == (st, score, nn), (i, g[:3], (st, score, nn))
        assert np.array_equal(g[3], al), (i, g[3].tolist(), al.tolist())
    return got


@pytest.mark.parametrize("case", [c for c in G["variant_built"] if c["queries"]], ids=lambda c: c["name"])
def test_golden_variant_graphs(case):
    """Exact (score, traversed nodes) of the reference's tests, observed through the allele mapping of
    read_parsing.rs:790-800 applied to the expected node sets and node_to_alleles table."""
    specs = [spec_from_golden(case, read=bytes(q["seq"])) for q in case["queries"]]
    got = wfa_assign_batch(specs, prune_distance=0, max_edit_distance=1000)
    n_hets = len(case["variants"])
    for q, g in zip(case["queries"], got):
        exp = [3] * n_hets
        for node in q["nodes"]:
            for vi, a in case["node_to_alleles"][str(node)]:
                if exp[vi] == 3:
                    exp[vi] = a
                elif exp[vi] != a:
                    exp[vi] = 2
        assert g[0] == 0 and g[1] == q["score"] and g[2] == case["num_nodes"], (case["name"], q, g)
        assert g[3].tolist() == exp, (case["name"], q, g[3].tolist(), exp)
    check_specs(specs, prune=0, max_ed=1000)


def test_golden_graph_shapes():
    for name in ("test_variant_before_start", "test_span_ref_end"):
        case = next(c for c in G["variant_built"] if c["name"] == name)
        spec = spec_from_golden(case, read=case["reference"][case["ref_start"]:case["ref_end"]].encode())
        (st, score, nn, al), = wfa_assign_batch([spec], prune_distance=0, max_edit_distance=1000)
        assert (st, score, nn) == (0, 0, case["num_nodes"])
        check_specs([spec], prune=0, max_ed=1000)


def test_synthetic_batch_default_params():
    specs = [synth_wfa_job(seed, ref_len=3000 + 37 * seed, n_vars=6 + seed % 9, noise=0.003 + 0.001 * (seed % 5))[0]
             for seed in range(1, 49)]
    check_specs(specs)


def test_synthetic_long_reads():
    specs = [synth_wfa_job(100 + s, ref_len=17000, n_vars=24, n_homs=8, noise=0.004)[0] for s in range(6)]
    check_specs(specs)


def test_noisy_reads_and_band_retry(monkeypatch):
    """Reads whose edit distance exceeds the first-pass band are re-run with a wider one."""
    monkeypatch.setenv("HP_WFA_BAND", "6")
    specs = [synth_wfa_job(200 + s, ref_len=2500, n_vars=8, noise=0.02)[0] for s in range(12)]
    got = check_specs(specs)
    assert max(g[1] for g in got) > 6


def test_max_edit_distance_fallback_status():
    """Err(MaxEditDistance) (wfa_graph.rs:645-648): status 1, score == max_ed, alleles all NoOverlap; the largest
    returnable score equals max_edit_distance."""
    spec, _ = synth_wfa_job(300, ref_len=2000, n_vars=6, noise=0.05)
    full = check_specs([spec], max_ed=500)[0]
    assert full[0] == 0 and full[1] > 10
    capped = check_specs([spec], max_ed=full[1] - 1)[0]
    assert capped[0] == 1 and capped[1] == full[1] - 1 and (capped[3] == 3).all()
    exact = check_specs([spec], max_ed=full[1])[0]
    assert exact[0] == 0 and exact[1] == full[1]


@pytest.mark.parametrize("prune", [0, 30, 150])
def test_prune_distance(prune):
    specs = [synth_wfa_job(400 + s, ref_len=4000, n_vars=10, noise=0.01)[0] for s in range(8)]
    check_specs(specs, prune=prune)


def test_edge_reads():
    spec, _ = synth_wfa_job(500, ref_len=1200, n_vars=5)
    from dataclasses import replace
    specs = [replace(spec, read=b""), replace(spec, read=b"ACGT"), replace(spec, read=spec.read[:300]),
             replace(spec, read=b"N" * 50)]
    check_specs(specs, prune=0, max_ed=2000)


def test_ignored_and_multiallelic():
    spec, _ = synth_wfa_job(600, ref_len=3000, n_vars=14, multiallelic=0.6)
    spec.hets[2].is_ignored = True
    check_specs([spec])


def test_structural_variants_put_paths_far_apart():
    """A 140-base deletion and a 90-base insertion upstream: the reference-allele and the alternate-allele path reach the
    nodes behind them 140 / 90 diagonals apart, so a node's capped diagonals do not fit one record window (the overflow
    hash set of the compact kernels) and its sources are far-apart items. Reads of both haplotypes, with noise, pruning
    off (both paths stay alive) and on."""
    import random
    rng = random.Random(77)
    ref = bytes(rng.choice(b"ACGT") for _ in range(3200))
    hets = [Variant.new_sv_deletion(0, 400, 141, ref[400:541], ref[400:401]),
            Variant.new_snv(1, 700, ref[700:701], bytes([b"ACGT"[(b"ACGT".index(ref[700]) + 1) % 4]]), 0, 1),
            Variant.new_sv_insertion(2, 1100, 1, ref[1100:1101], ref[1100:1101] + bytes(rng.choice(b"ACGT") for _ in range(90))),
            Variant.new_snv(3, 1500, ref[1500:1501], bytes([b"ACGT"[(b"ACGT".index(ref[1500]) + 2) % 4]]), 0, 1),
            Variant.new_snv(4, 2300, ref[2300:2301], bytes([b"ACGT"[(b"ACGT".index(ref[2300]) + 3) % 4]]), 0, 1)]
    specs = []
    for hap in range(4):
        seq = bytearray()
        pos = 100
        for k, v in enumerate(hets):
            seq += ref[pos:v.position]
            seq += v.allele1 if (hap >> (k % 2)) & 1 else v.allele0
            pos = v.position + v.ref_len
        seq += ref[pos:3100]
        for noise in (0.0, 0.004, 0.012):
            s = bytearray(seq)
            for i in range(len(s)):
                if rng.random() < noise:
                    s[i] = rng.choice(b"ACGT")
            specs.append(WfaJobSpec(ref, 100, 3100, hets, [], bytes(s)))
    check_specs(specs, prune=0, max_ed=500)
    check_specs(specs, prune=500, max_ed=500)
    check_specs(specs, prune=60, max_ed=500)


# ---- Levenshtein -------------------------------------------------------------------------------------------

def test_edit_distance_golden():
    g = load_golden("sequence_alignment.json")
    pairs = [(bytes(a), bytes(b)) for a, b, _ in g["edit_distance"]]
    assert edit_distance_batch(pairs) == [e for _, _, e in g["edit_distance"]]
    ca = g["closest_allele"]
    for obs, exp_allele, dmin, dother in ca["cases"]:
        assert closest_allele_clip(obs.encode(), ca["allele0"].encode(), ca["allele1"].encode()) == (exp_allele, dmin, dother)
    # clipping = comparing against the allele without part of its padding (variants.rs:624-628)
    assert closest_allele_clip(b"AGGC", ca["allele0"].encode(), ca["allele1"].encode(), head_clip=2) == (0, 0, 2)


def test_edit_distance_random_vs_oracle():
    r = _Rng(7)
    d = oracle()
    pairs = []
    for i in range(200):
        la, lb = r.randint(0, 90), r.randint(0, 90)
        a = r.dna(la)
        b = bytearray(a[:lb]) if r.u01() < 0.5 else bytearray(r.dna(lb))
        for _ in range(r.randint(0, 6)):
            if b:
                b[r.randint(0, len(b) - 1)] = b"ACGT"[r.next() & 3]
        pairs.append((a, bytes(b)))
    pairs += [(r.dna(3000), r.dna(2500)), (r.dna(130), r.dna(4000)), (b"", r.dna(70)), (r.dna(64), r.dna(64)), (r.dna(65), r.dna(63))]
    got = edit_distance_batch(pairs)
    for (a, b), g in zip(pairs, got):
        A = np.frombuffer(a, np.uint8) if a else np.zeros(1, np.uint8)
        Bv = np.frombuffer(b, np.uint8) if b else np.zeros(1, np.uint8)
        assert g == d.hpo_edit_distance(A.ctypes.data, len(a), Bv.ctypes.data, len(b)), (len(a), len(b))


def test_concurrent_callers_and_thread_exit():
    """hp_wfa_assign_batch is called from HiPhase's worker threads (main.rs:332,385; read_parsing.rs:769-780), each
    with its own records: concurrent calls must not share state, and a worker that exits hands its per-thread
    device buffers back (thread-local teardown order once corrupted the heap at exit)."""
    import threading
    specs = [synth_wfa_job(7100 + s, ref_len=1500 + 40 * s, n_vars=10, n_homs=3, noise=0.01)[0] for s in range(24)]
    expect = wfa_assign_batch(specs, prune_distance=500, max_edit_distance=500)
    errors = []

    def work(tid):
        try:
            for rep in range(3):
                mine = specs[tid::4]
                got = wfa_assign_batch(mine, prune_distance=500, max_edit_distance=500)
                for g, e in zip(got, expect[tid::4]):
                    assert g[:3] == e[:3] and np.array_equal(g[3], e[3])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    for wave in range(2):   # the second wave runs after the first wave's threads (and their caches) are gone
        ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errors, errors
    check_specs(specs[:4])


def test_jobs_sharing_one_reference_buffer():
    """The reads of a block pass windows of ONE chromosome buffer (read_parsing.rs:738-768): the library uploads the
    union of the windows' address ranges once. Overlapping, nested, adjacent and disjoint windows, plus a job with its
    own buffer, must all score as if each had its own copy."""
    from hiphase_amd.wfa_graph import WfaJobSpec
    r = _Rng(4242)
    base, _ = synth_wfa_job(5150, ref_len=9000, n_vars=26, n_homs=5, noise=0.0, multiallelic=0.2)
    L = len(base.reference)
    windows = [(0, 900), (100, 700), (700, 1500), (1500, 2300), (2299, 4000), (5000, L), (5200, 5900), (0, L)]
    windows += [(a, min(L, a + r.randint(300, 2500))) for a in (r.randint(0, L - 400) for _ in range(24))]
    specs = []
    for a, b in windows:
        read = bytearray(base.reference[a:b])
        for k in range(len(read)):
            if r.u01() < 0.01:
                read[k] = b"ACGT"[r.next() & 3]
        for v in base.hets + base.homs:   # carry some same-length alternate alleles
            if a <= v.position and v.position + v.ref_len <= b and len(v.allele1) == v.ref_len and r.u01() < 0.5:
                read[v.position - a:v.position - a + v.ref_len] = v.allele1
        specs.append(WfaJobSpec(reference=base.reference, ref_start=a, ref_end=b, hets=base.hets, homs=base.homs,
                                read=bytes(read), ref_base=0))
    specs.append(synth_wfa_job(5151, ref_len=1200, n_vars=6)[0])   # a buffer of its own in the same batch
    check_specs(specs)
    check_specs(specs[::-1], prune=50, max_ed=40)


# ---- graphs beyond the LDS budget / wide fan-in (round 2: no longer HP_ERR_UNSUPPORTED) ---------------------------------

def test_graph_beyond_the_lds_budget():
    """> 1024 nodes (hp_wfa_dev.h WFA_MAX_NODES): hp_wfa_big_kernel keeps the per-node state in HBM; a dense region of a
    real genome (a 20-kb read over several hundred calls) gets there."""
    specs = [synth_wfa_job(900 + s, ref_len=20000, n_vars=420, n_homs=60, noise=0.003, margin=30)[0] for s in range(3)]
    got = check_specs(specs)
    assert min(g[2] for g in got) > 1024, [g[2] for g in got]


def test_graphs_of_257_to_512_nodes_take_the_sixteen_word_sets(monkeypatch):
    """a read over 90 - 160 calls builds a graph no class of the launch set holds (256 nodes at most); with HP_WFA2_WIDE_MIN
    reached, those jobs go through hp_wfa2_kernel<16, 16, true> (512 nodes) instead of the dense-band kernels - same results"""
    monkeypatch.setenv("HP_WFA2_WIDE_MIN", "8")
    specs = [synth_wfa_job(1300 + s, ref_len=9000, n_vars=95 + 5 * (s % 12), n_homs=10, noise=0.004 + 0.002 * (s % 5), margin=30, multiallelic=0.1)[0] for s in range(24)]
    got = check_specs(specs)
    assert sum(1 for g in got if 256 < g[2] <= 512) >= 8, [g[2] for g in got]
    check_specs(specs, prune=60, max_ed=35)


def test_node_with_more_than_32_parents():
    """40 insertion alleles at one position all reconnect on the same reference node (41 parents)."""
    from hiphase_amd.wfa_graph import Variant, WfaJobSpec
    r = _Rng(4242)
    ref = r.dna(1500)
    pos = 600
    hets = [Variant.new_insertion(0, pos, ref[pos:pos + 1], ref[pos:pos + 1] + r.dna(3 + k % 5) + bytes([b"ACGT"[k % 4]]) * (1 + k // 4), 0, 1) for k in range(40)]
    reads = []
    for k in (0, 7, 39):
        reads.append(ref[100:pos] + hets[k].allele1 + ref[pos + 1:1400])
    reads.append(ref[100:1400])
    specs = [WfaJobSpec(reference=ref, ref_start=100, ref_end=1400, hets=hets, homs=[], read=rd) for rd in reads]
    got = check_specs(specs, prune=0, max_ed=100)
    assert all(g[0] == 0 for g in got) and got[0][2] >= 42


def test_job_beyond_the_kernels_limits_is_soft_and_alone():
    """70 insertion alleles reconnecting on one reference node (71 parents, the kernels hold 64): THAT job comes back with the
    soft status HP_WFA_UNSUPPORTED, the ordinary jobs of the same call are aligned as ever"""
    from hiphase_amd.wfa_graph import Variant, WfaJobSpec
    r = _Rng(4343)
    ref = r.dna(1500)
    pos = 600
    hets = [Variant.new_insertion(0, pos, ref[pos:pos + 1], ref[pos:pos + 1] + r.dna(3 + k % 5) + bytes([b"ACGT"[k % 4]]) * (1 + k // 4), 0, 1) for k in range(70)]
    wide = WfaJobSpec(reference=ref, ref_start=100, ref_end=1400, hets=hets, homs=[], read=ref[100:1400])
    plain = WfaJobSpec(reference=ref, ref_start=100, ref_end=1400, hets=hets[:5], homs=[], read=ref[100:pos] + hets[3].allele1 + ref[pos + 1:1400])
    got = wfa_assign_batch([plain, wide, plain], prune_distance=0, max_edit_distance=100)
    assert got[1][0] == 3 and (got[1][3] == 3).all()
    st, score, nn, al = oracle_assign(plain, 0, 100)
    for g in (got[0], got[2]):
        assert (g[0], g[1], g[2]) == (st, score, nn) and np.array_equal(g[3], al)


@pytest.mark.parametrize("case", G["hand_built"], ids=lambda c: c["name"])
def test_golden_hand_built_graphs(case):
    """The reference's hand-built topologies (wfa_graph.rs:677-839: single node, two-node splits, basic variant, triple /
    nested / double / overlapping split) through WFAGraph::add_node + edit_distance_with_pruning on the device
    (hp_wfa_align_graphs): exact (score, traversed_nodes) - no variant set builds the nested and overlapping ones."""
    from hiphase_amd.wfa_graph import WFAGraph
    g = WFAGraph()
    for i, nd in enumerate(case["nodes"]):
        assert g.add_node(nd["seq"], nd["parents"]) == i
    res = g.edit_distance_with_pruning([q["seq"] for q in case["queries"]], prune_distance=None, max_edit_distance=500)
    for q, (score, nodes) in zip(case["queries"], res):
        assert score == q["score"], (case["name"], q)
        if q["nodes"] is not None:
            assert nodes == q["nodes"], (case["name"], q)


def test_hand_built_graph_errors_and_max_ed():
    """add_node's asserts (wfa_graph.rs:305-312) and the MaxEditDistance status on caller-built graphs."""
    from hiphase_amd.wfa_graph import WFAGraph
    g = WFAGraph()
    g.add_node([0, 1, 2, 3], [])
    g.add_node([0, 1], [0])
    assert g.edit_distance_with_pruning([[3, 3, 3, 3, 3, 3, 3, 3, 3]], max_edit_distance=2) == [(None, [])]
    bad = WFAGraph()
    bad.add_node([1], [])
    bad.add_node([2], [])            # a later node without parents
    with pytest.raises(Exception):
        bad.edit_distance_with_pruning([[1, 2]])
    bad2 = WFAGraph()
    bad2.add_node([1], [])
    bad2.add_node([2], [1])          # parent must precede
    with pytest.raises(Exception):
        bad2.edit_distance_with_pruning([[1, 2]])
