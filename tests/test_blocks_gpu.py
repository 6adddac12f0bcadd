"""The whole-block C entry (hp_solve_blocks / hp_blockset_*, hiphase_amd/csrc/hp_block.hip): graph-WFA over the records
of all blocks in one device batch, fallback + `global_disabled` replay, quality assignment, collapse, A*, span counts
and haplotags INSIDE the library - checked against (a) the same pipeline assembled record by record from the CPU
oracle in the reference's order and (b) the stage-by-stage mirror above the ABI (hiphase_amd.phaser.solve_block)."""
import ctypes as C

import numpy as np
import pytest

from e2e_util import make_block
from hiphase_amd.blocks import BlockSet, BlockSpec, solve_blocks
from hiphase_amd.phaser import solve_block
from hiphase_amd.read_parsing import GlobalRealignmentConfig, LocalRecord
from hiphase_amd.read_segments import BlockMatrix
from hiphase_amd.wfa_graph import VariantType
from local_util import make_local_block
from oracle_ffi import oracle, oracle_solve, oracle_solve_blocks
from test_e2e_gpu import oracle_pipeline
from test_local_gpu import oracle_segments, reference_order_replay, seg_tuple, to_aligned

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["compact", "compact-gen2", "dense-band"])
def wfa_path(request, monkeypatch):
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0" if request.param.startswith("compact") else "1000000000")
    monkeypatch.setenv("HP_WFA_GEN", "2" if request.param == "compact-gen2" else "3")   # hp_wfa3_kernel (flat slot lists) / hp_wfa2_kernel (hull arenas)
    return request.param


def check_against_oracle(res, osegs, hets, ophasable=None):
    """res (BlockResult) == the oracle-assembled solver segments -> oracle A* -> oracle post-processing"""
    solver = [s for s in res.segments if s[5]]
    assert [(s[0], s[1], s[2], s[3], s[4]) for s in solver] == [seg_tuple(s) for s in osegs]
    if ophasable is not None:
        assert [(s[0], s[1], s[2], s[3], s[4]) for s in res.segments if not s[5]] == [seg_tuple(s) for s in ophasable]
    flags = np.asarray([(1 if v.is_ignored else 0) | (2 if v.variant_type == VariantType.Snv else 0) for v in hets], np.uint8)
    om = BlockMatrix.from_segments(osegs, len(hets), flags)
    h1, h2, st, _ = oracle_solve(om)
    assert np.array_equal(res.haplotype_1, h1) and np.array_equal(res.haplotype_2, h2) and res.statistics == st
    d = oracle()
    v = om.view()
    spans = np.zeros(max(len(hets) - 1, 1), np.uint64)
    assert d.hpo_solution_span_counts(C.byref(v), h1.ctypes.data, h2.ctypes.data, spans.ctypes.data) == 0
    assert res.span_counts.tolist() == spans[:len(hets) - 1].tolist()
    ht = np.zeros(max(om.n_reads, 1), np.uint8)
    pb = np.zeros(max(om.n_reads, 1), np.uint64)
    idx = np.arange(len(hets), dtype=np.uint64)   # block_tags[i] = i: phase_block comes back as the first het index
    assert d.hpo_haplotag_reads(C.byref(v), h1.ctypes.data, h2.ctypes.data, idx.ctypes.data, ht.ctypes.data, pb.ctypes.data) == 0
    exp = {osegs[i].read_name: (int(pb[i]), int(ht[i])) for i in range(om.n_reads) if ht[i] != 2}
    got = {q: t for q, t in res.haplotags.items() if q in {s.read_name for s in osegs}}
    assert got == exp


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_block_entry_vs_oracle_pipeline(seed, wfa_path):
    ref, hets, homs, records, truth = make_block(seed)
    cfg = GlobalRealignmentConfig()
    res, = solve_blocks([BlockSpec(7, ref, hets, homs, records)], config=cfg)
    check_against_oracle(res, oracle_pipeline(ref, hets, homs, records, cfg), hets)
    assert res.local_aligned == 0 and res.global_aligned == len(res.edit_distances)


def test_block_entry_equals_stagewise_mirror_and_batches():
    """several blocks in ONE call == each block alone == the stage-by-stage mirror above the ABI"""
    specs, mirrors = [], []
    for seed in (4, 5, 6, 7):
        ref, hets, homs, records, _ = make_block(seed, ref_len=25000, n_hets=30, n_homs=6, n_reads=60)
        specs.append(BlockSpec(seed, ref, hets, homs, records))
        mirrors.append(solve_block(seed, records, hets, homs, ref))
    together = solve_blocks(specs)
    for spec, t, (m, matrix, segs) in zip(specs, together, mirrors):
        alone, = solve_blocks([spec])
        for r in (t, alone):
            assert np.array_equal(r.haplotype_1, m.haplotype_1) and np.array_equal(r.haplotype_2, m.haplotype_2)
            assert r.statistics == m.statistics
            assert [(s[0], s[1], s[2], s[3], s[4]) for s in r.segments if s[5]] == [seg_tuple(s) for s in segs]
            tags = m.block_ids
            assert {q: (tags[f], h) for q, (f, h) in r.haplotags.items()} == m.haplotags
        assert t.segments == alone.segments and t.span_counts.tolist() == alone.span_counts.tolist()


@pytest.mark.parametrize("max_ed,minimum,ratio,expect_flip", [(4, 5, 0.3, True), (8, 10, 0.9, False), (3000, 1, 0.5, False)])
def test_block_entry_fallback_replay(max_ed, minimum, ratio, expect_flip, wfa_path):
    """Err(MaxEditDistance) -> local re-alignment and the order-dependent global_disabled switch (read_parsing.rs:556-600),
    now replayed inside the library, against the reference's record-by-record order on the oracle."""
    ref, variants, truth, lrecs = make_local_block(21, ref_len=20000, n_vars=100, n_reads=120, read_len=(800, 3000), noise=0.004)
    hets = [v for v in variants if int(v.variant_type) in (0, 1, 2, 3)]
    records = [to_aligned(r) for r in lrecs if any(op in "M=X" for op, _ in r.cigar)]
    cfg = GlobalRealignmentConfig(max_edit_distance=max_ed, wfa_prune_distance=max_ed, global_failure_minimum=minimum, global_failure_ratio=ratio)
    res, = solve_blocks([BlockSpec(1, ref, hets, [], records)], config=cfg)
    osegs, n_local, n_global, flipped = reference_order_replay(oracle(), ref, hets, records, cfg)
    assert flipped == expect_flip
    assert (res.local_aligned, res.global_aligned) == (n_local, n_global)
    # the rest of ReadStats (num_alleles + exact / inexact / failed / allele0 / allele1 per type, phase_stats.rs:12-33) on a block
    # with fallbacks: against the count made in Python record by record, and against hpo_solve_block
    assert res.read_stats == reference_order_replay.joint and res.read_stats[0] > 0
    ores, = oracle_solve_blocks([BlockSpec(1, ref, hets, [], records)], config=cfg)
    assert same_result(res, ores)
    check_against_oracle(res, osegs, hets)


def test_block_entry_local_mode():
    """--disable-global-realignment: load_read_segments inside the library (read_parsing.rs:47-113)"""
    ref, variants, truth, records = make_local_block(11, ref_len=20000, n_vars=120, n_reads=200, read_len=(1500, 6000))
    records.append(LocalRecord(records[0].qname, records[5].pos, records[5].cigar, records[5].seq, records[5].qual))
    res, = solve_blocks([BlockSpec(3, ref, variants, [], records)], global_realignment=False)
    osegs, ophas = oracle_segments(oracle(), records, variants)
    check_against_oracle(res, osegs, variants, ophas)
    assert res.read_stats == oracle_segments.joint and res.read_stats[0] > 0
    assert res.global_aligned == 0 and res.local_aligned > 0


def test_blockset_resident_resolve(wfa_path):
    """hp_blockset_*: upload once, solve twice, same answers as the one-shot entry; stage times come back"""
    specs = []
    for seed in (8, 9):
        ref, hets, homs, records, _ = make_block(seed, ref_len=25000, n_hets=30, n_homs=6, n_reads=60)
        specs.append(BlockSpec(seed, ref, hets, homs, records))
    once = solve_blocks(specs)
    bs = BlockSet(specs)
    for _ in range(2):
        ms = bs.solve()
        assert len(ms) == 8 and ms[5] > 0
        again = bs.results()
        for a, b in zip(once, again):
            assert np.array_equal(a.haplotype_1, b.haplotype_1) and a.statistics == b.statistics and a.segments == b.segments
            assert a.haplotags == b.haplotags and a.span_counts.tolist() == b.span_counts.tolist()
    bs.close()


def test_block_entry_needs_cigar_for_fallback():
    ref, variants, truth, lrecs = make_local_block(22, n_reads=10)
    hets = [v for v in variants if int(v.variant_type) in (0, 1, 2, 3)]
    records = [to_aligned(r) for r in lrecs if any(op in "M=X" for op, _ in r.cigar)]
    for r in records:
        r.local = None
    from hiphase_amd._ffi import HpError
    with pytest.raises(HpError):
        solve_blocks([BlockSpec(1, ref, hets, [], records)], config=GlobalRealignmentConfig(max_edit_distance=0, wfa_prune_distance=0))


def test_block_queue_over_devices(monkeypatch):
    """hp_solve_blocks(device_id = -1): the multi-GPU block queue (LPT chunks, two workers per device pulling
    dynamically). HP_QUEUE_WORKERS makes a 1-GPU box run it with 3 queue devices; answers equal the single-device call."""
    specs = []
    for seed in range(20, 32):
        ref, hets, homs, records, _ = make_block(seed, ref_len=15000 + 1500 * (seed % 5), n_hets=12 + 3 * (seed % 7), n_homs=4, n_reads=30 + 4 * (seed % 6))
        specs.append(BlockSpec(seed, ref, hets, homs, records))
    one = solve_blocks(specs, device_id=0)
    monkeypatch.setenv("HP_QUEUE_WORKERS", "3")
    many = solve_blocks(specs, device_id=-1)
    for a, b in zip(one, many):
        assert np.array_equal(a.haplotype_1, b.haplotype_1) and np.array_equal(a.haplotype_2, b.haplotype_2)
        assert a.statistics == b.statistics and a.segments == b.segments and a.haplotags == b.haplotags
        assert a.span_counts.tolist() == b.span_counts.tolist() and a.edit_distances == b.edit_distances


@pytest.mark.timeout(900)
def test_bench_path_two_ranks_gloo():
    """bench.py's N > 1 control flow for the whole-path workload (one process per rank, own blocks per rank, barrier +
    max-over-ranks timing, no block data between ranks), on this 1-GPU box with the gloo backend and both ranks on GPU 0."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--total-hets", "2500", "--no-cpu"], capture_output=True, text=True, timeout=800, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["hets_per_step_per_gpu"] == 2500
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and all(r["hets_per_s"] > 0 and r["elapsed_s"] <= out["ms_per_step"] * out["steps"] / 1e3 + 1e-6 for r in out["per_rank"])


@pytest.mark.timeout(900)
def test_block_set_with_large_graphs_hand_over_and_two_phases(monkeypatch):
    """Resident block set through the compact kernels where every mechanism is busy: dense variants put most reads'
    graphs in the largest class (> 128 nodes) and some beyond it (> 256: dense-band pass), 1 % noise makes reads of
    the smaller classes outgrow their tables (handed over on the device), and the results come back in two phases.
    Compared block by block with the whole path on the oracle (hpo_solve_block), twice (re-solve of the resident set)."""
    from oracle_ffi import oracle_solve_blocks
    from hiphase_amd.synth_reads import synth_read_block
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    specs = [synth_read_block(4100 + i, n, block_index=i, het_spacing=sp, coverage=20.0, noise=nz)[0]
             for i, (n, sp, nz) in enumerate([(60, 300.0, 0.01), (25, 700.0, 0.003), (90, 220.0, 0.006), (12, 1000.0, 0.01), (150, 450.0, 0.01)])]
    cfg = GlobalRealignmentConfig()
    expect = oracle_solve_blocks(specs, config=cfg)
    bs = BlockSet(specs, config=cfg)
    try:
        for _ in range(2):
            bs.solve()
            assert all(same_result(r, e) for r, e in zip(bs.results(), expect))
    finally:
        bs.close()


def same_result(a, b):
    return (np.array_equal(a.haplotype_1, b.haplotype_1) and np.array_equal(a.haplotype_2, b.haplotype_2) and a.statistics == b.statistics
            and a.segments == b.segments and a.haplotags == b.haplotags and a.span_counts.tolist() == b.span_counts.tolist()
            and a.edit_distances == b.edit_distances and (a.num_reads, a.skipped_reads, a.global_aligned, a.local_aligned) ==
            (b.num_reads, b.skipped_reads, b.global_aligned, b.local_aligned) and a.read_stats == b.read_stats and a.status == b.status)


def test_block_entry_bam4_reads_equal_ascii(wfa_path):
    """HP_SEQ_BAM4: the records' bases handed over in the BAM's own 4-bit encoding (even and odd read_offset), expanded on the
    device (compact path) or decoded on the host (dense-band path) - same results as the ASCII hand-over, which is held to the
    oracle above. Includes forced MaxEditDistance fallbacks, whose local re-alignment reads the 4-bit record too."""
    from hiphase_amd import _ffi
    specs = []
    for seed in (31, 32, 33):
        ref, hets, homs, records, _ = make_block(seed, ref_len=25000, n_hets=30, n_homs=6, n_reads=60)
        specs.append(BlockSpec(seed, ref, hets, homs, records))
    a = solve_blocks(specs)
    b = solve_blocks(specs, seq_format=_ffi.SEQ_BAM4)
    assert all(same_result(x, y) for x, y in zip(a, b))
    ref, variants, truth, lrecs = make_local_block(21, ref_len=20000, n_vars=100, n_reads=120, read_len=(800, 3000), noise=0.004)
    hets = [v for v in variants if int(v.variant_type) in (0, 1, 2, 3)]
    records = [to_aligned(r) for r in lrecs if any(op in "M=X" for op, _ in r.cigar)]
    cfg = GlobalRealignmentConfig(max_edit_distance=4, wfa_prune_distance=4, global_failure_minimum=5, global_failure_ratio=0.3)
    a, = solve_blocks([BlockSpec(1, ref, hets, [], records)], config=cfg)
    b, = solve_blocks([BlockSpec(1, ref, hets, [], records)], config=cfg, seq_format=_ffi.SEQ_BAM4)
    assert a.local_aligned > 0 and same_result(a, b)


def test_block_entry_bam4_local_mode():
    from hiphase_amd import _ffi
    ref, variants, truth, records = make_local_block(11, ref_len=20000, n_vars=120, n_reads=200, read_len=(1500, 6000))
    a, = solve_blocks([BlockSpec(3, ref, variants, [], records)], global_realignment=False)
    b, = solve_blocks([BlockSpec(3, ref, variants, [], records)], global_realignment=False, seq_format=_ffi.SEQ_BAM4)
    assert same_result(a, b)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("fmt", ["ascii", "bam4"])
def test_blockstream_equals_one_shot_entry(fmt, monkeypatch):
    """hp_blockstream_*: seven block sets in flight three deep (layout + PCIe of one, graph-WFA of another, rows + A* + post of
    a third at the same time), through the compact kernels; every set's results equal hp_solve_blocks on that set, in order.
    Set 3 is too small for the compact path (latency path inside the pipeline), set 5 is empty."""
    from hiphase_amd import _ffi
    from hiphase_amd.blocks import BlockStream
    from hiphase_amd.synth_reads import synth_read_block
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "64")
    seq_format = _ffi.SEQ_BAM4 if fmt == "bam4" else _ffi.SEQ_ASCII
    cfg = GlobalRealignmentConfig()
    sets = []
    for k in range(7):
        nb = [5, 3, 6, 1, 4, 0, 5][k]
        sets.append([synth_read_block(7000 + 10 * k + i, 10 + 7 * ((i + k) % 4), block_index=i, coverage=(3.0 if k == 3 else 14.0))[0] for i in range(nb)])
    expect = [solve_blocks(s, config=cfg, device_id=0) if s else [] for s in sets]
    st = BlockStream(config=cfg, device_id=0, depth=3, seq_format=seq_format)
    try:
        for rep in range(2):
            tickets = [st.submit(s) for s in sets[:3]]
            for k in range(3, len(sets)):
                res, ms, work = st.wait(tickets[k - 3])
                assert len(ms) == 16 and ms[7] > 0
                assert all(same_result(x, y) for x, y in zip(res, expect[k - 3])) and len(res) == len(expect[k - 3])
                tickets.append(st.submit(sets[k]))
            for k in range(len(sets) - 3, len(sets)):
                res, ms, work = st.wait(tickets[k])
                assert all(same_result(x, y) for x, y in zip(res, expect[k])) and len(res) == len(expect[k])
    finally:
        st.close()


def test_blockstream_reports_a_failed_set_and_goes_on():
    """a malformed set (a record whose alignment ends before it starts) fails ITS wait with the reference's assert
    (read_parsing.rs:685) - the sets around it are solved"""
    from hiphase_amd._ffi import HpError
    from hiphase_amd.blocks import BlockStream
    from hiphase_amd.synth_reads import synth_read_block
    good = [synth_read_block(7300 + i, 12, block_index=i, coverage=8.0)[0] for i in range(3)]
    bad = [synth_read_block(7400, 12, coverage=8.0)[0]]
    bad[0].records[2].max_position = bad[0].records[2].min_position - 5
    expect = solve_blocks(good, device_id=0)
    st = BlockStream(device_id=0, depth=2)
    try:
        t1, t2 = st.submit(good), st.submit(bad)
        r1, _, _ = st.wait(t1)
        t3 = st.submit(good)
        with pytest.raises(HpError):
            st.wait(t2)
        r3, _, _ = st.wait(t3)
        assert all(same_result(x, y) for x, y in zip(r1, expect)) and all(same_result(x, y) for x, y in zip(r3, expect))
    finally:
        st.close()


def test_unsupported_block_is_soft_alone_together_and_on_the_queue(monkeypatch):
    """HP_BLOCK_UNSUPPORTED: a block outside the device solver's packed-key limits comes back with the soft status (segments
    filled, haplotypes untouched) whether it is submitted alone, with other blocks, or through the multi-device queue - and the
    other blocks are solved. HP_TEST_UNSUPPORTED_N makes the A* pack treat blocks of that many hets as beyond the limits."""
    specs = []
    for seed in (41, 42, 43, 44):
        ref, hets, homs, records, _ = make_block(seed, ref_len=20000, n_hets=20 + seed % 3, n_homs=4, n_reads=40)
        specs.append(BlockSpec(seed, ref, hets, homs, records))
    normal = solve_blocks(specs, device_id=0)
    victim = 2
    monkeypatch.setenv("HP_TEST_UNSUPPORTED_N", str(len(specs[victim].variant_calls)))
    assert len({len(s.variant_calls) for s in specs}) > 1 and sum(len(s.variant_calls) == len(specs[victim].variant_calls) for s in specs) >= 1
    flagged = [len(s.variant_calls) == len(specs[victim].variant_calls) for s in specs]
    alone, = solve_blocks([specs[victim]], device_id=0)
    assert alone.status == 2 and alone.segments == normal[victim].segments and not alone.haplotype_1.any() and alone.haplotags == {}
    together = solve_blocks(specs, device_id=0)
    monkeypatch.setenv("HP_QUEUE_WORKERS", "3")
    queued = solve_blocks(specs, device_id=-1)
    for res in (together, queued):
        for r, n, f in zip(res, normal, flagged):
            if f:
                assert r.status == 2 and r.segments == n.segments and not r.haplotype_1.any()
            else:
                assert same_result(r, n)
    # parameters the device solver cannot hold are a property of the call: every block is handed back
    allb = solve_blocks(specs[:2], device_id=0, min_queue_size=10 ** 9)
    assert [r.status for r in allb] == [2, 2]
    monkeypatch.delenv("HP_TEST_UNSUPPORTED_N")
    # ... and so are graph-WFA parameters beyond the kernels' range (HP_WFA_UNSUPPORTED per record -> the block, without segments)
    for min_jobs in ("0", "1000000000"):
        monkeypatch.setenv("HP_WFA2_MIN_JOBS", min_jobs)
        wide = solve_blocks(specs[:2], device_id=0, config=GlobalRealignmentConfig(max_edit_distance=70000, wfa_prune_distance=500))
        assert [(r.status, len(r.segments)) for r in wide] == [(2, 0), (2, 0)]


def test_block_sets_from_several_host_threads(monkeypatch):
    """Three host threads, each with block sets of its own through the compact kernels (per-thread device context, the
    sessions' helper threads, the shared worker pool): every solve equals the single-threaded result."""
    import threading
    from hiphase_amd.synth_reads import synth_read_block
    monkeypatch.setenv("HP_WFA2_MIN_JOBS", "0")
    cfg = GlobalRealignmentConfig()

    def make(seed):
        return [synth_read_block(seed * 100 + i, 20 + 15 * (i % 4), block_index=i, coverage=15.0)[0] for i in range(6)]

    ref = {}
    for s in (1, 2, 3):
        bs = BlockSet(make(s), config=cfg)
        bs.solve()
        ref[s] = bs.results()
        bs.close()
    bad = []

    def work(s):
        try:
            for rep in range(3):
                bs = BlockSet(make(s), config=cfg)
                for _ in range(3):
                    bs.solve()
                    for a, b in zip(bs.results(), ref[s]):
                        if not (np.array_equal(a.haplotype_1, b.haplotype_1) and np.array_equal(a.haplotype_2, b.haplotype_2)
                                and a.segments == b.segments and a.span_counts.tolist() == b.span_counts.tolist() and a.statistics == b.statistics):
                            bad.append((s, rep))
                bs.close()
        except Exception as e:   # noqa: BLE001
            bad.append((s, repr(e)))

    th = [threading.Thread(target=work, args=(s,)) for s in (1, 2, 3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert bad == []
