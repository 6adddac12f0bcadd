"""Parity of the HIP A* solver (through the C ABI) with the CPU oracle: byte-equal h1/h2, equal
PhaseStats, equal heuristic arrays and equal work counters (the search trajectory is a total order)."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from hiphase_amd import BlockMatrix, ReadSegment, ResidentBatch, astar_solver, astar_solve_batch, synth_block
from oracle_ffi import oracle, oracle_solve

pytestmark = pytest.mark.gpu


def check_block(blk, **kw):
    got = astar_solver(0, blk, **kw)
    h1, h2, st, ctr = oracle_solve(blk, **kw)
    assert np.array_equal(got.haplotype_1, h1), (got.haplotype_1.tolist(), h1.tolist())
    assert np.array_equal(got.haplotype_2, h2)
    assert got.statistics.as_tuple() == st
    return got


def check_batch(blocks, **kw):
    rb = ResidentBatch(blocks, **kw)
    ms = rb.solve()
    res, ctrs, heur = rb.results(want_heuristics=True)
    rb.close()
    for blk, r, c, h in zip(blocks, res, ctrs, heur):
        h1, h2, st, octr, oh = oracle_solve(blk, want_heuristics=True, **kw)
        assert np.array_equal(r.haplotype_1, h1) and np.array_equal(r.haplotype_2, h2)
        assert r.statistics.as_tuple() == st
        assert np.array_equal(h, oh), "heuristic arrays differ"
        assert c.as_tuple() == octr, (c.as_tuple(), octr)
    return ms


def test_golden_simple_reads():
    """get_simple_reads of astar_phaser.rs:642-660 (the reads behind test_astarnode): all-0 row qual 2,
    all-1 row qual 3 -> the optimum is the het path with cost 0."""
    g = load_golden("astar_phaser.json")["astarnode"]
    blk = BlockMatrix.from_rows([(r["alleles"], r["quals"]) for r in g["reads"]])
    got = check_block(blk)
    assert got.statistics.actual_cost == 0 and got.statistics.phased_variants == 4
    assert got.haplotype_1.tolist() == [0, 0, 0, 0] and got.haplotype_2.tolist() == [1, 1, 1, 1]


def test_golden_collapse_row_scores():
    """read_segments.rs:278-308 rows (with 2/3 cells inside the region) as a 1-read + 1-read block."""
    g = load_golden("read_segments.json")["collapse"]
    rows = [(r["alleles"], r["quals"]) for r in g["rows"]] + [(g["expect_alleles"], g["expect_quals"])]
    check_block(BlockMatrix.from_rows(rows))


def test_c1_plumbing():
    blk, _ = synth_block(50, 8, 20, 0.01, 0.02, 1)
    check_batch([blk])


@pytest.mark.parametrize("seed", range(1, 9))
def test_bruteforce_sized(seed):
    blk, _ = synth_block(7, 6, 4, 0.1, 0.05, seed)
    got = check_block(blk)
    v = blk.view()
    assert oracle().hpo_bruteforce_mec(C.byref(v)) == got.statistics.actual_cost


def test_single_variant_and_tiny():
    for n in (1, 2, 3):
        blk, _ = synth_block(n, 5, 2, 0.0, 0.0, 5 + n)
        check_block(blk)


def test_uncovered_variants_and_empty_rows():
    rows = [
        ([0, 1, 3, 3, 3, 3, 3, 3], [10, 10, 0, 0, 0, 0, 0, 0]),
        ([3, 3, 3, 3, 3, 3, 1, 0], [0, 0, 0, 0, 0, 0, 7, 9]),
        ([3, 3, 3, 3, 3, 3, 3, 3], [0] * 8),   # no set allele: inert row (start == end)
        ([1, 0, 3, 3, 3, 3, 0, 1], [5, 5, 0, 0, 0, 0, 5, 5]),
    ]
    check_block(BlockMatrix.from_rows(rows))


def test_ambiguous_with_quality():
    """Local-mode rows may carry qual > 0 on Ambiguous cells (read_parsing.rs:281-285,327):
    the mismatch predicate is on the haplotype allele, not the row allele."""
    rows = [
        ([0, 2, 1, 0, 2, 1], [9, 4, 9, 9, 3, 9]),
        ([1, 2, 0, 1, 1, 0], [8, 6, 8, 8, 8, 8]),
        ([0, 0, 2, 2, 1, 1], [7, 7, 5, 5, 7, 7]),
    ]
    check_block(BlockMatrix.from_rows(rows))


def test_ignored_variants():
    blk, _ = synth_block(120, 20, 12, 0.02, 0.02, 7, ignored_permille=80)
    got = check_block(blk)
    assert got.statistics.skipped_variants == int((blk.var_flags & 1).sum()) > 0


def test_long_rows_chain_walk():
    """Rows spanning > 96 variants force the haplotype-window chain walk (look-back beyond 3 chunks)."""
    blk, _ = synth_block(400, 12, 150, 0.03, 0.02, 9)
    check_batch([blk])


def test_high_coverage_batches():
    """Coverage > 64 rows per variant: several 64-lane batches per expansion."""
    blk, _ = synth_block(150, 150, 16, 0.05, 0.02, 10)
    check_batch([blk])


@pytest.mark.parametrize("minq,qinc,e", [(20, 1, 0.30), (50, 3, 0.25), (10, 1, 0.35), (100, 0, 0.30)])
def test_pruning_dynamics(minq, qinc, e):
    """Small queues + noisy data: threshold reset at the first prune, min_progress, full prune
    (astar_phaser.rs:497-585)."""
    blk, _ = synth_block(150, 30, 10, e, 0.02, 11)
    got = check_block(blk, min_queue_size=minq, queue_increment=qinc)
    assert got.statistics.pruned_solutions > 0


def test_reference_assert_is_reported():
    """min_queue_size=10, queue_increment=0 gives the sub-solver a single visit, so the reference's
    `assert!(solve_size >= max_clip_size.min(2))` (astar_phaser.rs:268) fires; the ABI reports it as
    HP_ERR_INVARIANT (-3) instead of returning a wrong answer — and the oracle agrees."""
    from hiphase_amd._ffi import HpError
    blk, _ = synth_block(150, 30, 10, 0.35, 0.02, 11)
    with pytest.raises(HpError) as e1:
        astar_solver(0, blk, min_queue_size=10, queue_increment=0)
    assert e1.value.code == -3
    with pytest.raises(HpError) as e2:
        oracle_solve(blk, min_queue_size=10, queue_increment=0)
    assert e2.value.code == -3


@pytest.mark.parametrize("cap0", [None, "6", "1"])
def test_noisy_default_params_overflow_retry(monkeypatch, cap0):
    """C2-noisy shape, small: the frontier outgrows the node pool of the first attempt and the host retries with a pool sized from
    how far that attempt came (hp_astar.hip, first_pass_cap and the retry loop of hp_batch_solve). HP_ASTAR_CAP0 = nodes per variant of
    the first attempt: 6 was every first attempt until round 6 (one retry here), 1 cannot even hold clean data (several); the default
    is sized from a memory budget."""
    if cap0 is not None:
        monkeypatch.setenv("HP_ASTAR_CAP0", cap0)
    blk, _ = synth_block(300, 60, 20, 0.30, 0.02, 13)
    clean, _ = synth_block(400, 30, 20, 0.01, 0.02, 14)
    check_batch([blk, clean])
    check_batch([blk])


@pytest.mark.parametrize("warm", [("4", "8"), ("8", "24"), ("48", "48")])
def test_blocks_with_open_seams_take_over_the_segments_that_verify(monkeypatch, warm):
    """Segment-parallel heuristic (hp_astar_dev.h): with warm-ups this short most seams stay open after both rounds and no block is
    accepted as a whole. The main kernel's own chain then walks such a block from its end and, at the top of every segment, holds the
    TRUE look-ahead state: a segment whose warm-up reproduced it is taken over (offset applied, its work counted), any other is
    computed in place (hp_astar_kernel.hip, heuristic_phase). Heuristic array, work counters, haplotypes and statistics must be the
    oracle's whichever segments were taken over; HP_SEG_NO_TAKEOVER=1 is the whole sequential chain, same results."""
    monkeypatch.setenv("HP_SEG_WARM", warm[0])
    monkeypatch.setenv("HP_SEG_WARM2", warm[1])
    blocks = [synth_block(n, c, 20, e, 0.02, 900 + i)[0] for i, (n, c, e) in enumerate(
        [(700, 30, 0.15), (333, 60, 0.10), (64, 30, 0.02), (1000, 30, 0.01), (97, 30, 0.2), (640, 20, 0.05)])]
    check_batch(blocks)
    check_batch(blocks[:1])
    monkeypatch.setenv("HP_SEG_NO_TAKEOVER", "1")
    check_batch(blocks[:2])


def test_batch_many_blocks_mixed():
    blocks = [synth_block(n, c, s, e, 0.02, 100 + i)[0]
              for i, (n, c, s, e) in enumerate([(15, 30, 20, 0.01), (220, 30, 20, 0.01), (63, 10, 20, 0.05),
                                                (500, 30, 20, 0.02), (5, 4, 3, 0.1), (90, 60, 20, 0.15),
                                                (33, 30, 40, 0.0), (64, 30, 20, 0.01), (65, 30, 20, 0.01)] * 3)]
    check_batch(blocks)
    res = astar_solve_batch(blocks[:5])
    for blk, r in zip(blocks[:5], res):
        h1, h2, st, _ = oracle_solve(blk)
        assert np.array_equal(r.haplotype_1, h1) and r.statistics.as_tuple() == st


def test_c2_full_size():
    """BASELINE.json configs[1]: N=5000, C=30, S=20, e=0.01, a=0.02, seed=20250509 (R=7500)."""
    blk, _ = synth_block(5000, 30, 20, 0.01, 0.02, 20250509)
    assert blk.n_reads == 7500
    check_batch([blk])


def test_many_c2_blocks_properties():
    """Full-size property checks without the oracle: determinism across solves and the MEC identity
    actual_cost == sum_r min(score(h1), score(h2)) recomputed on the host from the returned haplotypes."""
    blocks = [synth_block(2000, 30, 20, 0.01, 0.02, 7000 + i)[0] for i in range(24)]
    rb = ResidentBatch(blocks)
    rb.solve()
    r1, c1, _ = rb.results()
    rb.solve()
    r2, c2, _ = rb.results()
    rb.close()
    for a, b in zip(r1, r2):
        assert np.array_equal(a.haplotype_1, b.haplotype_1) and a.statistics.as_tuple() == b.statistics.as_tuple()
    for blk, r in zip(blocks[:6], r1):
        tot = 0
        for seg in blk.segments():
            al = np.asarray(seg.alleles)
            q = np.asarray(seg.quals, dtype=np.int64)
            sl = slice(seg.start, seg.end)
            s = []
            for h in (r.haplotype_1[sl], r.haplotype_2[sl]):
                s.append(int(q[(h < 2) & (al != h)].sum()))
            tot += min(s)
        assert tot == r.statistics.actual_cost
        assert r.statistics.actual_cost >= r.statistics.estimated_cost


def test_random_stress_vs_oracle():
    """Many small random blocks x several queue parameter sets (rare paths: full prune, chunk transitions,
    main-search scratch retry, ignored variants, rows longer than 3 chunks, coverage > 64)."""
    from hiphase_amd._ffi import HpError
    rng = np.random.default_rng(2025)
    param_sets = [(1000, 3), (200, 2), (60, 1), (25, 1)]
    for pi, (minq, qinc) in enumerate(param_sets):
        blocks, expect = [], []
        for i in range(40):
            n = int(rng.integers(1, 260))
            c = int(rng.integers(2, 90))
            s = int(rng.integers(2, 110))
            e = float(rng.choice([0.0, 0.01, 0.05, 0.15, 0.3]))
            a = float(rng.choice([0.0, 0.02, 0.1]))
            ign = int(rng.choice([0, 0, 60]))
            blk, _ = synth_block(n, c, s, e, a, 9000 + 100 * pi + i, ignored_permille=ign)
            try:
                exp = oracle_solve(blk, min_queue_size=minq, queue_increment=qinc, want_heuristics=True)
            except HpError as err:
                assert err.code == -3
                with pytest.raises(HpError):
                    astar_solver(0, blk, min_queue_size=minq, queue_increment=qinc)
                continue
            blocks.append(blk)
            expect.append(exp)
        rb = ResidentBatch(blocks, min_queue_size=minq, queue_increment=qinc)
        rb.solve()
        res, ctrs, heur = rb.results(want_heuristics=True)
        rb.close()
        for k, (r, c, h, exp) in enumerate(zip(res, ctrs, heur, expect)):
            h1, h2, st, octr, oh = exp
            assert np.array_equal(r.haplotype_1, h1) and np.array_equal(r.haplotype_2, h2), (pi, k)
            assert r.statistics.as_tuple() == st, (pi, k)
            assert np.array_equal(h, oh), (pi, k)
            assert c.as_tuple() == octr, (pi, k, c.as_tuple(), octr)


def test_threads_are_reentrant():
    """The ABI is called from HiPhase's --threads pool (main.rs:332,385): concurrent callers, each with its own
    blocks, must all get the oracle's answers."""
    import threading
    blocks = [synth_block(120 + 7 * i, 20, 15, 0.03, 0.02, 4000 + i)[0] for i in range(12)]
    expect = [oracle_solve(b)[:3] for b in blocks]
    errors = []

    def work(tid):
        try:
            for rep in range(3):
                for i in range(tid, len(blocks), 4):
                    r = astar_solver(i, blocks[i])
                    h1, h2, st = expect[i]
                    assert np.array_equal(r.haplotype_1, h1) and np.array_equal(r.haplotype_2, h2)
                    assert r.statistics.as_tuple() == st
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_postprocess_span_counts_and_haplotags():
    """hp_batch_postprocess == get_solution_span_counts (phaser.rs:350-388) + haplotag_reads (phaser.rs:714-750)
    of the oracle, on the golden fixtures' shapes and on noisy synthetic blocks (homozygous calls, ignored
    variants, inert rows, ties)."""
    g = load_golden("phaser.json")
    blocks = [synth_block(n, c, s, e, 0.05, 600 + i, ignored_permille=ign)[0]
              for i, (n, c, s, e, ign) in enumerate([(200, 30, 20, 0.15, 0), (90, 12, 40, 0.3, 50), (33, 4, 3, 0.05, 0),
                                                     (1, 3, 2, 0.0, 0), (2, 6, 2, 0.2, 0), (400, 60, 20, 0.25, 30)])]
    blocks.append(BlockMatrix.from_rows([(r["alleles"], r["quals"]) for r in g["haplotag"]["reads"]]))
    blocks.append(BlockMatrix.from_rows([(r["alleles"], r["quals"]) for r in g["span_counts"]["reads"]] +
                                        [([3] * 6, [0] * 6)]))   # + an inert row
    rb = ResidentBatch(blocks)
    rb.solve()
    res, _, _ = rb.results()
    post = rb.postprocess()
    rb.close()
    d = oracle()
    for blk, r, (spans, ht, fh) in zip(blocks, res, post):
        v = blk.view()
        n = blk.n_variants
        exp_sp = np.zeros(max(n - 1, 1), np.uint64)
        assert d.hpo_solution_span_counts(C.byref(v), r.haplotype_1.ctypes.data, r.haplotype_2.ctypes.data, exp_sp.ctypes.data) == 0
        assert np.array_equal(spans, exp_sp[:max(n - 1, 0)])
        tags = np.arange(n, dtype=np.uint64)
        eht = np.zeros(blk.n_reads, np.uint8)
        epb = np.zeros(blk.n_reads, np.uint64)
        assert d.hpo_haplotag_reads(C.byref(v), r.haplotype_1.ctypes.data, r.haplotype_2.ctypes.data, tags.ctypes.data,
                                    eht.ctypes.data, epb.ctypes.data) == 0
        assert np.array_equal(ht, eht)
        tagged = eht != 2
        assert np.array_equal(fh[tagged].astype(np.uint64), epb[tagged])
        assert (fh[~tagged] == 0xFFFFFFFF).all()


@pytest.mark.parametrize("target,warm,warm2", [(64, 64, 160), (64, 160, 160), (128, 200, 100), (64, 8, 160), (64, 48, 56), (64, 8, 8)])
def test_segment_parallel_heuristic_is_exact(monkeypatch, target, warm, warm2):
    """Large blocks are cut into concurrently solved segments with a cold warm-up; seams are accepted only when the
    40-value look-ahead state matches exactly. Segments below a seam that stays open are solved again with the long
    warm-up (warm=8 leaves every seam open: the second round does all the work), and a block with a seam that is still
    open falls back to the sequential chain (warm = warm2 = 8 forces that). Either way H[], haplotypes, stats and work
    counters equal the oracle's."""
    monkeypatch.setenv("HP_SEG_TARGET", str(target))
    monkeypatch.setenv("HP_SEG_WARM", str(warm))
    monkeypatch.setenv("HP_SEG_WARM2", str(warm2))
    blocks = [synth_block(n, c, s, e, 0.02, 8100 + i, ignored_permille=ign)[0]
              for i, (n, c, s, e, ign) in enumerate([(700, 30, 20, 0.01, 0), (513, 30, 20, 0.15, 0), (900, 60, 40, 0.05, 20),
                                                     (130, 30, 20, 0.01, 0), (40, 30, 20, 0.01, 0), (1300, 12, 150, 0.03, 0)])]
    check_batch(blocks)


@pytest.mark.parametrize("target,warm,warm2", [(64, 64, 160), (64, 8, 160), (32, 48, 160)])
def test_sub_solver_window_staged_in_lds_is_exact(monkeypatch, target, warm, warm2):
    """HP_SEG_CRING=1 (BASELINE.json north_star: the allele matrix staged into LDS - here the sub-solver's <= 40-variant window of the
    per-variant cell table, a 64-variant ring per wavefront that the heuristic chain feeds with one coalesced read per step,
    reference src/astar_phaser.rs:311-405): same H[], haplotypes, statistics and work counters as the oracle, with seams that close,
    seams that stay open for the second round, and blocks of coverage 60 in the batch (two tiles per variant: those launches do
    not stage). Off by default - measured neutral on one block, slower inside a stream (hp_astar.hip launch_segments)."""
    monkeypatch.setenv("HP_SEG_CRING", "1")
    monkeypatch.setenv("HP_SEG_TARGET", str(target))
    monkeypatch.setenv("HP_SEG_WARM", str(warm))
    monkeypatch.setenv("HP_SEG_WARM2", str(warm2))
    blocks = [synth_block(n, c, s, e, 0.02, 8300 + i, ignored_permille=ign)[0]
              for i, (n, c, s, e, ign) in enumerate([(700, 30, 20, 0.01, 0), (513, 30, 20, 0.15, 0), (130, 30, 20, 0.01, 0),
                                                     (40, 30, 20, 0.01, 0), (1300, 12, 150, 0.03, 30), (2100, 28, 25, 0.02, 0)])]
    check_batch(blocks)
    check_batch(blocks + [synth_block(900, 60, 40, 0.05, 0.02, 8399, ignored_permille=20)[0]])


def test_single_block_latency_uses_segments(monkeypatch):
    """One 5000-het block alone (BASELINE.json configs[1] as written): the planner segments it by itself."""
    monkeypatch.delenv("HP_SEG_TARGET", raising=False)
    blk, _ = synth_block(5000, 30, 20, 0.01, 0.02, 20250509)
    ms_seg = check_batch([blk])
    monkeypatch.setenv("HP_NO_SEGMENTS", "1")
    rb = ResidentBatch([blk])
    ms_seq = rb.solve()
    rb.close()
    assert ms_seg < ms_seq   # and typically > 10x faster


def test_multi_device_queue_path(monkeypatch):
    """hp_astar_solve_batch(device_id=-1): LPT-sorted, interleaved chunks pulled by one worker thread per device.
    With one GPU on the box the queue is exercised with 3 workers sharing it (HP_QUEUE_WORKERS)."""
    monkeypatch.setenv("HP_QUEUE_WORKERS", "3")
    blocks = [synth_block(30 + 17 * i, 20, 12, 0.03, 0.02, 5200 + i)[0] for i in range(29)]
    res = astar_solve_batch(blocks, device_id=-1)
    for blk, r in zip(blocks, res):
        h1, h2, st, _ = oracle_solve(blk)
        assert np.array_equal(r.haplotype_1, h1) and np.array_equal(r.haplotype_2, h2) and r.statistics.as_tuple() == st


def test_c2_noisy_frontier_stress():
    """SURVEY.md §8(d) "C2-noisy" (C=60, e=0.15; proxy for the 60x config): > 64 candidate rows per variant, so
    the cell-table path is interleaved with VAR_NOFAST variants and multi-tile plane-word expansions."""
    blk, _ = synth_block(1500, 60, 20, 0.15, 0.02, 20250509)
    check_block(blk)


def test_cell_table_on_off(monkeypatch):
    """The incremental scoring path (per-position cell table) and the plane-word path give the same bits."""
    blocks = [synth_block(n, c, s, e, 0.03, 8800 + i, ignored_permille=ign)[0]
              for i, (n, c, s, e, ign) in enumerate([(600, 30, 20, 0.01, 0), (400, 45, 25, 0.05, 20), (300, 70, 18, 0.1, 0),
                                                     (200, 10, 40, 0.02, 50), (90, 25, 70, 0.03, 0)])]
    exp = [oracle_solve(b, want_heuristics=True) for b in blocks]
    for off in (False, True):
        if off:
            monkeypatch.setenv("HP_NO_CTAB", "1")
        rb = ResidentBatch(blocks)
        rb.solve()
        res, ctrs, hs = rb.results(want_heuristics=True)
        rb.close()
        for r, c, h, (h1, h2, st, octr, oh) in zip(res, ctrs, hs, exp):
            assert np.array_equal(r.haplotype_1, h1) and np.array_equal(r.haplotype_2, h2)
            assert r.statistics.as_tuple() == st
            assert np.array_equal(h, oh)
            assert c.as_tuple() == octr


def test_large_queue_parameters_use_the_hbm_sub_heap():
    """min_queue_size large enough that the sub-solver's queue no longer fits the LDS budget: the kernel variant with
    the queue in HBM scratch (and TILES = 2) must give the same bits."""
    blocks = [synth_block(n, c, 12, e, 0.02, 8600 + i)[0] for i, (n, c, e) in enumerate([(60, 20, 0.05), (90, 70, 0.1), (40, 10, 0.3)])]
    check_batch(blocks, min_queue_size=35000, queue_increment=3)


@pytest.mark.parametrize("minq", [50000, 400000])
def test_queue_sizes_beyond_39730_take_the_wide_index_keys(minq):
    """--phase-min-queue-size above 39 730 (min_queue_size / 10 + queue_increment x 40 visits, four nodes each, no longer fit
    the 14 index bits of the sub-solver's packed key): the key form with 20 index bits and 30 cost bits, same bits out.
    Noisy blocks: their sub-problems actually fill the larger queue."""
    blocks = [synth_block(n, c, 12, e, 0.02, 8700 + i)[0] for i, (n, c, e) in enumerate([(60, 20, 0.05), (120, 40, 0.15), (45, 12, 0.3), (200, 30, 0.08)])]
    check_batch(blocks, min_queue_size=minq, queue_increment=3)
