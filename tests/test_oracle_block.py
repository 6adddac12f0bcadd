"""hpo_solve_block (oracle/hp_oracle_block.cpp) - the whole path for one block on the CPU oracle, in C - against the same
pipeline assembled record by record in Python from the oracle's pinned pieces (hpo_wfa_assign, hpo_local_realignment,
ReadSegment::new / collapse restated in hiphase_amd.read_segments, hpo_astar_solve, hpo_solution_span_counts,
hpo_haplotag_reads) in the reference's order (read_parsing.rs:546-629, phaser.rs:541-630). Two independent restatements of
the control flow the reference holds no test for; the GPU suite compares the product with both. No GPU needed."""
import numpy as np
import pytest

from e2e_util import make_block
from hiphase_amd import _ffi
from hiphase_amd.blocks import BlockSpec
from hiphase_amd.read_parsing import GlobalRealignmentConfig, LocalRecord
from local_util import make_local_block
from oracle_ffi import oracle, oracle_solve_blocks
from test_blocks_gpu import check_against_oracle, same_result
from test_e2e_gpu import oracle_pipeline
from test_local_gpu import oracle_segments, reference_order_replay, to_aligned


@pytest.mark.parametrize("seed", [1, 2])
def test_global_mode(seed):
    ref, hets, homs, records, truth = make_block(seed, ref_len=25000, n_hets=30, n_homs=6, n_reads=60)
    cfg = GlobalRealignmentConfig()
    res, = oracle_solve_blocks([BlockSpec(7, ref, hets, homs, records)], config=cfg)
    check_against_oracle(res, oracle_pipeline(ref, hets, homs, records, cfg), hets)
    assert res.local_aligned == 0 and res.global_aligned == len(res.edit_distances)
    packed, = oracle_solve_blocks([BlockSpec(7, ref, hets, homs, records)], config=cfg, seq_format=_ffi.SEQ_BAM4)
    assert same_result(res, packed)


@pytest.mark.parametrize("max_ed,minimum,ratio,expect_flip", [(4, 5, 0.3, True), (8, 10, 0.9, False), (3000, 1, 0.5, False)])
def test_fallback_replay(max_ed, minimum, ratio, expect_flip):
    ref, variants, truth, lrecs = make_local_block(21, ref_len=20000, n_vars=100, n_reads=120, read_len=(800, 3000), noise=0.004)
    hets = [v for v in variants if int(v.variant_type) in (0, 1, 2, 3)]
    records = [to_aligned(r) for r in lrecs if any(op in "M=X" for op, _ in r.cigar)]
    cfg = GlobalRealignmentConfig(max_edit_distance=max_ed, wfa_prune_distance=max_ed, global_failure_minimum=minimum, global_failure_ratio=ratio)
    res, = oracle_solve_blocks([BlockSpec(1, ref, hets, [], records)], config=cfg)
    osegs, n_local, n_global, flipped = reference_order_replay(oracle(), ref, hets, records, cfg)
    assert flipped == expect_flip
    assert (res.local_aligned, res.global_aligned) == (n_local, n_global)
    assert res.read_stats == reference_order_replay.joint and res.read_stats[0] > 0   # the rest of ReadStats, counted in Python
    assert res.read_stats[0] == sum(res.read_stats[1]) + sum(res.read_stats[2]) == sum(res.read_stats[4]) + sum(res.read_stats[5])   # phase_stats.rs:62-64
    check_against_oracle(res, osegs, hets)
    packed, = oracle_solve_blocks([BlockSpec(1, ref, hets, [], records)], config=cfg, seq_format=_ffi.SEQ_BAM4)
    assert same_result(res, packed)


def test_local_mode():
    ref, variants, truth, records = make_local_block(11, ref_len=20000, n_vars=120, n_reads=200, read_len=(1500, 6000))
    records.append(LocalRecord(records[0].qname, records[5].pos, records[5].cigar, records[5].seq, records[5].qual))
    res, = oracle_solve_blocks([BlockSpec(3, ref, variants, [], records)], global_realignment=False)
    osegs, ophas = oracle_segments(oracle(), records, variants)
    check_against_oracle(res, osegs, variants, ophas)
    assert res.read_stats == oracle_segments.joint and res.read_stats[0] > 0
    assert res.global_aligned == 0 and res.local_aligned > 0
