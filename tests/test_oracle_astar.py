"""astar_solver end-to-end is PARITY-UNPINNED upstream (no read-bearing fixture); the
oracle's evidence: brute-force MEC equality on small blocks, the reference's asserts, determinism."""
import ctypes as C

import numpy as np
import pytest

from oracle_ffi import oracle, oracle_solve, oracle_synth


@pytest.mark.parametrize("seed", range(1, 21))
def test_bruteforce_mec_small(seed):
    blk, _ = oracle_synth(7, 6, 4, 0.1, 0.05, seed)
    h1, h2, st, ctr = oracle_solve(blk)
    assert st[0] == 0  # nothing pruned => guaranteed optimum
    v = blk.view()
    assert oracle().hpo_bruteforce_mec(C.byref(v)) == st[2]
    assert st[2] >= st[1]  # actual >= estimated (phase_stats.rs:163)


def test_c1_plumbing():
    """BASELINE.json configs[0]: single synthetic block, 50 hets x 20 reads, CPU reference path."""
    blk, truth = oracle_synth(50, 8, 20, 0.01, 0.02, 1)
    assert blk.n_reads == 20
    h1, h2, st, ctr = oracle_solve(blk, want_heuristics=False)
    pruned, est, act, phased, snvs, hom, skipped = st
    assert pruned == 0 and phased + hom + skipped == 50 and act >= est
    # the recovered phase equals the planted truth (up to haplotype swap) wherever it is phased
    ph = h1 != h2
    agree = (h1[ph] == truth[ph]).mean()
    assert agree in (0.0, 1.0) or min(agree, 1 - agree) < 0.1
    assert ctr[1] >= 50 and ctr[2] > 0 and ctr[3] >= ctr[2]


def test_ignored_variants():
    blk, _ = oracle_synth(60, 10, 12, 0.02, 0.02, 7, ignored_permille=100)
    h1, h2, st, _ = oracle_solve(blk)
    ign = (blk.var_flags & 1) != 0
    assert ign.sum() > 0 and st[6] == ign.sum()
    assert (h1[ign] == 2).all() and (h2[ign] == 2).all()
    assert (h1[~ign] < 2).all()


def test_noisy_block_prunes():
    """Small queue + high error forces pruning (threshold dynamics of astar_phaser.rs:497-585)."""
    blk, _ = oracle_synth(120, 30, 10, 0.30, 0.02, 11)
    h1, h2, st, ctr = oracle_solve(blk, min_queue_size=20, queue_increment=1)
    assert st[0] > 0 and st[2] >= st[1]
    a = oracle_solve(blk, min_queue_size=20, queue_increment=1)
    assert (a[0] == h1).all() and a[2] == st  # deterministic


def test_heuristic_monotone():
    blk, _ = oracle_synth(200, 30, 20, 0.05, 0.02, 3)
    *_, heur = oracle_solve(blk, want_heuristics=True)
    assert heur[-1] == 0 and (np.diff(heur.astype(np.int64)) <= 0).all()
