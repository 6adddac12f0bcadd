"""End to end on a synthetic read-bearing block: HIP WFA allele assignment -> matrix -> HIP A* -> host
post-processing (hiphase_amd.phaser.solve_block) against the same pipeline assembled from the CPU oracle
(hpo_wfa_assign, hpo_astar_solve, hpo_solution_span_counts, hpo_haplotag_reads)."""
import ctypes as C

import numpy as np
import pytest

from e2e_util import make_block
from hiphase_amd import _ffi
from hiphase_amd.phaser import solve_block
from hiphase_amd.read_parsing import GlobalRealignmentConfig, load_full_read_segments
from hiphase_amd.read_segments import BlockMatrix, ReadSegment
from hiphase_amd.wfa_graph import BASE_QUAL, VariantType, WfaJobSpec, make_jobs
from oracle_ffi import oracle, oracle_solve

pytestmark = pytest.mark.gpu


def oracle_pipeline(ref, hets, homs, records, cfg):
    d = oracle()
    groups = {}
    n = len(hets)
    for rec in records:
        idx = [i for i, v in enumerate(hets) if rec.min_position <= v.position <= rec.max_position]
        if not idx:
            continue
        first, last = idx[0], idx[-1] + 1
        hs = [v for v in homs if rec.min_position <= v.position <= rec.max_position]
        spec = WfaJobSpec(ref, rec.min_position, rec.max_position + 1, hets[first:last], hs, rec.read_align)
        jobs, keep = make_jobs([spec])
        out = _ffi.WfaResult()
        al = np.full(max(1, last - first), 3, np.uint8)
        assert d.hpo_wfa_assign(C.byref(jobs[0]), cfg.wfa_prune_distance, cfg.max_edit_distance, C.byref(out), al.ctypes.data) == 0
        assert out.status == 0, "test data should not need the local fallback"
        alleles = [3] * n
        quals = [0] * n
        for k, i in enumerate(range(first, last)):
            alleles[i] = int(al[k])
            if alleles[i] < 2:
                quals[i] = 2 * BASE_QUAL[VariantType(hets[i].variant_type)]
        groups.setdefault(rec.qname, []).append(ReadSegment(rec.qname, alleles, quals))
    segs = []
    for q, grp in groups.items():
        col = ReadSegment.collapse(grp)
        if col.get_num_set() >= 2:
            segs.append(col)
    return segs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_block_end_to_end(seed):
    ref, hets, homs, records, truth = make_block(seed)
    cfg = GlobalRealignmentConfig()
    res, matrix, segs = solve_block(7, records, hets, homs, ref, global_config=cfg)
    osegs = oracle_pipeline(ref, hets, homs, records, cfg)
    assert [(s.read_name, s.start, s.end, s.alleles, s.quals) for s in segs] == \
           [(s.read_name, s.start, s.end, s.alleles, s.quals) for s in osegs]
    flags = np.asarray([2 if v.variant_type == VariantType.Snv else 0 for v in hets], np.uint8)
    om = BlockMatrix.from_segments(osegs, len(hets), flags)
    h1, h2, st, _ = oracle_solve(om)
    assert np.array_equal(res.haplotype_1, h1) and np.array_equal(res.haplotype_2, h2) and res.statistics == st
    d = oracle()
    v = om.view()
    spans = np.zeros(len(hets) - 1, np.uint64)
    assert d.hpo_solution_span_counts(C.byref(v), h1.ctypes.data, h2.ctypes.data, spans.ctypes.data) == 0
    tags, cur = [], hets[0].position
    for i, var in enumerate(hets):
        if i > 0 and spans[i - 1] == 0:
            cur = var.position
        tags.append(cur)
    assert res.block_ids == tags
    ht = np.zeros(om.n_reads, np.uint8)
    pb = np.zeros(om.n_reads, np.uint64)
    t64 = np.asarray(tags, np.uint64)
    assert d.hpo_haplotag_reads(C.byref(v), h1.ctypes.data, h2.ctypes.data, t64.ctypes.data, ht.ctypes.data, pb.ctypes.data) == 0
    exp = {osegs[i].read_name: (int(pb[i]), int(ht[i])) for i in range(om.n_reads) if ht[i] != 2}
    got = {k: v for k, v in res.haplotags.items() if k in exp or True}
    assert {k: v for k, v in got.items() if k in {s.read_name for s in osegs}} == exp
    # the phase recovered from noisy reads equals the planted one (up to the global haplotype swap) where phased
    ph = h1 != h2
    agree = (h1[ph] == np.asarray(truth)[ph]).mean()
    assert min(agree, 1 - agree) < 0.05
