"""Host-side mirror of the post-solve part of reference src/phaser.rs (`solve_block`, :406-649) over decoded
records: allele assignment (HIP WFA) -> matrix -> HIP A* -> span counts / block split / haplotags. The
post-processing (phaser.rs:350-388, :546-630, :714-750) stays on the host, as it does in the reference."""
from dataclasses import dataclass

import numpy as np

from .astar_phaser import astar_solver
from .read_parsing import load_full_read_segments, load_read_segments
from .read_segments import AlleleType, BlockMatrix
from .wfa_graph import VariantType


def get_solution_span_counts(read_segments, h1, h2):
    """phaser.rs:350-388"""
    n = len(h1)
    counts = [0] * (n - 1)
    for rs in read_segments:
        js, je = rs.start, rs.end - 1
        while js < je and h1[js] == h2[js]:
            js += 1
        while js < je and h1[je] == h2[je]:
            je -= 1
        for j in range(js, je):
            counts[j] += 1
    return counts


def haplotag_reads(read_segments, h1, h2, block_tags):
    """phaser.rs:714-750 -> {read_name: (phase block id, haplotag)}"""
    out = {}
    for rs in read_segments:
        sl = slice(rs.start, rs.end)
        al, q = np.asarray(rs.alleles), np.asarray(rs.quals, dtype=np.int64)
        s1 = int(q[(h1[sl] < 2) & (al != h1[sl])].sum())
        s2 = int(q[(h2[sl] < 2) & (al != h2[sl])].sum())
        if s1 == s2:
            continue
        first = rs.start
        while h1[first] == h2[first] or rs.allele(first) >= AlleleType.Ambiguous:
            first += 1
        assert rs.read_name not in out
        out[rs.read_name] = (block_tags[first], 0 if s1 < s2 else 1)
    return out


def add_reference_buffer(variant_calls, reference, reference_buffer=15, ref_base=0):
    """The +-reference_buffer allele padding of `load_variant_calls` (phaser.rs:236-294) for the het calls of a
    block, in VCF order: prefix = up to `reference_buffer` reference bases before the variant (not reaching into the
    previous het, whose postfix is truncated instead), postfix = `reference_buffer` bases after it. Only local
    re-alignment reads the padded alleles; the WFA path uses the truncated ones."""
    if reference_buffer <= 0:
        return
    previous_het_end = 0
    prev = None
    for v in variant_calls:
        position, ref_len = v.position, v.ref_len
        ref_prefix_start = position - reference_buffer if position > reference_buffer else 0
        ref_postfix_start = position + ref_len
        if ref_prefix_start < previous_het_end:
            assert prev is not None
            current_end = prev.position + prev.ref_len + prev.postfix_len
            prev.truncate_reference_postfix(min(current_end - position, prev.postfix_len))
            ref_prefix_start = min(previous_het_end, position)
        v.add_reference_prefix(reference[ref_prefix_start - ref_base:position - ref_base])
        v.add_reference_postfix(reference[ref_postfix_start - ref_base:ref_postfix_start + reference_buffer - ref_base])
        previous_het_end = position + ref_len
        prev = v


def ignore_tandem_repeat_contained(variant_calls, hom_calls):
    """phaser.rs:448-513: every non-TR variant fully contained in a loaded TandemRepeat call is set ignored."""
    trs = [(v.position, v.position + v.ref_len) for v in list(variant_calls) + list(hom_calls)
           if v.variant_type == VariantType.TandemRepeat]
    for v in list(variant_calls) + list(hom_calls):
        if v.variant_type != VariantType.TandemRepeat:
            start, end = v.position, v.position + v.ref_len
            if any(s <= start and e >= end and s < end and start < e for s, e in trs):
                v.is_ignored = True


@dataclass
class PhaseResult:
    """phaser.rs:326-343 (the solver-facing fields)"""
    haplotype_1: np.ndarray
    haplotype_2: np.ndarray
    block_ids: list
    sub_phase_blocks: list      # list of lists of variant indices (phased hets per sub-block)
    statistics: tuple
    haplotags: dict


def solve_block(block_index, records, variant_calls, hom_calls, reference, ref_base=0, min_matched_alleles=2,
                min_queue_size=1000, queue_increment=3, global_config=None, device_id=0, global_realignment=True):
    """phaser.rs:406-649 from the read loading on: global re-alignment (`load_full_read_segments`, records are
    AlignedRecord) or, with global_realignment=False (--disable-global-realignment, phaser.rs:521-537), local
    re-alignment (`load_read_segments`, records are LocalRecord)."""
    if global_realignment:
        segs, phasable, _stats = load_full_read_segments(records, variant_calls, hom_calls, reference, ref_base,
                                                          min_matched_alleles, global_config, device_id)
    else:
        segs, phasable, _stats, _ = load_read_segments(records, variant_calls, min_matched_alleles, device_id)
    flags = np.asarray([(1 if v.is_ignored else 0) | (2 if v.variant_type == VariantType.Snv else 0)
                        for v in variant_calls], np.uint8)
    matrix = BlockMatrix.from_segments(segs, len(variant_calls), flags)
    res = astar_solver(block_index, matrix, min_queue_size, queue_increment)
    h1, h2 = res.haplotype_1, res.haplotype_2
    spans = get_solution_span_counts(segs, h1, h2)
    tags, cur = [], variant_calls[0].position
    for i, v in enumerate(variant_calls):           # phaser.rs:557-565
        if i > 0 and spans[i - 1] == 0:
            cur = v.position
        tags.append(cur)
    subs, block, cur_tag = [], [], tags[0]          # phaser.rs:569-611
    for i in range(len(variant_calls)):
        if h1[i] < 2 and h2[i] < 2 and h1[i] != h2[i]:
            if cur_tag != tags[i]:
                if block:
                    subs.append(block)
                    block = []
                cur_tag = tags[i]
            block.append(i)
    if block:
        subs.append(block)
    ht = haplotag_reads(segs, h1, h2, tags)
    extra = haplotag_reads(phasable, h1, h2, tags)
    assert not (set(ht) & set(extra))
    ht.update(extra)
    return PhaseResult(h1, h2, tags, subs, res.statistics.as_tuple(), ht), matrix, segs
