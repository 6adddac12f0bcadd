"""Host-side mirror of reference src/read_parsing.rs:520-637 (`load_full_read_segments`, global mode) for
records that are already decoded (BAM decoding is out of scope and stays in htslib/Rust): one
hp_wfa_assign_batch call per block replaces the per-record graph build + WFA (read_parsing.rs:769-780), then the
order-dependent tail is replayed on the host exactly as the reference runs it:
  * Err(MaxEditDistance) -> local re-alignment of that record (read_parsing.rs:564-575);
  * the `global_disabled` switch (read_parsing.rs:597-600) — each record's WFA result is independent of the
    others, only the decision to USE it depends on the records before it;
  * qualities = 2 x base(type) (read_parsing.rs:803-835), ReadSegment::new, collapse per qname, the
    min_matched_alleles split (read_parsing.rs:611-629).
"""
from dataclasses import dataclass, field

import numpy as np

from .read_segments import AlleleType, BlockMatrix, ReadSegment
from .wfa_graph import BASE_QUAL, PreparedWfaBatch, VariantType, WfaJobSpec


@dataclass
class GlobalRealignmentConfig:
    """read_parsing.rs:25-34 (defaults: cli.rs:189,196,203,210)"""
    max_edit_distance: int = 500
    wfa_prune_distance: int = 500
    global_failure_ratio: float = 0.5
    global_failure_minimum: int = 50


@dataclass
class AlignedRecord:
    """What global_realignment needs from one BAM record (read_parsing.rs:672-742)."""
    qname: str
    min_position: int      # first reference base of the alignment
    max_position: int      # last reference base of the alignment (inclusive)
    read_align: bytes      # seq[read_start..=read_end]


@dataclass
class LoadStats:
    num_reads: int = 0
    skipped_reads: int = 0
    global_aligned: int = 0
    local_aligned: int = 0
    edit_distances: list = field(default_factory=list)


def _overlap_range(variants, lo, hi):
    """indices [first, last) of variants with lo <= position <= hi (read_parsing.rs:688-700, 721-730)."""
    first, last = None, 0
    for i, v in enumerate(variants):
        if lo <= v.position <= hi:
            if first is None:
                first = i
            last = i + 1
    return first, last


def load_full_read_segments(records, variant_calls, hom_calls, reference, ref_base=0, min_matched_alleles=2,
                            config=None, local_realignment=None, device_id=0):
    """Returns (read_segments, phasable_segments, stats): lists of ReadSegment in first-seen qname order."""
    config = config or GlobalRealignmentConfig()
    n_var = len(variant_calls)
    jobs, meta = [], []
    for rec in records:
        first, last = _overlap_range(variant_calls, rec.min_position, rec.max_position)
        if first is None:
            meta.append(None)  # no overlaps: skipped read (read_parsing.rs:703-712)
            continue
        hf, hl = _overlap_range(hom_calls, rec.min_position, rec.max_position)
        homs = hom_calls[hf:hl] if hf is not None else []
        jobs.append(WfaJobSpec(reference=reference, ref_start=rec.min_position, ref_end=rec.max_position + 1,
                               hets=variant_calls[first:last], homs=homs, read=rec.read_align, ref_base=ref_base))
        meta.append((len(jobs) - 1, first, last))
    results = PreparedWfaBatch(jobs).run(config.wfa_prune_distance, config.max_edit_distance, device_id) if jobs else []

    stats = LoadStats()
    groups = {}
    global_disabled = False
    num_global_failures = 0.0
    total_parsed = 0.0
    for rec, m in zip(records, meta):
        if m is None:
            stats.skipped_reads += 1
            continue
        j, first, last = m
        status, score, _, al = results[j]
        use_local = global_disabled or status != 0
        if use_local:
            if local_realignment is None:
                raise NotImplementedError("record needs local re-alignment (read_parsing.rs:121-503), which needs the "
                                          "BAM CIGAR; pass local_realignment=callable(record) -> (alleles, quals)")
            alleles, quals = local_realignment(rec)
            wfa_score = config.max_edit_distance
            stats.local_aligned += 1
        else:
            alleles = np.full(n_var, int(AlleleType.NoOverlap), np.uint8)
            alleles[first:last] = al
            quals = np.zeros(n_var, np.uint8)
            for i in range(first, last):
                if alleles[i] < 2:
                    quals[i] = 2 * BASE_QUAL[VariantType(variant_calls[i].variant_type)]
            wfa_score = score
            stats.global_aligned += 1
        groups.setdefault(rec.qname, []).append(ReadSegment(rec.qname, alleles.tolist(), np.asarray(quals).tolist()))
        stats.edit_distances.append(int(wfa_score))
        num_global_failures += 1.0 if use_local else 0.0
        total_parsed += 1.0
        if (not global_disabled and num_global_failures >= config.global_failure_minimum
                and num_global_failures / total_parsed >= config.global_failure_ratio):
            global_disabled = True  # read_parsing.rs:597-600
    read_segments, phasable = [], []
    for qname, grp in groups.items():
        col = ReadSegment.collapse(grp)
        num_set = col.get_num_set()
        if num_set >= min_matched_alleles:
            read_segments.append(col)
            stats.num_reads += len(grp)
        else:
            stats.skipped_reads += len(grp)
            if num_set > 0:
                phasable.append(col)
    return read_segments, phasable, stats
