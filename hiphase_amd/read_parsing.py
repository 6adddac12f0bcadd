"""Host-side mirror of reference src/read_parsing.rs:520-637 (`load_full_read_segments`, global mode) for
records that are already decoded (BAM decoding is out of scope and stays in htslib/Rust): one
hp_wfa_assign_batch call per block replaces the per-record graph build + WFA (read_parsing.rs:769-780), then the
order-dependent tail is replayed on the host exactly as the reference runs it:
  * Err(MaxEditDistance) -> local re-alignment of that record (read_parsing.rs:564-575), batched through
    hp_local_realign_batch (the local mode itself, `load_read_segments` read_parsing.rs:47-113, is mirrored too);
  * the `global_disabled` switch (read_parsing.rs:597-600) — each record's WFA result is independent of the
    others, only the decision to USE it depends on the records before it;
  * qualities = 2 x base(type) (read_parsing.rs:803-835), ReadSegment::new, collapse per qname, the
    min_matched_alleles split (read_parsing.rs:611-629).
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _ffi
from .read_segments import AlleleType, BlockMatrix, ReadSegment
from .wfa_graph import BASE_QUAL, PreparedWfaBatch, VariantType, WfaJobSpec


@dataclass
class GlobalRealignmentConfig:
    """read_parsing.rs:25-34 (defaults: cli.rs:189,196,203,210)"""
    max_edit_distance: int = 500
    wfa_prune_distance: int = 500
    global_failure_ratio: float = 0.5
    global_failure_minimum: int = 50


CIGAR_OPS = "MIDNSHP=X"   # BAM op codes 0..8


@dataclass
class LocalRecord:
    """What local_realignment reads from one BAM record (read_parsing.rs:121-160)."""
    qname: str
    pos: int               # read.pos()
    cigar: list            # [(op, length)], op a BAM code 0..8 or its "MIDNSHP=X" letter
    seq: bytes             # read.seq().as_bytes()
    qual: bytes            # read.qual()


@dataclass
class AlignedRecord:
    """What global_realignment needs from one BAM record (read_parsing.rs:672-742)."""
    qname: str
    min_position: int      # first reference base of the alignment
    max_position: int      # last reference base of the alignment (inclusive)
    read_align: bytes      # seq[read_start..=read_end]
    local: LocalRecord = None   # CIGAR view of the same record, needed only if it falls back to local mode


@dataclass
class LoadStats:
    num_reads: int = 0
    skipped_reads: int = 0
    global_aligned: int = 0
    local_aligned: int = 0
    edit_distances: list = field(default_factory=list)


def local_realignment_batch(records, variant_calls, device_id=0):
    """`local_realignment(read, variant_calls)` (read_parsing.rs:121-503) for a list of LocalRecord through
    hp_local_realign_batch: returns (alleles[R, N] u8, quals[R, N] u8, [ReadStats])."""
    dll = _ffi.lib()
    n_r, n_v = len(records), len(variant_calls)
    keep = []

    def u8(buf):
        a = np.frombuffer(bytes(buf), np.uint8) if len(buf) else np.zeros(1, np.uint8)
        keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_uint8))

    vs = (_ffi.LocalVariant * max(n_v, 1))()
    for i, v in enumerate(variant_calls):
        a0, a1 = v.get_allele0(), v.get_allele1()
        vs[i].position, vs[i].ref_len, vs[i].variant_type = v.position, v.ref_len, int(v.variant_type)
        vs[i].prefix_len, vs[i].postfix_len = v.prefix_len, v.postfix_len
        vs[i].allele0, vs[i].allele1, vs[i].allele0_len, vs[i].allele1_len = u8(a0), u8(a1), len(a0), len(a1)
        vs[i].flags = 1 if v.is_ignored else 0
    rs = (_ffi.LocalRead * max(n_r, 1))()
    for i, r in enumerate(records):
        if len(r.seq) != len(r.qual):
            raise ValueError("sequence and quality lengths differ")   # read_parsing.rs:155 assert_eq!
        cg = np.array([(int(n) << 4) | (CIGAR_OPS.index(op) if isinstance(op, str) else int(op)) for op, n in r.cigar] or [0],
                      np.uint32)
        keep.append(cg)
        rs[i].pos, rs[i].cigar, rs[i].n_cigar = r.pos, cg.ctypes.data_as(C.POINTER(C.c_uint32)), len(r.cigar)
        rs[i].seq_len, rs[i].seq, rs[i].qual = len(r.seq), u8(r.seq), u8(r.qual)
    alleles = np.zeros((n_r, max(n_v, 1)), np.uint8)
    quals = np.zeros((n_r, max(n_v, 1)), np.uint8)
    stats = (_ffi.ReadStats * max(n_r, 1))()
    _ffi.check(dll.hp_local_realign_batch(rs, n_r, vs, n_v, alleles.ctypes.data, quals.ctypes.data, stats, device_id))
    return alleles[:, :n_v], quals[:, :n_v], list(stats)[:n_r]


def load_read_segments(records, variant_calls, min_matched_alleles=2, device_id=0):
    """`load_read_segments` (read_parsing.rs:47-113, --disable-global-realignment) over decoded records: one
    hp_local_realign_batch call per block, then ReadSegment::new, collapse per qname and the min_matched_alleles
    split. Returns (read_segments, phasable_segments, LoadStats, [ReadStats per record])."""
    alleles, quals, rstats = local_realignment_batch(records, variant_calls, device_id)
    stats = LoadStats()
    groups = {}
    for i, rec in enumerate(records):
        if rstats[i].skipped_reads == 0:
            groups.setdefault(rec.qname, []).append(ReadSegment(rec.qname, alleles[i].tolist(), quals[i].tolist()))
            stats.local_aligned += 1
        else:
            stats.skipped_reads += 1
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
    return read_segments, phasable, stats, rstats


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
                            config=None, device_id=0):
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

    # The tail is order-dependent (read_parsing.rs:556-605): a record goes to local re-alignment when its WFA hit
    # max_edit_distance or once `global_disabled` has flipped, and the counters behind that switch only advance for
    # records that were not skipped. A record's local result does not depend on any other record, so the records
    # that need one are solved in at most two hp_local_realign_batch calls: the WFA failures up front, and everything
    # after the flip the first time the replay needs it.
    local_rows = {}

    def solve_local(indices):
        indices = [i for i in indices if i not in local_rows]
        missing = [i for i in indices if records[i].local is None]
        if missing:
            raise NotImplementedError(f"record {missing[0]} needs local re-alignment (read_parsing.rs:121-503), which "
                                      "needs its CIGAR: set AlignedRecord.local")
        if indices:
            la, lq, ls = local_realignment_batch([records[i].local for i in indices], variant_calls, device_id)
            for k, i in enumerate(indices):
                local_rows[i] = (la[k], lq[k], ls[k])

    solve_local([i for i, m in enumerate(meta) if m is not None and results[m[0]][0] != 0])
    stats = LoadStats()
    groups = {}
    global_disabled = False
    num_global_failures = 0.0
    total_parsed = 0.0
    for idx, (rec, m) in enumerate(zip(records, meta)):
        if m is None:
            stats.skipped_reads += 1   # no variant inside the alignment: both modes skip it (:703-712; local: no overlaps)
            continue
        j, first, last = m
        status, score, _, al = results[j]
        if global_disabled or status != 0:
            if idx not in local_rows:
                solve_local([i for i in range(idx, len(records)) if meta[i] is not None])
            alleles, quals, ls = local_rows[idx]
            wfa_score = config.max_edit_distance
            skipped, local_aligned = ls.skipped_reads == 1, 1.0
        else:
            alleles = np.full(n_var, int(AlleleType.NoOverlap), np.uint8)
            alleles[first:last] = al
            quals = np.zeros(n_var, np.uint8)
            for i in range(first, last):
                if alleles[i] < 2:
                    quals[i] = 2 * BASE_QUAL[VariantType(variant_calls[i].variant_type)]
            wfa_score = score
            skipped, local_aligned = False, 0.0
        if skipped:
            stats.skipped_reads += 1
            continue
        stats.local_aligned += int(local_aligned)
        stats.global_aligned += 1 - int(local_aligned)
        groups.setdefault(rec.qname, []).append(ReadSegment(rec.qname, np.asarray(alleles).tolist(), np.asarray(quals).tolist()))
        stats.edit_distances.append(int(wfa_score))
        num_global_failures += local_aligned
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
