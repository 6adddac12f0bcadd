"""Whole phase blocks through ONE C entry (include/hiphase_gpu.h `hp_solve_blocks` / `hp_blockset_*`): the mirror of
reference src/phaser.rs:513-630 from the decoded records on, for any number of blocks per call. The library runs
graph-WFA over the records of all blocks in one device batch, the local fallback and the `global_disabled` replay,
quality assignment, collapse, the A* solver and the post-processing; this module only marshals."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _ffi
from .read_parsing import CIGAR_OPS, GlobalRealignmentConfig
from .wfa_graph import VariantType

U64_MAX = 2 ** 64 - 1


@dataclass
class BlockSpec:
    """One phase block: what `solve_block` has once variants and records are loaded (phaser.rs:440-533)."""
    block_index: int
    reference: bytes
    variant_calls: list          # hiphase_amd.wfa_graph.Variant, position-sorted
    hom_calls: list
    records: list                # hiphase_amd.read_parsing.AlignedRecord (local=LocalRecord for fallbacks / local mode)
    ref_base: int = 0


@dataclass
class BlockResult:
    haplotype_1: np.ndarray
    haplotype_2: np.ndarray
    statistics: tuple
    span_counts: np.ndarray
    segments: list               # (qname, start, end, alleles list, quals list, in_solver)
    haplotags: dict              # qname -> (first_het, haplotag)
    num_reads: int = 0
    skipped_reads: int = 0
    global_aligned: int = 0
    local_aligned: int = 0
    edit_distances: list = field(default_factory=list)
    read_stats: tuple = ()       # (num_alleles, exact, inexact, failed, allele0, allele1 _matches[11]): the rest of ReadStats
    status: int = 0             # 0, or 2 = HP_BLOCK_UNSUPPORTED (outside the device solver's limits: the caller solves it)


_BAM4_CODE = np.full(256, 15, np.uint8)          # htslib's seq_nt16_table: anything that is not an IUPAC code -> N
for _i, _c in enumerate(b"=ACMGRSVTWYHKDBN"):
    _BAM4_CODE[_c] = _i


def pack_bam4(seq, offset=0):
    """bytes -> the BAM record encoding of the bases (4 bits per base, high nibble first), after `offset` filler bases: what
    rust-htslib's `read.seq().encoded` holds. Only IUPAC upper-case bases survive the BAM encoding (as in a real BAM)."""
    codes = _BAM4_CODE[np.frombuffer(bytes(seq), np.uint8)] if len(seq) else np.zeros(0, np.uint8)
    codes = np.concatenate([np.full(offset, 15, np.uint8), codes, np.zeros((offset + len(codes)) & 1, np.uint8)])
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)


class _Marshalled:
    """ctypes views of a list of BlockSpec (keeps every buffer alive). seq_format = _ffi.SEQ_BAM4 hands the reads over in the
    BAM's own 4-bit encoding (every other record behind an odd number of filler bases, as a clipped record's read_start is)."""

    def __init__(self, blocks, need_local, seq_format=_ffi.SEQ_ASCII):
        self.keep = []
        self.seq_format = seq_format
        self.n = len(blocks)
        self.inputs = (_ffi.BlockInput * max(self.n, 1))()
        self.qnames = []
        for b, blk in enumerate(blocks):
            self._block(b, blk, need_local)

    def _u8(self, buf):
        a = np.frombuffer(bytes(buf), np.uint8) if len(buf) else np.zeros(1, np.uint8)
        self.keep.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_uint8))

    def _wfa_variants(self, vs):
        arr = (_ffi.WfaVariant * max(len(vs), 1))()
        for i, v in enumerate(vs):
            arr[i].position, arr[i].ref_len = v.position, v.ref_len
            arr[i].flags = (1 if v.is_ignored else 0) | (2 if v.index_allele0 != 0 else 0)
            arr[i].allele0, arr[i].allele0_len = self._u8(v.allele0), len(v.allele0)
            arr[i].allele1, arr[i].allele1_len = self._u8(v.allele1), len(v.allele1)
        self.keep.append(arr)
        return arr

    def _block(self, b, blk, need_local):
        I = self.inputs[b]
        hets, homs = blk.variant_calls, blk.hom_calls
        I.block_index, I.ref_base = blk.block_index, blk.ref_base
        I.reference = self._u8(blk.reference)
        I.n_hets, I.n_homs, I.n_records = len(hets), len(homs), len(blk.records)
        I.hets = self._wfa_variants(hets)
        I.homs = self._wfa_variants(homs)
        types = np.asarray([int(v.variant_type) for v in hets] or [0], np.uint8)
        self.keep.append(types)
        I.het_types = types.ctypes.data_as(C.POINTER(C.c_uint8))
        if need_local or any(getattr(r, "local", None) is not None for r in blk.records):
            lv = (_ffi.LocalVariant * max(len(hets), 1))()
            for i, v in enumerate(hets):
                a0, a1 = v.get_allele0(), v.get_allele1()
                lv[i].position, lv[i].ref_len, lv[i].variant_type = v.position, v.ref_len, int(v.variant_type)
                lv[i].prefix_len, lv[i].postfix_len = v.prefix_len, v.postfix_len
                lv[i].allele0, lv[i].allele1, lv[i].allele0_len, lv[i].allele1_len = self._u8(a0), self._u8(a1), len(a0), len(a1)
                lv[i].flags = 1 if v.is_ignored else 0
            self.keep.append(lv)
            I.local_hets = lv
        ids, names = {}, []
        recs = (_ffi.BlockRecord * max(len(blk.records), 1))()
        locs = (_ffi.LocalRead * max(len(blk.records), 1))()
        for i, r in enumerate(blk.records):
            if r.qname not in ids:
                ids[r.qname] = len(names)
                names.append(r.qname)
            loc = getattr(r, "local", None)
            if hasattr(r, "min_position"):
                recs[i].min_position, recs[i].max_position = r.min_position, r.max_position
                if self.seq_format == _ffi.SEQ_BAM4:
                    off = (i * 7) % 5   # 0, 2, 4, 1, 3: even and odd read_start
                    recs[i].read_align, recs[i].read_len, recs[i].read_offset = self._u8(pack_bam4(r.read_align, off).tobytes()), len(r.read_align), off
                else:
                    recs[i].read_align, recs[i].read_len = self._u8(r.read_align), len(r.read_align)
            else:           # a LocalRecord on its own (local mode)
                loc = r
                recs[i].min_position = recs[i].max_position = r.pos
                recs[i].read_align, recs[i].read_len = self._u8(b""), 0
            recs[i].qname_id = ids[r.qname]
            if loc is not None:
                cg = np.array([(int(n) << 4) | (CIGAR_OPS.index(op) if isinstance(op, str) else int(op)) for op, n in loc.cigar] or [0], np.uint32)
                self.keep.append(cg)
                locs[i].pos, locs[i].cigar, locs[i].n_cigar = loc.pos, cg.ctypes.data_as(C.POINTER(C.c_uint32)), len(loc.cigar)
                locs[i].seq_len, locs[i].qual = len(loc.seq), self._u8(loc.qual)
                if self.seq_format == _ffi.SEQ_BAM4:
                    locs[i].seq, locs[i].seq_format = self._u8(pack_bam4(loc.seq).tobytes()), _ffi.SEQ_BAM4
                else:
                    locs[i].seq = self._u8(loc.seq)
                recs[i].local = C.pointer(locs[i])
        self.keep += [recs, locs]
        I.records, I.n_qnames, I.seq_format = recs, len(names), self.seq_format
        self.qnames.append(names)


class _Outputs:
    def __init__(self, m):
        self.arr = (_ffi.BlockOutput * max(m.n, 1))()
        self.bufs = []
        for b in range(m.n):
            I, O = m.inputs[b], self.arr[b]
            n, q = I.n_hets, max(I.n_qnames, 1)
            d = dict(h1=np.zeros(n, np.uint8), h2=np.zeros(n, np.uint8), span=np.zeros(max(n - 1, 1), np.uint64),
                     qn=np.zeros(q, np.uint32), st=np.zeros(q, np.uint32), en=np.zeros(q, np.uint32), so=np.zeros(q, np.uint8),
                     ht=np.zeros(q, np.uint8), fh=np.zeros(q, np.uint32), ro=np.zeros(q + 1, np.uint64),
                     al=np.zeros(q * n + 1, np.uint8), ql=np.zeros(q * n + 1, np.uint8), ed=np.zeros(max(I.n_records, 1), np.uint64))
            self.bufs.append(d)
            p8, p32, p64 = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
            O.h1, O.h2, O.span_counts = d["h1"].ctypes.data_as(p8), d["h2"].ctypes.data_as(p8), d["span"].ctypes.data_as(p64)
            O.seg_qname, O.seg_start, O.seg_end = d["qn"].ctypes.data_as(p32), d["st"].ctypes.data_as(p32), d["en"].ctypes.data_as(p32)
            O.seg_solver, O.seg_haplotag, O.seg_first_het = d["so"].ctypes.data_as(p8), d["ht"].ctypes.data_as(p8), d["fh"].ctypes.data_as(p32)
            O.seg_row_off, O.seg_alleles, O.seg_quals = d["ro"].ctypes.data_as(p64), d["al"].ctypes.data_as(p8), d["ql"].ctypes.data_as(p8)
            O.seg_cell_cap = q * n + 1
            O.edit_distances = d["ed"].ctypes.data_as(p64)

    def results(self, m):
        out = []
        for b in range(m.n):
            O, d, names = self.arr[b], self.bufs[b], m.qnames[b]
            n = m.inputs[b].n_hets
            segs, tags = [], {}
            for k in range(O.n_segments):
                a, e = int(d["ro"][k]), int(d["ro"][k + 1])
                q = names[int(d["qn"][k])]
                segs.append((q, int(d["st"][k]), int(d["en"][k]), d["al"][a:e].tolist(), d["ql"][a:e].tolist(), bool(d["so"][k])))
                if d["ht"][k] != 2:
                    tags[q] = (int(d["fh"][k]), int(d["ht"][k]))
            out.append(BlockResult(d["h1"].copy(), d["h2"].copy(), O.stats.as_tuple(), d["span"][:max(n - 1, 0)].copy(), segs, tags,
                                   int(O.num_reads), int(O.skipped_reads), int(O.global_aligned), int(O.local_aligned),
                                   d["ed"][:O.n_edit_distances].tolist(), O.read_stats(), int(O.status)))
        return out


def _params(min_matched_alleles, min_queue_size, queue_increment, config, global_realignment):
    config = config or GlobalRealignmentConfig()
    p = _ffi.BlockParams()
    p.astar.min_queue_size, p.astar.queue_increment = min_queue_size, queue_increment
    p.wfa_prune_distance = U64_MAX if config.wfa_prune_distance in (0, None) else config.wfa_prune_distance
    p.max_edit_distance = config.max_edit_distance
    p.global_failure_ratio, p.global_failure_minimum = config.global_failure_ratio, config.global_failure_minimum
    p.min_matched_alleles = min_matched_alleles
    p.global_realignment = 1 if global_realignment else 0
    return p


def solve_blocks(blocks, min_matched_alleles=2, min_queue_size=1000, queue_increment=3, config=None, global_realignment=True,
                 device_id=-1, seq_format=_ffi.SEQ_ASCII):
    """hp_solve_blocks for a list of BlockSpec -> [BlockResult]."""
    m = _Marshalled(blocks, need_local=not global_realignment, seq_format=seq_format)
    o = _Outputs(m)
    p = _params(min_matched_alleles, min_queue_size, queue_increment, config, global_realignment)
    _ffi.check(_ffi.lib().hp_solve_blocks(m.n, m.inputs, C.byref(p), o.arr, device_id))
    return o.results(m)


class BlockSet:
    """Resident form (hp_blockset_*): sequences uploaded once, `solve()` any number of times."""

    def __init__(self, blocks, min_matched_alleles=2, min_queue_size=1000, queue_increment=3, config=None,
                 global_realignment=True, device_id=-1, seq_format=_ffi.SEQ_ASCII):
        self.m = _Marshalled(blocks, need_local=not global_realignment, seq_format=seq_format)
        self.o = _Outputs(self.m)
        self.p = _params(min_matched_alleles, min_queue_size, queue_increment, config, global_realignment)
        st = C.c_int(0)
        self.h = _ffi.lib().hp_blockset_create(self.m.n, self.m.inputs, C.byref(self.p), device_id, C.byref(st))
        if not self.h:
            raise _ffi.HpError(st.value, _ffi.lib().hp_last_error().decode())
        self.stage_ms = (C.c_double * 8)()

    def solve(self):
        _ffi.check(_ffi.lib().hp_blockset_solve(self.h, self.o.arr, self.stage_ms))
        return list(self.stage_ms)

    def results(self):
        return self.o.results(self.m)

    def work(self):
        """hp_blockset_work of the last solve -> dict"""
        w = (C.c_uint64 * 8)()
        _ffi.check(_ffi.lib().hp_blockset_work(self.h, w))
        return dict(zip(("wfa_reads", "wfa_read_bytes", "wfa_node_bytes", "wfa_updates", "astar_cells", "astar_evals", "hets", "rows"), list(w)))

    def close(self):
        if self.h:
            _ffi.lib().hp_blockset_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class BlockStream:
    """Pipelined form (hp_blockstream_*): `submit` a list of BlockSpec, `wait` for its results; sets complete in order."""

    def __init__(self, min_matched_alleles=2, min_queue_size=1000, queue_increment=3, config=None, global_realignment=True,
                 device_id=-1, depth=0, seq_format=_ffi.SEQ_ASCII):
        self.p = _params(min_matched_alleles, min_queue_size, queue_increment, config, global_realignment)
        self.need_local, self.seq_format = not global_realignment, seq_format
        st = C.c_int(0)
        self.h = _ffi.lib().hp_blockstream_create(C.byref(self.p), device_id, depth, C.byref(st))
        if not self.h:
            raise _ffi.HpError(st.value, _ffi.lib().hp_last_error().decode())
        self.inflight = {}

    def submit(self, blocks):
        m = _Marshalled(blocks, need_local=self.need_local, seq_format=self.seq_format)
        o = _Outputs(m)
        t = C.c_uint64(0)
        _ffi.check(_ffi.lib().hp_blockstream_submit(self.h, m.n, m.inputs, o.arr, C.byref(t)))
        self.inflight[t.value] = (m, o)
        return t.value

    def wait(self, ticket):
        """-> ([BlockResult], stage_ms[16], work[8])"""
        m, o = self.inflight.pop(ticket)
        ms, work = (C.c_double * 16)(), (C.c_uint64 * 8)()
        _ffi.check(_ffi.lib().hp_blockstream_wait(self.h, ticket, ms, work))
        return o.results(m), list(ms), list(work)

    def close(self):
        if self.h:
            _ffi.lib().hp_blockstream_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
