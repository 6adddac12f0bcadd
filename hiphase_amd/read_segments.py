"""Host-side mirror of reference src/data_types/read_segments.rs (matrix-row format) and the
CSR packing (`BlockMatrix`) that turns an interval tree of ReadSegments into an ``hp_block_view``.

This is data preparation only (it stays on the host in the reference too: read_parsing.rs builds
these rows, phaser.rs:514-533 hands them to the solver). Scoring on the solver path happens in HIP.
"""
import ctypes as C
import enum

import numpy as np

from . import _ffi


class AlleleType(enum.IntEnum):
    """read_segments.rs:5-16"""
    Reference = 0
    Alternate = 1
    Ambiguous = 2
    NoOverlap = 3


class ReadSegment:
    """read_segments.rs:19-62 — one clipped matrix row."""

    __slots__ = ("read_name", "alleles", "quals", "start", "end")

    def __init__(self, read_name, alleles, quals):
        alleles = [int(a) for a in alleles]
        quals = [int(q) for q in quals]
        assert len(alleles) == len(quals)
        n = len(alleles)
        first = next((i for i, a in enumerate(alleles) if a < AlleleType.Ambiguous), n)
        last = next((i + 1 for i in range(n - 1, -1, -1) if alleles[i] < AlleleType.Ambiguous), n)
        self.read_name = read_name
        self.start, self.end = first, last
        self.alleles = alleles[first:last]
        self.quals = quals[first:last]

    @classmethod
    def from_clipped(cls, read_name, start, end, alleles, quals):
        rs = cls.__new__(cls)
        rs.read_name, rs.start, rs.end = read_name, int(start), int(end)
        rs.alleles, rs.quals = [int(a) for a in alleles], [int(q) for q in quals]
        assert len(rs.alleles) == rs.end - rs.start == len(rs.quals)
        return rs

    def region(self):
        return range(self.start, self.end)

    def allele(self, index):  # read_segments.rs:128-134
        return self.alleles[index - self.start] if self.start <= index < self.end else int(AlleleType.NoOverlap)

    def qual(self, index):  # read_segments.rs:137-143
        return self.quals[index - self.start] if self.start <= index < self.end else 0

    def get_num_set(self):  # read_segments.rs:151-155
        return sum(1 for a in self.alleles if a < AlleleType.Ambiguous)

    @staticmethod
    def collapse(read_segments):  # read_segments.rs:71-121
        assert read_segments
        if len(read_segments) == 1:
            return read_segments[0]
        min_start = min(rs.start for rs in read_segments)
        max_end = max(rs.end for rs in read_segments)
        alleles = [int(AlleleType.NoOverlap)] * max_end
        quals = [0] * max_end
        for rs in read_segments:
            assert rs.read_name == read_segments[0].read_name
            for i in range(min_start, max_end):
                rsa, rsq = rs.allele(i), rs.qual(i)
                if rsa != AlleleType.NoOverlap:
                    if alleles[i] == AlleleType.NoOverlap:
                        alleles[i], quals[i] = rsa, rsq
                    elif alleles[i] == AlleleType.Ambiguous:
                        pass
                    elif alleles[i] == rsa:
                        quals[i] = max(quals[i], rsq)
                        assert quals[i] > 0
                    else:
                        alleles[i], quals[i] = int(AlleleType.Ambiguous), 0
        return ReadSegment(read_segments[0].read_name, alleles, quals)

    def __eq__(self, o):
        return (self.read_name, self.alleles, self.quals, self.start, self.end) == \
               (o.read_name, o.alleles, o.quals, o.start, o.end)

    def __repr__(self):
        return f"ReadSegment({self.read_name!r}, {self.start}..{self.end}, {self.alleles}, {self.quals})"


class BlockMatrix:
    """The read x variant allele matrix of one phase block in hp_block_view (CSR, 2-bit) layout."""

    def __init__(self, n_variants, read_start, read_end, row_off, alleles_2bit, quals, var_flags, names=None):
        self.n_variants = int(n_variants)
        self.read_start = np.ascontiguousarray(read_start, dtype=np.uint32)
        self.read_end = np.ascontiguousarray(read_end, dtype=np.uint32)
        self.row_off = np.ascontiguousarray(row_off, dtype=np.uint64)
        self.alleles_2bit = np.ascontiguousarray(alleles_2bit, dtype=np.uint8)
        self.quals = np.ascontiguousarray(quals, dtype=np.uint8)
        self.var_flags = np.ascontiguousarray(var_flags, dtype=np.uint8)
        self.names = names
        assert self.var_flags.shape[0] == self.n_variants
        assert self.row_off.shape[0] == self.n_reads + 1

    @property
    def n_reads(self):
        return int(self.read_start.shape[0])

    @property
    def n_cells(self):
        return int(self.row_off[-1])

    @classmethod
    def from_segments(cls, segments, n_variants, var_flags=None):
        """segments: iterable of ReadSegment (what the solver's IntervalTree holds)."""
        segments = list(segments)
        R = len(segments)
        rs_, re_ = np.zeros(R, np.uint32), np.zeros(R, np.uint32)
        off = np.zeros(R + 1, np.uint64)
        cells_a, cells_q = [], []
        for r, s in enumerate(segments):
            rs_[r], re_[r] = s.start, s.end
            off[r + 1] = off[r] + np.uint64(s.end - s.start)
            cells_a.extend(s.alleles)
            cells_q.extend(s.quals)
        a = np.asarray(cells_a, dtype=np.uint8)
        pad = (-len(a)) % 4
        a4 = np.concatenate([a, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        packed = (a4[:, 0] | (a4[:, 1] << 2) | (a4[:, 2] << 4) | (a4[:, 3] << 6)).astype(np.uint8)
        if packed.size == 0:
            packed = np.zeros(1, np.uint8)
        q = np.asarray(cells_q, dtype=np.uint8)
        if q.size == 0:
            q = np.zeros(1, np.uint8)
        if var_flags is None:
            var_flags = np.full(n_variants, _SNV, np.uint8)
        return cls(n_variants, rs_, re_, off, packed, q, var_flags, names=[s.read_name for s in segments])

    @classmethod
    def from_rows(cls, rows, var_flags=None):
        """rows: list of (alleles, quals) full block-length rows -> ReadSegment::new clipping."""
        segs = [ReadSegment(f"read_{i}", a, q) for i, (a, q) in enumerate(rows)]
        n = len(rows[0][0]) if rows else 0
        return cls.from_segments(segs, n, var_flags)

    def segments(self):
        out = []
        for r in range(self.n_reads):
            s, e, o = int(self.read_start[r]), int(self.read_end[r]), int(self.row_off[r])
            cells = np.arange(o, o + (e - s))
            al = (self.alleles_2bit[cells >> 2] >> (2 * (cells & 3))) & 3
            name = self.names[r] if self.names else f"read_{r}"
            out.append(ReadSegment.from_clipped(name, s, e, al.tolist(), self.quals[o:o + (e - s)].tolist()))
        return out

    def view(self):
        v = _ffi.BlockView()
        v.n_variants = self.n_variants
        v.n_reads = self.n_reads
        v.read_start = self.read_start.ctypes.data_as(C.POINTER(C.c_uint32))
        v.read_end = self.read_end.ctypes.data_as(C.POINTER(C.c_uint32))
        v.row_off = self.row_off.ctypes.data_as(C.POINTER(C.c_uint64))
        v.alleles_2bit = self.alleles_2bit.ctypes.data_as(C.POINTER(C.c_uint8))
        v.quals = self.quals.ctypes.data_as(C.POINTER(C.c_uint8))
        v.var_flags = self.var_flags.ctypes.data_as(C.POINTER(C.c_uint8))
        return v


_SNV = 0x2
_IGNORED = 0x1


def synth_block(n_variants, coverage, span, error_rate, ambig_rate, seed, ignored_permille=0, dll=None):
    """Deterministic synthetic block of SURVEY.md §8(d) (hp_synth_block). Returns (BlockMatrix, truth)."""
    dll = dll or _ffi.lib()
    spec = _ffi.SynthSpec(n_variants, coverage, span, ignored_permille, error_rate, ambig_rate, seed)
    n_cells = C.c_uint64(0)
    R = dll.hp_synth_block_size(C.byref(spec), C.byref(n_cells))
    rs_, re_ = np.zeros(R, np.uint32), np.zeros(R, np.uint32)
    off = np.zeros(R + 1, np.uint64)
    a2 = np.zeros((n_cells.value + 3) // 4 + 1, np.uint8)
    q = np.zeros(n_cells.value + 1, np.uint8)
    flags = np.zeros(n_variants, np.uint8)
    truth = np.zeros(n_variants, np.uint8)
    st = dll.hp_synth_block(C.byref(spec), rs_.ctypes.data, re_.ctypes.data, off.ctypes.data, a2.ctypes.data,
                            q.ctypes.data, flags.ctypes.data, truth.ctypes.data)
    if st != 0:
        raise _ffi.HpError(st, "hp_synth_block")
    return BlockMatrix(n_variants, rs_, re_, off, a2, q, flags), truth
