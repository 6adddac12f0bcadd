"""`.hpbk` phase-block capture / replay format.

BASELINE.json configs[2-5] (real HG002 blocks) can only ever run here from blocks captured on a machine that
has Rust + htslib + the data: a patched HiPhase dumps, at reference src/phaser.rs:541 (just before
`astar_solver`), the solver's exact input and — after the call — its output. This module reads and writes that
file so captured blocks can be replayed through hp_astar_solve* and compared bit-for-bit with the real binary.

Layout (little endian), one record per block, records simply concatenated:
  magic    8 B   b"HPBK0001"
  header   10 x u64: block_index, n_variants N, n_reads R, n_cells, min_queue_size, queue_increment,
                     has_expected (0/1), reserved x 3
  read_start u32[R], read_end u32[R], row_off u64[R+1]
  alleles_2bit u8[ceil(n_cells/4)], quals u8[n_cells], var_flags u8[N]
  if has_expected: h1 u8[N], h2 u8[N], stats u64[7] (PhaseStats field order of phase_stats.rs:131-147)
  every array is padded with zeros to a multiple of 8 bytes.
"""
import struct

import numpy as np

from .read_segments import BlockMatrix

MAGIC = b"HPBK0001"


def _pad(f, nbytes):
    f.write(b"\0" * ((-nbytes) % 8))


def _wr(f, arr):
    b = np.ascontiguousarray(arr).tobytes()
    f.write(b)
    _pad(f, len(b))


def write_block(f, block: BlockMatrix, block_index=0, min_queue_size=1000, queue_increment=3, expected=None):
    """expected = (h1, h2, stats tuple of 7) or None."""
    n_cells = block.n_cells
    f.write(MAGIC)
    f.write(struct.pack("<10Q", block_index, block.n_variants, block.n_reads, n_cells, min_queue_size, queue_increment,
                        1 if expected is not None else 0, 0, 0, 0))
    _wr(f, block.read_start.astype("<u4"))
    _wr(f, block.read_end.astype("<u4"))
    _wr(f, block.row_off.astype("<u8"))
    _wr(f, block.alleles_2bit[: (n_cells + 3) // 4])
    _wr(f, block.quals[:n_cells])
    _wr(f, block.var_flags)
    if expected is not None:
        h1, h2, stats = expected
        _wr(f, np.asarray(h1, np.uint8))
        _wr(f, np.asarray(h2, np.uint8))
        f.write(struct.pack("<7Q", *[int(x) for x in stats]))


def _rd(f, dtype, count):
    nbytes = np.dtype(dtype).itemsize * count
    raw = f.read(nbytes)
    if len(raw) != nbytes:
        raise EOFError("truncated .hpbk record")
    f.read((-nbytes) % 8)
    return np.frombuffer(raw, dtype=dtype, count=count).copy()


def read_blocks(f):
    """Yields (BlockMatrix, meta dict, expected or None)."""
    while True:
        magic = f.read(8)
        if not magic:
            return
        if magic != MAGIC:
            raise ValueError("not an .hpbk stream")
        hdr = struct.unpack("<10Q", f.read(80))
        block_index, n, r, n_cells, minq, qinc, has_exp = hdr[:7]
        rs = _rd(f, "<u4", r)
        re = _rd(f, "<u4", r)
        off = _rd(f, "<u8", r + 1)
        a2 = _rd(f, "u1", (n_cells + 3) // 4)
        q = _rd(f, "u1", n_cells)
        fl = _rd(f, "u1", n)
        if a2.size == 0:
            a2 = np.zeros(1, np.uint8)
        if q.size == 0:
            q = np.zeros(1, np.uint8)
        exp = None
        if has_exp:
            h1 = _rd(f, "u1", n)
            h2 = _rd(f, "u1", n)
            stats = struct.unpack("<7Q", f.read(56))
            exp = (h1, h2, stats)
        yield BlockMatrix(n, rs, re, off, a2, q, fl), {"block_index": block_index, "min_queue_size": minq,
                                                       "queue_increment": qinc}, exp
