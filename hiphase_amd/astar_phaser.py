"""Host-side mirror of reference src/astar_phaser.rs:408-429 — `astar_solver` / `AstarResult` — over
the C ABI (hp_astar_solve*, hp_batch_*). The solve itself runs in HIP on the MI355X; nothing here computes."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from .read_segments import BlockMatrix


@dataclass
class AstarResult:
    """astar_phaser.rs:408-415"""
    haplotype_1: np.ndarray
    haplotype_2: np.ndarray
    statistics: _ffi.PhaseStats


def _params(min_queue_size, queue_increment, max_segment_size=40, block_index=0):
    return _ffi.AstarParams(min_queue_size, queue_increment, max_segment_size, block_index)


def astar_solver(phase_block_index, block: BlockMatrix, min_queue_size=1000, queue_increment=3,
                 max_segment_size=40) -> AstarResult:
    """astar_solver(phase_block, variants, read_segments, min_queue_size, queue_increment)
    (astar_phaser.rs:426-429). `block` carries the variants' flags and the read-segment matrix."""
    dll = _ffi.lib()
    p = _params(min_queue_size, queue_increment, max_segment_size, phase_block_index)
    h1 = np.zeros(block.n_variants, np.uint8)
    h2 = np.zeros(block.n_variants, np.uint8)
    st = _ffi.PhaseStats()
    v = block.view()
    _ffi.check(dll.hp_astar_solve(C.byref(v), C.byref(p), h1.ctypes.data, h2.ctypes.data, C.byref(st)))
    return AstarResult(h1, h2, st)


def astar_solve_batch(blocks, min_queue_size=1000, queue_increment=3, max_segment_size=40, device_id=0):
    """hp_astar_solve_batch: independent blocks; device_id=-1 shards over all visible GPUs."""
    dll = _ffi.lib()
    n = len(blocks)
    views = (_ffi.BlockView * n)(*[b.view() for b in blocks])
    p = _params(min_queue_size, queue_increment, max_segment_size)
    h1 = [np.zeros(b.n_variants, np.uint8) for b in blocks]
    h2 = [np.zeros(b.n_variants, np.uint8) for b in blocks]
    p1 = (C.c_void_p * n)(*[a.ctypes.data for a in h1])
    p2 = (C.c_void_p * n)(*[a.ctypes.data for a in h2])
    stats = (_ffi.PhaseStats * n)()
    _ffi.check(dll.hp_astar_solve_batch(n, views, C.byref(p), p1, p2, stats, device_id))
    return [AstarResult(h1[i], h2[i], stats[i]) for i in range(n)]


class ResidentBatch:
    """hp_batch_*: pack + upload once, solve many times with inputs resident in HBM (bench.py)."""

    def __init__(self, blocks, min_queue_size=1000, queue_increment=3, max_segment_size=40, device_id=0):
        dll = _ffi.lib()
        self._dll = dll
        self.n_vars = [b.n_variants for b in blocks]
        self.n_reads = [b.n_reads for b in blocks]
        n = len(blocks)
        views = (_ffi.BlockView * n)(*[b.view() for b in blocks])
        p = _params(min_queue_size, queue_increment, max_segment_size)
        status = C.c_int(0)
        self._h = dll.hp_batch_create(n, views, C.byref(p), device_id, C.byref(status))
        if not self._h:
            raise _ffi.HpError(status.value, dll.hp_last_error().decode())
        self.n_blocks = n

    def solve(self, stream=None):
        """Returns the HIP-event time of the solve kernel(s) in ms."""
        ms = C.c_float(0.0)
        _ffi.check(self._dll.hp_batch_solve(self._h, C.c_void_p(stream or 0), C.byref(ms)))
        return ms.value

    def results(self, want_heuristics=False):
        tot = sum(self.n_vars)
        h1 = np.zeros(tot, np.uint8)
        h2 = np.zeros(tot, np.uint8)
        stats = (_ffi.PhaseStats * self.n_blocks)()
        ctr = (_ffi.WorkCounters * self.n_blocks)()
        heur = np.zeros(tot + self.n_blocks, np.uint64) if want_heuristics else None
        _ffi.check(self._dll.hp_batch_results(self._h, h1.ctypes.data, h2.ctypes.data, stats, ctr,
                                              heur.ctypes.data if want_heuristics else None))
        offs = np.concatenate([[0], np.cumsum(self.n_vars)])
        res = [AstarResult(h1[offs[i]:offs[i + 1]], h2[offs[i]:offs[i + 1]], stats[i]) for i in range(self.n_blocks)]
        hs = None
        if want_heuristics:
            ho = np.concatenate([[0], np.cumsum([n + 1 for n in self.n_vars])])
            hs = [heur[ho[i]:ho[i + 1]] for i in range(self.n_blocks)]
        return res, list(ctr), hs

    def postprocess(self):
        """hp_batch_postprocess -> per block (span_counts[N-1], haplotag[R], first_het[R]) in caller row order."""
        nj = sum(max(n - 1, 0) for n in self.n_vars)
        nr = sum(self.n_reads)
        spans = np.zeros(max(nj, 1), np.uint64)
        ht = np.zeros(max(nr, 1), np.uint8)
        fh = np.zeros(max(nr, 1), np.uint32)
        _ffi.check(self._dll.hp_batch_postprocess(self._h, spans.ctypes.data, ht.ctypes.data, fh.ctypes.data))
        out, jo, ro = [], 0, 0
        for n, r in zip(self.n_vars, self.n_reads):
            out.append((spans[jo:jo + max(n - 1, 0)], ht[ro:ro + r], fh[ro:ro + r]))
            jo += max(n - 1, 0)
            ro += r
        return out

    def close(self):
        if self._h:
            self._dll.hp_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
