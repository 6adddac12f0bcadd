"""ctypes loader for oracle/liboracle.so — the CPU restatement of the reference (TEST INFRASTRUCTURE).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

from hiphase_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def oracle():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_PATH):
        build_oracle()
    d = C.CDLL(ORACLE_PATH)
    _ffi.declare_common(d)
    u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    d.hpo_read_segment_new.restype = None
    d.hpo_read_segment_new.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    d.hpo_read_segment_collapse.restype = C.c_int
    d.hpo_read_segment_collapse.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    d.hpo_score_partial_haplotype.restype = C.c_uint64
    d.hpo_score_partial_haplotype.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.c_size_t]
    d.hpo_astar_node_walk.restype = C.c_int
    d.hpo_astar_node_walk.argtypes = [C.POINTER(_ffi.BlockView), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    d.hpo_hap_tracker_script.restype = C.c_int
    d.hpo_hap_tracker_script.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    d.hpo_astar_heuristic.restype = C.c_int
    d.hpo_astar_heuristic.argtypes = [C.POINTER(_ffi.BlockView), C.POINTER(_ffi.AstarParams), C.c_void_p]
    d.hpo_astar_solve.restype = C.c_int
    d.hpo_astar_solve.argtypes = [C.POINTER(_ffi.BlockView), C.POINTER(_ffi.AstarParams), C.c_void_p, C.c_void_p,
                                  C.POINTER(_ffi.PhaseStats), C.POINTER(_ffi.WorkCounters), C.c_void_p]
    d.hpo_bruteforce_mec.restype = C.c_uint64
    d.hpo_bruteforce_mec.argtypes = [C.POINTER(_ffi.BlockView)]
    d.hpo_solution_span_counts.restype = C.c_int
    d.hpo_solution_span_counts.argtypes = [C.POINTER(_ffi.BlockView), C.c_void_p, C.c_void_p, C.c_void_p]
    d.hpo_haplotag_reads.restype = C.c_int
    d.hpo_haplotag_reads.argtypes = [C.POINTER(_ffi.BlockView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    d.hpo_edit_distance.restype = C.c_uint64
    d.hpo_edit_distance.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    d.hpo_match_allele.restype = C.c_int
    d.hpo_match_allele.argtypes = [C.POINTER(_ffi.LocalVariant), C.c_void_p, C.c_size_t]
    d.hpo_closest_allele_clip.restype = C.c_int
    d.hpo_closest_allele_clip.argtypes = [C.POINTER(_ffi.LocalVariant), C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                          u64p, u64p]
    d.hpo_local_realignment.restype = C.c_int
    d.hpo_local_realignment.argtypes = [C.POINTER(_ffi.LocalRead), C.POINTER(_ffi.LocalVariant), C.c_size_t, C.c_void_p,
                                        C.c_void_p, C.POINTER(_ffi.ReadStats)]
    d.hpo_graph_new.restype = C.c_void_p
    d.hpo_graph_new.argtypes = [C.c_uint64]
    d.hpo_graph_free.restype = None
    d.hpo_graph_free.argtypes = [C.c_void_p]
    d.hpo_graph_add_node.restype = C.c_int64
    d.hpo_graph_add_node.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    d.hpo_graph_from_job.restype = C.c_void_p
    d.hpo_graph_from_job.argtypes = [C.POINTER(_ffi.WfaJob), C.c_uint64, C.POINTER(C.c_int)]
    d.hpo_graph_num_nodes.restype = C.c_uint64
    d.hpo_graph_num_nodes.argtypes = [C.c_void_p]
    d.hpo_graph_node_alleles.restype = C.c_size_t
    d.hpo_graph_node_alleles.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t]
    for name in ("hpo_graph_node_seq", "hpo_graph_node_parents", "hpo_graph_node_edges"):
        f = getattr(d, name)
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
    d.hpo_graph_edit_distance.restype = C.c_int
    d.hpo_graph_edit_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, u64p,
                                          C.c_void_p, C.POINTER(C.c_size_t)]
    d.hpo_graph_bruteforce.restype = C.c_int
    d.hpo_graph_bruteforce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
    d.hpo_wfa_assign.restype = C.c_int
    d.hpo_wfa_assign.argtypes = [C.POINTER(_ffi.WfaJob), C.c_uint64, C.c_uint64, C.POINTER(_ffi.WfaResult), C.c_void_p]
    d.hpo_solve_block.restype = C.c_int
    d.hpo_solve_block.argtypes = [C.POINTER(_ffi.BlockInput), C.POINTER(_ffi.BlockParams), C.POINTER(_ffi.BlockOutput)]
    _lib = d
    return d


# ---- convenience wrappers -----------------------------------------------------------------------

def params(min_queue_size=1000, queue_increment=3, max_segment_size=40):
    return _ffi.AstarParams(min_queue_size, queue_increment, max_segment_size, 0)


def oracle_solve(block, min_queue_size=1000, queue_increment=3, max_segment_size=40, want_heuristics=False):
    """hpo_astar_solve -> (h1, h2, stats tuple, counters tuple[, heuristics])."""
    d = oracle()
    v = block.view()
    p = params(min_queue_size, queue_increment, max_segment_size)
    h1 = np.zeros(block.n_variants, np.uint8)
    h2 = np.zeros(block.n_variants, np.uint8)
    st = _ffi.PhaseStats()
    ctr = _ffi.WorkCounters()
    heur = np.zeros(block.n_variants + 1, np.uint64)
    rc = d.hpo_astar_solve(C.byref(v), C.byref(p), h1.ctypes.data, h2.ctypes.data, C.byref(st), C.byref(ctr),
                           heur.ctypes.data)
    if rc != 0:
        raise _ffi.HpError(rc, "oracle hpo_astar_solve")
    if want_heuristics:
        return h1, h2, st.as_tuple(), ctr.as_tuple(), heur
    return h1, h2, st.as_tuple(), ctr.as_tuple()


def oracle_solve_blocks(blocks, min_matched_alleles=2, min_queue_size=1000, queue_increment=3, config=None, global_realignment=True,
                        seq_format=_ffi.SEQ_ASCII):
    """The whole path on the CPU oracle (hpo_solve_block, one block at a time) for a list of hiphase_amd.blocks.BlockSpec,
    marshalled exactly as hiphase_amd.blocks.solve_blocks marshals them -> [BlockResult]."""
    from hiphase_amd import blocks as B
    d = oracle()
    m = B._Marshalled(blocks, need_local=not global_realignment, seq_format=seq_format)
    o = B._Outputs(m)
    p = B._params(min_matched_alleles, min_queue_size, queue_increment, config, global_realignment)
    for b in range(m.n):
        rc = d.hpo_solve_block(C.byref(m.inputs[b]), C.byref(p), C.byref(o.arr[b]))
        if rc != 0:
            raise _ffi.HpError(rc, f"oracle hpo_solve_block, block {b}")
    return o.results(m)


def oracle_synth(n_variants, coverage, span, error_rate, ambig_rate, seed, ignored_permille=0):
    from hiphase_amd.read_segments import synth_block
    return synth_block(n_variants, coverage, span, error_rate, ambig_rate, seed, ignored_permille, dll=oracle())


class OracleGraph:
    """wfa_graph.rs WFAGraph restatement handle."""

    def __init__(self, handle=None, max_edit_distance=1000):
        self.d = oracle()
        self.h = handle if handle is not None else self.d.hpo_graph_new(max_edit_distance)

    def add_node(self, seq, parents):
        s = np.asarray(list(seq), dtype=np.uint8)
        p = np.asarray(list(parents), dtype=np.uint64)
        return self.d.hpo_graph_add_node(self.h, s.ctypes.data if s.size else None, s.size,
                                         p.ctypes.data if p.size else None, p.size)

    def num_nodes(self):
        return self.d.hpo_graph_num_nodes(self.h)

    def node_alleles(self, node):
        vi = np.zeros(16, np.uint64)
        al = np.zeros(16, np.uint8)
        n = self.d.hpo_graph_node_alleles(self.h, node, vi.ctypes.data, al.ctypes.data, 16)
        return [(int(vi[i]), int(al[i])) for i in range(n)]

    def node_seq(self, node):
        buf = np.zeros(1 << 16, np.uint8)
        n = self.d.hpo_graph_node_seq(self.h, node, buf.ctypes.data, buf.size)
        return bytes(buf[:n])

    def node_parents(self, node):
        buf = np.zeros(64, np.uint64)
        n = self.d.hpo_graph_node_parents(self.h, node, buf.ctypes.data, 64)
        return [int(x) for x in buf[:n]]

    def node_edges(self, node):
        buf = np.zeros(64, np.uint64)
        n = self.d.hpo_graph_node_edges(self.h, node, buf.ctypes.data, 64)
        return [int(x) for x in buf[:n]]

    def edit_distance(self, other, prune_distance=2 ** 64 - 1, shuffle_seed=0):
        o = np.asarray(list(other), dtype=np.uint8)
        score = C.c_uint64(0)
        trav = np.zeros(max(1, self.num_nodes()), np.uint64)
        n = C.c_size_t(trav.size)
        st = self.d.hpo_graph_edit_distance(self.h, o.ctypes.data if o.size else None, o.size, prune_distance,
                                            shuffle_seed, C.byref(score), trav.ctypes.data, C.byref(n))
        return st, int(score.value), [int(x) for x in trav[:n.value]]

    def bruteforce(self, other, wfa_nodes=()):
        """(min Levenshtein over every root -> last-node path, paths, optimal paths, union of the optimal paths' nodes, whether
        some optimal path lies inside wfa_nodes) - hp_oracle_brute.cpp, nothing shared with the wavefront code"""
        o = np.asarray(list(other), dtype=np.uint8)
        out = np.zeros(5, np.uint64)
        mask = 0
        for n in wfa_nodes:
            mask |= 1 << n
        rc = self.d.hpo_graph_bruteforce(self.h, o.ctypes.data if o.size else None, o.size, mask, out.ctypes.data)
        assert rc == 0, rc
        return int(out[0]), int(out[1]), int(out[2]), [k for k in range(64) if (int(out[3]) >> k) & 1], bool(out[4])

    def __del__(self):
        try:
            self.d.hpo_graph_free(self.h)
        except Exception:
            pass
