"""Host-side mirror of the graph-WFA seam (reference src/read_parsing.rs:769-800 calling
src/wfa_graph.rs:119-284,350-650) over hp_wfa_assign_batch, plus the minimal `Variant` view the
graph builder reads (reference src/data_types/variants.rs:67-94,549-591,649-660)."""
import ctypes as C
import enum
from dataclasses import dataclass, field

import numpy as np

from . import _ffi


class VariantType(enum.IntEnum):
    """variants.rs:10-33"""
    Snv = 0
    Insertion = 1
    Deletion = 2
    Indel = 3
    SvInsertion = 4
    SvDeletion = 5
    SvDuplication = 6
    SvInversion = 7
    SvBreakend = 8
    TandemRepeat = 9
    Unknown = 10


# read_parsing.rs:18-22 base qualities; global realignment doubles them (read_parsing.rs:815)
BASE_QUAL = {
    VariantType.Snv: 80,
    VariantType.Insertion: 10, VariantType.Deletion: 10, VariantType.Indel: 10,
    VariantType.SvInsertion: 20, VariantType.SvDeletion: 20,
    VariantType.TandemRepeat: 40,
}


@dataclass
class Variant:
    """What WFAGraph::from_reference_variants_with_hom reads from a `Variant` (truncated alleles, i.e.
    without the +-reference_buffer padding of variants.rs:497-539)."""
    variant_type: VariantType
    position: int
    ref_len: int
    allele0: bytes
    allele1: bytes
    index_allele0: int = 0
    index_allele1: int = 1
    is_ignored: bool = False
    vcf_index: int = 0
    prefix: bytes = b""     # reference bases added by add_reference_prefix (local re-alignment only)
    postfix: bytes = b""    # ... add_reference_postfix

    # +-reference_buffer padding (variants.rs:497-539; applied by phaser.rs:236-294), used by local re-alignment
    def add_reference_prefix(self, prefix):
        assert len(prefix) <= self.position - len(self.prefix)
        self.prefix = bytes(prefix) + self.prefix

    def add_reference_postfix(self, postfix):
        self.postfix = self.postfix + bytes(postfix)

    def truncate_reference_postfix(self, amount):
        assert amount <= len(self.postfix)
        self.postfix = self.postfix[:len(self.postfix) - amount]

    @property
    def prefix_len(self):
        return len(self.prefix)

    @property
    def postfix_len(self):
        return len(self.postfix)

    def get_allele0(self):
        return self.prefix + self.allele0 + self.postfix

    def get_allele1(self):
        return self.prefix + self.allele1 + self.postfix

    def match_allele(self, allele):
        """variants.rs:598-606"""
        allele = bytes(allele)
        return 0 if allele == self.get_allele0() else (1 if allele == self.get_allele1() else 2)

    # constructors with the reference's validation essentials (variants.rs:109-492)
    @staticmethod
    def new_snv(vcf_index, position, allele0, allele1, i0, i1):
        assert i0 < i1 and len(allele0) == 1 and len(allele1) == 1
        return Variant(VariantType.Snv, position, 1, bytes(allele0), bytes(allele1), i0, i1, False, vcf_index)

    @staticmethod
    def new_deletion(vcf_index, position, ref_len, allele0, allele1, i0, i1):
        assert i0 < i1 and ref_len > 1 and len(allele1) == 1
        assert len(allele0) == (ref_len if i0 == 0 else 1)
        return Variant(VariantType.Deletion, position, ref_len, bytes(allele0), bytes(allele1), i0, i1, False, vcf_index)

    @staticmethod
    def new_insertion(vcf_index, position, allele0, allele1, i0, i1):
        assert i0 < i1 and len(allele1) >= 1 and (len(allele0) == 1 if i0 == 0 else len(allele0) >= 1)
        return Variant(VariantType.Insertion, position, 1, bytes(allele0), bytes(allele1), i0, i1, False, vcf_index)

    @staticmethod
    def new_indel(vcf_index, position, ref_len, allele0, allele1, i0, i1):
        assert i0 < i1 and ref_len > 1 and len(allele1) >= 1
        assert len(allele0) == ref_len if i0 == 0 else len(allele0) >= 1
        return Variant(VariantType.Indel, position, ref_len, bytes(allele0), bytes(allele1), i0, i1, False, vcf_index)

    @staticmethod
    def new_sv_deletion(vcf_index, position, ref_len, allele0, allele1, i0=0, i1=1):
        assert (i0, i1) == (0, 1) and len(allele0) == ref_len and 1 <= len(allele1) <= len(allele0)
        return Variant(VariantType.SvDeletion, position, ref_len, bytes(allele0), bytes(allele1), i0, i1, False, vcf_index)

    @staticmethod
    def new_sv_insertion(vcf_index, position, ref_len, allele0, allele1, i0=0, i1=1):
        assert (i0, i1) == (0, 1) and len(allele0) == ref_len and len(allele1) >= len(allele0) >= 1
        return Variant(VariantType.SvInsertion, position, ref_len, bytes(allele0), bytes(allele1), i0, i1, False, vcf_index)

    @staticmethod
    def new_tandem_repeat(vcf_index, position, ref_len, allele0, allele1, i0, i1):
        assert i0 < i1 and len(allele0) >= 1 and len(allele1) >= 1 and (i0 != 0 or len(allele0) == ref_len)
        return Variant(VariantType.TandemRepeat, position, ref_len, bytes(allele0), bytes(allele1), i0, i1, False, vcf_index)


@dataclass
class WfaJobSpec:
    """One BAM record's global-realignment job (read_parsing.rs:738-780)."""
    reference: bytes          # chromosome (or slice) with reference[0] at coordinate ref_base
    ref_start: int
    ref_end: int
    hets: list
    homs: list
    read: bytes
    ref_base: int = 0
    _keep: list = field(default_factory=list, repr=False)


def _u8(buf):
    a = np.frombuffer(bytes(buf), dtype=np.uint8) if len(buf) else np.zeros(1, np.uint8)
    return a


def _pack_variants(variants, keep):
    n = len(variants)
    arr = (_ffi.WfaVariant * max(n, 1))()
    for i, v in enumerate(variants):
        a0, a1 = _u8(v.allele0), _u8(v.allele1)
        keep.extend([a0, a1])
        arr[i].position = v.position
        arr[i].ref_len = v.ref_len
        arr[i].flags = (1 if v.is_ignored else 0) | (2 if v.index_allele0 != 0 else 0)
        arr[i].allele0 = a0.ctypes.data_as(C.POINTER(C.c_uint8))
        arr[i].allele0_len = len(v.allele0)
        arr[i].allele1 = a1.ctypes.data_as(C.POINTER(C.c_uint8))
        arr[i].allele1_len = len(v.allele1)
    keep.append(arr)
    return arr


def make_jobs(specs):
    """list[WfaJobSpec] -> (ctypes array of hp_wfa_job, keepalive list)."""
    keep = []
    jobs = (_ffi.WfaJob * max(len(specs), 1))()
    for i, s in enumerate(specs):
        ref, read = _u8(s.reference), _u8(s.read)
        keep.extend([ref, read])
        jobs[i].reference = ref.ctypes.data_as(C.POINTER(C.c_uint8))
        jobs[i].ref_base = s.ref_base
        jobs[i].ref_start = s.ref_start
        jobs[i].ref_end = s.ref_end
        jobs[i].hets = _pack_variants(s.hets, keep)
        jobs[i].n_hets = len(s.hets)
        jobs[i].homs = _pack_variants(s.homs, keep)
        jobs[i].n_homs = len(s.homs)
        jobs[i].read = read.ctypes.data_as(C.POINTER(C.c_uint8))
        jobs[i].read_len = len(s.read)
    return jobs, keep


class PreparedWfaBatch:
    """ctypes job array built once (host marshalling is not part of the measured C call)."""

    def __init__(self, specs):
        self.specs = specs
        self.n = len(specs)
        self.jobs, self._keep = make_jobs(specs)
        self.out = (_ffi.WfaResult * max(self.n, 1))()
        self.alleles = [np.full(max(len(s.hets), 1), 3, np.uint8) for s in specs]
        self.ptrs = (C.c_void_p * max(self.n, 1))(*[a.ctypes.data for a in self.alleles])

    def run(self, prune_distance=500, max_edit_distance=500, device_id=0):
        dll = _ffi.lib()
        prune = (2 ** 64 - 1) if prune_distance in (0, None) else prune_distance
        import time
        t0 = time.perf_counter()
        rc = dll.hp_wfa_assign_batch(self.jobs, self.n, prune, max_edit_distance, self.out, self.ptrs, device_id)
        self.last_call_s = time.perf_counter() - t0   # the C call alone (the result list below is Python overhead)
        _ffi.check(rc)
        return [(self.out[i].status, self.out[i].score, self.out[i].n_nodes, self.alleles[i][:len(self.specs[i].hets)])
                for i in range(self.n)]


def wfa_assign_batch(specs, prune_distance=500, max_edit_distance=500, device_id=0):
    """hp_wfa_assign_batch. Returns a list of (status, score, n_nodes, alleles ndarray[n_hets]);
    status 1 == WFAGraphError::MaxEditDistance (caller falls back to local realignment,
    read_parsing.rs:564-575). prune_distance 0 means "no pruning" as in cli.rs:352-354."""
    dll = _ffi.lib()
    n = len(specs)
    jobs, keep = make_jobs(specs)
    out = (_ffi.WfaResult * max(n, 1))()
    alleles = [np.full(max(len(s.hets), 1), 3, np.uint8) for s in specs]
    ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in alleles])
    prune = (2 ** 64 - 1) if prune_distance in (0, None) else prune_distance
    _ffi.check(dll.hp_wfa_assign_batch(jobs, n, prune, max_edit_distance, out, ptrs, device_id))
    del keep
    return [(out[i].status, out[i].score, out[i].n_nodes, alleles[i][:len(specs[i].hets)]) for i in range(n)]


class WFAGraph:
    """WFAGraph::new / add_node / edit_distance_with_pruning (wfa_graph.rs:100-117, 298-331, 350-650) over
    hp_wfa_align_graphs: the layer under from_reference_variants_with_hom, for caller-built topologies."""

    def __init__(self):
        self.nodes = []   # (sequence bytes, sorted parents)

    def add_node(self, sequence, parents):
        """Returns the new node's index (creation order). The reference's asserts (the first node has no parents, every
        later one has some, all of them earlier nodes, :305-312) are checked by the library at alignment time."""
        self.nodes.append((np.ascontiguousarray(sequence, dtype=np.uint8), np.ascontiguousarray(sorted(parents), dtype=np.uint32)))
        return len(self.nodes) - 1

    def edit_distance_with_pruning(self, reads, prune_distance=None, max_edit_distance=500, device_id=0):
        """One call for a list of reads. Returns [(score, traversed node indices)] - WFAResult (:654-670); a read that
        exceeds max_edit_distance raises like the reference's WFAGraphError::MaxEditDistance would be matched: (None, [])."""
        dll = _ffi.lib()
        n = len(reads)
        gnodes = (_ffi.GraphNode * len(self.nodes))()
        for k, (seq, par) in enumerate(self.nodes):
            gnodes[k].seq = seq.ctypes.data_as(C.POINTER(C.c_uint8))
            gnodes[k].seq_len = len(seq)
            gnodes[k].n_parents = len(par)
            gnodes[k].parents = par.ctypes.data_as(C.POINTER(C.c_uint32))
        rd = [np.ascontiguousarray(r, dtype=np.uint8) for r in reads]
        jobs = (_ffi.GraphJob * max(n, 1))()
        for i, r in enumerate(rd):
            jobs[i].nodes = gnodes
            jobs[i].n_nodes = len(self.nodes)
            jobs[i].read = r.ctypes.data_as(C.POINTER(C.c_uint8))
            jobs[i].read_len = len(r)
        words = (len(self.nodes) + 31) // 32
        sets = [np.zeros(max(words, 1), np.uint32) for _ in range(n)]
        ptrs = (C.c_void_p * max(n, 1))(*[s.ctypes.data for s in sets])
        out = (_ffi.GraphResult * max(n, 1))()
        prune = (2 ** 64 - 1) if prune_distance in (0, None) else prune_distance
        _ffi.check(dll.hp_wfa_align_graphs(jobs, n, prune, max_edit_distance, out, ptrs, device_id))
        res = []
        for i in range(n):
            if out[i].status != 0:
                res.append((None, []))
                continue
            trav = [k for k in range(len(self.nodes)) if (int(sets[i][k >> 5]) >> (k & 31)) & 1]
            assert len(trav) == out[i].n_traversed
            res.append((int(out[i].score), trav))
        return res


def global_quals(alleles, variant_types):
    """read_parsing.rs:803-835: qual = 2 x base(type) for cells that ended 0/1, else 0."""
    q = np.zeros(len(alleles), np.uint8)
    for i, a in enumerate(alleles):
        if a < 2:
            q[i] = 2 * BASE_QUAL[VariantType(variant_types[i])]
    return q
