"""Synthetic records WITH CIGARs for the local re-alignment tests (reference src/read_parsing.rs:121-503): a
reference, every variant type the function handles (with the +-reference_buffer padding of phaser.rs:236-294), two
haplotypes, and reads whose CIGAR is the true edit script (variants + noise), plus soft/hard clips, zero qualities,
partially deleted SV deletions and reads that end inside a variant window."""
import ctypes as C

import numpy as np

from hiphase_amd import _ffi
from hiphase_amd.phaser import add_reference_buffer
from hiphase_amd.read_parsing import CIGAR_OPS, LocalRecord
from hiphase_amd.wfa_graph import Variant
from wfa_util import _Rng


def make_variants(r, ref, n_vars, buffer=15, min_gap=1):
    ref_len = len(ref)
    variants, pos = [], 40
    for _ in range(n_vars):
        pos += r.randint(min_gap, 120) if r.u01() < 0.8 else r.randint(min_gap, 12)   # some closer than the buffer
        if pos > ref_len - 700:
            break
        u = r.u01()
        b = ref[pos:pos + 1]
        if u < 0.55:
            v = Variant.new_snv(0, pos, b, bytes([b"ACGT"[(b"ACGT".index(b) + r.randint(1, 3)) % 4]]), 0, 1)
        elif u < 0.65:
            v = Variant.new_insertion(0, pos, b, b + r.dna(r.randint(1, 9)), 0, 1)
        elif u < 0.75:
            k = r.randint(1, 9)
            v = Variant.new_deletion(0, pos, k + 1, ref[pos:pos + k + 1], b, 0, 1)
        elif u < 0.80:
            k = r.randint(1, 6)
            if r.u01() < 0.5:
                v = Variant.new_indel(0, pos, k + 1, ref[pos:pos + k + 1], b + r.dna(r.randint(1, 6)), 0, 1)
            else:  # multi-allelic site: neither allele is the reference
                v = Variant.new_indel(0, pos, k + 1, b + r.dna(r.randint(1, 4)), b + r.dna(r.randint(5, 8)), 1, 2)
        elif u < 0.85:
            v = Variant.new_sv_insertion(0, pos, 1, b, b + r.dna(r.randint(50, 160)))
        elif u < 0.93:
            k = r.randint(50, 300)
            v = Variant.new_sv_deletion(0, pos, k, ref[pos:pos + k], b)
        else:
            k = r.randint(6, 30)
            v = Variant.new_tandem_repeat(0, pos, k, ref[pos:pos + k], ref[pos:pos + k] + r.dna(r.randint(2, 12)), 0, 1)
        if r.u01() < 0.04:
            v.is_ignored = True
        variants.append(v)
        pos += v.ref_len
    add_reference_buffer(variants, ref, buffer)
    return variants


def _rle(ops):
    out = []
    for o in ops:
        if out and out[-1][0] == o:
            out[-1][1] += 1
        else:
            out.append([o, 1])
    return [(o, n) for o, n in out]


def make_read(r, ref, variants, truth, a, b, noise, qname):
    """Haplotype `hap` over reference [a, b]; returns a LocalRecord whose CIGAR is the applied edit script."""
    hap = 0 if r.u01() < 0.5 else 1
    seq, ops, cur = bytearray(), [], a
    for i, v in enumerate(variants):
        if v.position < cur or v.position + v.ref_len > b + 1:
            continue
        seq += ref[cur:v.position]
        ops += ["M"] * (v.position - cur)
        al = truth[i] if hap == 0 else 1 - truth[i]
        ref_seg = ref[v.position:v.position + v.ref_len]
        allele = v.allele1 if al == 1 else v.allele0
        if int(v.variant_type) == 5 and al == 1 and r.u01() < 0.3:   # SV deletion only partly (or over-) deleted
            keep = r.randint(0, v.ref_len - 1)
            allele = ref_seg[:max(1, keep)]
        m = min(len(ref_seg), len(allele))
        seq += allele
        ops += ["M"] * m + ["I"] * (len(allele) - m) + ["D"] * (len(ref_seg) - m)
        cur = v.position + v.ref_len
    seq += ref[cur:b + 1]
    ops += ["M"] * (b + 1 - cur)
    # sequencing noise on top of the haplotype
    nseq, nops, k = bytearray(), [], 0
    for o in ops:
        if o == "D":
            nops.append("D")
            continue
        ch = seq[k]
        k += 1
        u = r.u01()
        if u < noise / 3:
            if o == "M":
                nops.append("D")
            continue
        if u < 2 * noise / 3:
            nseq.append(b"ACGT"[r.next() & 3]); nops.append(o)
            continue
        nseq.append(ch); nops.append(o)
        if u < noise:
            nseq.append(b"ACGT"[r.next() & 3]); nops.append("I")
    # a read must start and end on an aligned base for `pos` to be its first reference base
    while nops and nops[0] != "M":
        if nops[0] == "I":
            del nseq[0]
        else:
            a += 1
        del nops[0]
    while nops and nops[-1] != "M":
        if nops[-1] == "I":
            del nseq[-1]
        del nops[-1]
    cigar = _rle(nops)
    if r.u01() < 0.3:
        n = r.randint(1, 40)
        nseq = bytearray(r.dna(n)) + nseq
        cigar = [("S", n)] + cigar
    if r.u01() < 0.3:
        n = r.randint(1, 40)
        nseq = nseq + bytearray(r.dna(n))
        cigar = cigar + [("S", n)]
    if r.u01() < 0.15:
        cigar = [("H", r.randint(1, 500))] + cigar
    if r.u01() < 0.1:
        cigar = cigar + [("H", r.randint(1, 500))]
    qual = bytes((0 if r.u01() < 0.01 else r.randint(1, 60)) for _ in range(len(nseq)))
    return LocalRecord(qname, a, cigar, bytes(nseq), qual)


def make_local_block(seed, ref_len=8000, n_vars=60, n_reads=60, read_len=(400, 3000), noise=0.01, buffer=15):
    r = _Rng(seed)
    ref = r.dna(ref_len)
    variants = make_variants(r, ref, n_vars, buffer)
    truth = [1 if r.u01() < 0.5 else 0 for _ in variants]
    records = []
    for k in range(n_reads):
        ln = r.randint(*read_len)
        a = r.randint(0, ref_len - ln - 1)
        records.append(make_read(r, ref, variants, truth, a, a + ln - 1, noise, f"read{k}"))
    # the "weird CIGAR" of read_parsing.rs:370-372: soft clip, then a deletion, before the first aligned base
    v_sv = [v for v in variants if int(v.variant_type) == 5]
    if v_sv:
        v = v_sv[0]
        start = v.position + 3
        body = ref[start + 20:start + 20 + 400]
        records.append(LocalRecord("weird", start, [("S", 7), ("D", 20), ("M", len(body))], r.dna(7) + body,
                                   bytes([30]) * (7 + len(body))))
    return ref, variants, truth, records


# ---- oracle side (tests only) -------------------------------------------------------------------------------------
def pack_variants(variants, keep):
    vs = (_ffi.LocalVariant * max(len(variants), 1))()
    for i, v in enumerate(variants):
        a0 = np.frombuffer(v.get_allele0(), np.uint8)
        a1 = np.frombuffer(v.get_allele1(), np.uint8)
        keep += [a0, a1]
        vs[i].position, vs[i].ref_len, vs[i].variant_type = v.position, v.ref_len, int(v.variant_type)
        vs[i].prefix_len, vs[i].postfix_len = v.prefix_len, v.postfix_len
        vs[i].allele0 = a0.ctypes.data_as(C.POINTER(C.c_uint8))
        vs[i].allele1 = a1.ctypes.data_as(C.POINTER(C.c_uint8))
        vs[i].allele0_len, vs[i].allele1_len = a0.size, a1.size
        vs[i].flags = 1 if v.is_ignored else 0
    return vs


def pack_read(rec, keep):
    rd = _ffi.LocalRead()
    cg = np.array([(n << 4) | CIGAR_OPS.index(op) for op, n in rec.cigar] or [0], np.uint32)
    sq = np.frombuffer(rec.seq, np.uint8) if rec.seq else np.zeros(1, np.uint8)
    ql = np.frombuffer(rec.qual, np.uint8) if rec.qual else np.zeros(1, np.uint8)
    keep += [cg, sq, ql]
    rd.pos, rd.cigar, rd.n_cigar = rec.pos, cg.ctypes.data_as(C.POINTER(C.c_uint32)), len(rec.cigar)
    rd.seq_len = len(rec.seq)
    rd.seq, rd.qual = sq.ctypes.data_as(C.POINTER(C.c_uint8)), ql.ctypes.data_as(C.POINTER(C.c_uint8))
    return rd


def oracle_local(oracle_lib, records, variants):
    """hpo_local_realignment per record -> (alleles[R, N], quals[R, N], [stats tuples], [rc])."""
    keep = []
    vs = pack_variants(variants, keep)
    n = len(variants)
    al = np.zeros((len(records), max(n, 1)), np.uint8)
    ql = np.zeros((len(records), max(n, 1)), np.uint8)
    stats, rcs = [], []
    for i, rec in enumerate(records):
        rd = pack_read(rec, keep)
        st = _ffi.ReadStats()
        rcs.append(oracle_lib.hpo_local_realignment(C.byref(rd), vs, n, al[i].ctypes.data, ql[i].ctypes.data, C.byref(st)))
        stats.append(st.as_tuple())
    return al[:, :n], ql[:, :n], stats, rcs
