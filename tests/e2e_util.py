"""Synthetic read-bearing phase block (reference + het/hom variants + two haplotypes + noisy reads) for the
end-to-end tests: the closest thing to BASELINE.json configs[2] that can exist without BAM/VCF/htslib."""
from hiphase_amd.read_parsing import AlignedRecord
from hiphase_amd.wfa_graph import Variant
from wfa_util import _Rng


def make_block(seed, ref_len=40000, n_hets=50, n_homs=10, n_reads=120, read_len=(6000, 12000), noise=0.003):
    r = _Rng(seed)
    ref = bytearray(r.dna(ref_len))
    positions = sorted({r.randint(200, ref_len - 300) for _ in range(n_hets + n_homs)})
    variants, last_end = [], 0
    for pos in positions:
        if pos < last_end + 3:
            continue
        u = r.u01()
        if u < 0.85:
            alt = bytes([b"ACGT"[(b"ACGT".index(ref[pos]) + r.randint(1, 3)) % 4]])
            v = Variant.new_snv(0, pos, bytes(ref[pos:pos + 1]), alt, 0, 1)
        elif u < 0.93:
            ln = r.randint(1, 8)
            v = Variant.new_deletion(0, pos, ln + 1, bytes(ref[pos:pos + ln + 1]), bytes(ref[pos:pos + 1]), 0, 1)
        else:
            v = Variant.new_insertion(0, pos, bytes(ref[pos:pos + 1]), bytes(ref[pos:pos + 1]) + r.dna(r.randint(1, 8)), 0, 1)
        last_end = pos + v.ref_len
        variants.append(v)
    hom_idx = set()
    while len(hom_idx) < min(n_homs, len(variants) // 4):
        hom_idx.add(r.randint(0, len(variants) - 1))
    hets = [v for i, v in enumerate(variants) if i not in hom_idx]
    homs = [v for i, v in enumerate(variants) if i in hom_idx]
    truth = [1 if r.u01() < 0.5 else 0 for _ in hets]   # allele carried by haplotype 0
    het_of = {id(v): i for i, v in enumerate(hets)}

    def hap_window(hap, a, b):
        out, cur = bytearray(), a
        for v in variants:
            if v.position < cur or v.position + v.ref_len > b + 1:
                continue
            if id(v) in het_of:
                al = truth[het_of[id(v)]] if hap == 0 else 1 - truth[het_of[id(v)]]
            else:
                al = 1
            out += ref[cur:v.position]
            out += v.allele1 if al == 1 else ref[v.position:v.position + v.ref_len]
            cur = v.position + v.ref_len
        out += ref[cur:b + 1]
        return out

    records = []
    for k in range(n_reads):
        ln = r.randint(*read_len)
        a = r.randint(0, ref_len - ln - 1)
        b = a + ln - 1
        hap = 0 if r.u01() < 0.5 else 1
        seq = hap_window(hap, a, b)
        noisy = bytearray()
        for ch in seq:
            u = r.u01()
            if u < noise / 3:
                continue
            if u < 2 * noise / 3:
                noisy.append(b"ACGT"[r.next() & 3]); continue
            if u < noise:
                noisy.append(ch); noisy.append(b"ACGT"[r.next() & 3]); continue
            noisy.append(ch)
        # a few reads are split into two records with the same qname (supplementary alignments -> collapse)
        if k % 17 == 5 and ln > 4000:
            mid = a + ln // 2
            s1 = hap_window(hap, a, mid - 1)
            s2 = hap_window(hap, mid, b)
            records.append(AlignedRecord(f"read{k}", a, mid - 1, bytes(s1)))
            records.append(AlignedRecord(f"read{k}", mid, b, bytes(s2)))
        else:
            records.append(AlignedRecord(f"read{k}", a, b, bytes(noisy)))
    return bytes(ref), hets, homs, records, truth


# ---- a second, independent comparator of two hp_block_output structs (pure Python, field by field) ------------------------------
# Every whole-path parity claim of the suite and of bench.py goes through the product's hp_block_output_equal (compiled into the
# library under test). This one shares nothing with it: it names the FIRST field that differs, compares every scalar, and every
# array over the extent the outputs themselves declare. tests/test_comparator_cpu.py holds the two to each other, field class by
# field class.
def _arr(ptr, n):
    return [ptr[i] for i in range(n)]


def output_diff(inp, a, b):
    """-> None when every field hp_solve_blocks fills is identical in a and b (two hp_block_output of the same hp_block_input),
    else the name of the first differing field"""
    N = inp.n_hets
    scalars = ("status", "n_segments", "n_solver", "num_reads", "skipped_reads", "global_aligned", "local_aligned", "n_edit_distances", "num_alleles")
    for f in scalars:
        if getattr(a, f) != getattr(b, f):
            return f
    for f in ("exact_matches", "inexact_matches", "failed_matches", "allele0_matches", "allele1_matches"):
        if tuple(getattr(a, f)) != tuple(getattr(b, f)):
            return f
    ne = a.n_edit_distances
    if ne and _arr(a.edit_distances, ne) != _arr(b.edit_distances, ne):
        return "edit_distances"
    ns = a.n_segments
    for f in ("seg_qname", "seg_start", "seg_end", "seg_solver"):
        if ns and _arr(getattr(a, f), ns) != _arr(getattr(b, f), ns):
            return f
    if ns and _arr(a.seg_row_off, ns + 1) != _arr(b.seg_row_off, ns + 1):
        return "seg_row_off"
    cells = a.seg_row_off[ns] if ns else 0
    if cells:
        import ctypes as C
        for f in ("seg_alleles", "seg_quals"):
            if C.string_at(getattr(a, f), cells) != C.string_at(getattr(b, f), cells):
                return f
    if a.status != 0:
        return None   # (an unsupported block carries its segments only)
    for f in ("h1", "h2"):
        if _arr(getattr(a, f), N) != _arr(getattr(b, f), N):
            return f
    if a.stats.as_tuple() != b.stats.as_tuple():
        return "stats"
    if N > 1 and _arr(a.span_counts, N - 1) != _arr(b.span_counts, N - 1):
        return "span_counts"
    for f in ("seg_haplotag", "seg_first_het"):
        if ns and _arr(getattr(a, f), ns) != _arr(getattr(b, f), ns):
            return f
    return None


def outputs_diff(sset, got, exp):
    """[(block, first differing field)] over every block of a set"""
    return [(b, f) for b in range(sset.n) for f in [output_diff(sset.inputs[b], got.arr[b], exp.arr[b])] if f is not None]
