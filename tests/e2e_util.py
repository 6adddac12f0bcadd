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
