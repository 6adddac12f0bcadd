"""Shared helpers for the WFA tests: golden-variant construction and synthetic WFA jobs."""
import numpy as np

from hiphase_amd.wfa_graph import Variant, WfaJobSpec


def variant_from_golden(d):
    k = d["kind"]
    a0, a1 = d["allele0"].encode(), d["allele1"].encode()
    if k == "snv":
        return Variant.new_snv(d["vcf_index"], d["position"], a0, a1, d["index_allele0"], d["index_allele1"])
    if k == "deletion":
        return Variant.new_deletion(d["vcf_index"], d["position"], d["ref_len"], a0, a1, d["index_allele0"], d["index_allele1"])
    if k == "insertion":
        return Variant.new_insertion(d["vcf_index"], d["position"], a0, a1, d["index_allele0"], d["index_allele1"])
    if k == "indel":
        return Variant.new_indel(d["vcf_index"], d["position"], d["ref_len"], a0, a1, d["index_allele0"], d["index_allele1"])
    raise ValueError(k)


def spec_from_golden(g, read=b""):
    return WfaJobSpec(reference=g["reference"].encode(), ref_start=g["ref_start"], ref_end=g["ref_end"],
                      hets=[variant_from_golden(v) for v in g["variants"]],
                      homs=[variant_from_golden(v) for v in g["homs"]], read=bytes(read))


class _Rng:
    def __init__(self, seed):
        self.x = seed & (2 ** 64 - 1)

    def next(self):
        self.x = (self.x + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z = self.x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        return z ^ (z >> 31)

    def u01(self):
        return (self.next() >> 11) / 9007199254740992.0

    def randint(self, lo, hi):  # inclusive
        return lo + int(self.u01() * (hi - lo + 1))

    def dna(self, n):
        return bytes(b"ACGT"[self.next() & 3] for _ in range(n))


def synth_wfa_job(seed, ref_len=2000, n_vars=8, noise=0.005, n_homs=2, margin=50, multiallelic=0.1):
    """SURVEY.md §8(d) 'WFA synthetic': random reference, variant mix (SNV .85 / indel .12 / SV .01 / TR .02),
    read = one haplotype path through the window + uniform edit noise. Returns (WfaJobSpec, chosen alleles)."""
    r = _Rng(seed)
    ref = bytearray(r.dna(ref_len))
    # non-overlapping variant sites, sorted
    positions = sorted({r.randint(margin, ref_len - margin - 600) for _ in range(n_vars + n_homs)})
    hets, homs, chosen = [], [], []
    last_end = 0
    sites = []
    for pos in positions:
        if pos < last_end + 2:
            continue
        u = r.u01()
        if u < 0.85:
            alt = bytes([b"ACGT"[(b"ACGT".index(ref[pos]) + r.randint(1, 3)) % 4]])
            v = Variant.new_snv(0, pos, bytes(ref[pos:pos + 1]), alt, 0, 1)
        elif u < 0.97:
            ln = r.randint(1, 10)
            if r.u01() < 0.5:
                v = Variant.new_deletion(0, pos, ln + 1, bytes(ref[pos:pos + ln + 1]), bytes(ref[pos:pos + 1]), 0, 1)
            else:
                v = Variant.new_insertion(0, pos, bytes(ref[pos:pos + 1]), bytes(ref[pos:pos + 1]) + r.dna(ln), 0, 1)
        elif u < 0.98:
            ln = r.randint(50, 500)
            if r.u01() < 0.5:
                v = Variant.new_sv_deletion(0, pos, ln + 1, bytes(ref[pos:pos + ln + 1]), bytes(ref[pos:pos + 1]))
            else:
                v = Variant.new_sv_insertion(0, pos, 1, bytes(ref[pos:pos + 1]), bytes(ref[pos:pos + 1]) + r.dna(ln))
        else:
            unit = r.dna(r.randint(2, 6))
            copies = r.randint(5, 30)
            tr = unit * copies
            ref[pos + 1:pos + 1 + len(tr)] = tr  # plant the repeat in the reference
            delta = r.randint(1, 4)
            a0 = bytes(ref[pos:pos + 1 + len(tr)])
            a1 = a0 + unit * delta if r.u01() < 0.5 else a0[:len(a0) - len(unit) * min(delta, copies - 1)]
            v = Variant.new_tandem_repeat(0, pos, len(a0), a0, a1, 0, 1)
        if r.u01() < multiallelic and v.variant_type.name in ("Snv",):
            # multi-allelic SNV: allele0 is itself an ALT (index_allele0 != 0)
            others = [c for c in b"ACGT" if c not in (ref[pos], v.allele1[0])]
            v = Variant.new_snv(0, pos, bytes([others[0]]), v.allele1, 1, 2)
        last_end = pos + v.ref_len
        sites.append(v)
    # split into hets / homs
    for v in sites:
        if len(homs) < n_homs and r.u01() < n_homs / max(1, len(sites)):
            homs.append(v)
        else:
            hets.append(v)
    # read window and haplotype choice
    win_start = r.randint(0, margin - 1)
    win_end = ref_len - r.randint(0, margin - 1)
    seq = bytearray()
    cur = win_start
    for v in sorted(hets + homs, key=lambda x: x.position):
        is_hom = any(v is h for h in homs)
        pick = 1 if is_hom else (1 if r.u01() < 0.5 else 0)
        if not is_hom:
            chosen.append((v, pick))
        if v.position < cur:
            continue
        seq += ref[cur:v.position]
        if pick == 1:
            seq += v.allele1
        elif v.index_allele0 != 0:
            seq += v.allele0
        else:
            seq += ref[v.position:v.position + v.ref_len]
        cur = v.position + v.ref_len
    seq += ref[cur:win_end]
    # uniform edit noise
    out = bytearray()
    for b in seq:
        u = r.u01()
        if u < noise / 3:
            continue
        if u < 2 * noise / 3:
            out.append(b"ACGT"[r.next() & 3])
            continue
        if u < noise:
            out.append(b)
            out.append(b"ACGT"[r.next() & 3])
            continue
        out.append(b)
    spec = WfaJobSpec(reference=bytes(ref), ref_start=win_start, ref_end=win_end, hets=hets, homs=homs, read=bytes(out))
    return spec, chosen
