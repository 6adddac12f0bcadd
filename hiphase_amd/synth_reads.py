"""Deterministic synthetic READ-BEARING phase blocks (numpy; used by bench.py's whole-path workload and by tests): the
closest thing to BASELINE.json configs[2-3] that can exist without BAM/VCF/htslib - a random reference, het and hom
small variants at human-like density, two haplotypes, HiFi-like reads (length ~ N(15 kb, 3 kb), substitution noise)
at a given coverage, handed over exactly as `solve_block` sees them once variants and records are decoded
(hiphase_amd.blocks.BlockSpec: position-sorted Variant lists + AlignedRecord per read)."""
import numpy as np

from .blocks import BlockSpec
from .read_parsing import AlignedRecord
from .wfa_graph import Variant

_ACGT = np.frombuffer(b"ACGT", np.uint8)


def synth_read_block(seed, n_hets, block_index=0, het_spacing=1000.0, hom_ratio=0.6, coverage=30.0, read_mean=15000.0,
                     read_sd=3000.0, noise=0.003, indel_frac=0.10):
    """-> (BlockSpec, truth[n_hets]) ; truth[i] = allele haplotype 0 carries at het i."""
    rng = np.random.default_rng(seed)
    n_homs = int(round(n_hets * hom_ratio))
    n_var = n_hets + n_homs
    gaps = rng.exponential(het_spacing / (1.0 + hom_ratio), n_var).astype(np.int64) + 12
    pos = 3000 + np.cumsum(gaps)
    region_len = int(pos[-1]) + 3000
    code = rng.integers(0, 4, region_len, dtype=np.uint8)
    ref = _ACGT[code]
    u = rng.random(n_var)
    is_del = (u >= 1.0 - indel_frac) & (u < 1.0 - indel_frac / 2)
    is_ins = u >= 1.0 - indel_frac / 2
    klen = rng.integers(1, 9, n_var)
    ref_len = np.where(is_del, klen + 1, 1).astype(np.int64)
    het_idx = np.sort(rng.choice(n_var, n_hets, replace=False))
    is_het = np.zeros(n_var, bool)
    is_het[het_idx] = True
    snv_shift = rng.integers(1, 4, n_var)
    a0, a1 = [], []
    for i in range(n_var):
        p = int(pos[i])
        if is_del[i]:
            a0.append(ref[p:p + int(ref_len[i])].tobytes()); a1.append(ref[p:p + 1].tobytes())
        elif is_ins[i]:
            ins = _ACGT[rng.integers(0, 4, int(klen[i]))]
            a0.append(ref[p:p + 1].tobytes()); a1.append(ref[p:p + 1].tobytes() + ins.tobytes())
        else:
            a0.append(ref[p:p + 1].tobytes()); a1.append(_ACGT[(code[p] + snv_shift[i]) % 4:(code[p] + snv_shift[i]) % 4 + 1].tobytes())
    truth_all = rng.integers(0, 2, n_var)
    variants = []
    for i in range(n_var):
        p = int(pos[i])
        if is_del[i]:
            v = Variant.new_deletion(0, p, int(ref_len[i]), a0[i], a1[i], 0, 1)
        elif is_ins[i]:
            v = Variant.new_insertion(0, p, a0[i], a1[i], 0, 1)
        else:
            v = Variant.new_snv(0, p, a0[i], a1[i], 0, 1)
        variants.append(v)
    hets = [variants[i] for i in range(n_var) if is_het[i]]
    homs = [variants[i] for i in range(n_var) if not is_het[i]]
    # the two haplotypes and, per haplotype, the coordinate shift after each variant
    haps, shifts = [], []
    for h in (0, 1):
        pieces, cur, shift = [], 0, np.zeros(n_var + 1, np.int64)
        for i in range(n_var):
            p = int(pos[i])
            carries_alt = (not is_het[i]) or ((truth_all[i] ^ h) == 1)
            al = a1[i] if carries_alt else a0[i]
            pieces.append(ref[cur:p])
            pieces.append(np.frombuffer(al, np.uint8))
            cur = p + int(ref_len[i])
            shift[i + 1] = shift[i] + len(al) - int(ref_len[i])
        pieces.append(ref[cur:])
        haps.append(np.concatenate(pieces))
        shifts.append(shift)
    var_end = pos + ref_len   # first plain base after each variant

    def plain(x):   # move a coordinate off a variant's reference span (to the base before it)
        i = np.searchsorted(pos, x, "right") - 1
        inside = (i >= 0) & (x < var_end[np.maximum(i, 0)])
        return np.where(inside, pos[np.maximum(i, 0)] - 1, x)

    n_reads = max(2, int(np.ceil(coverage * region_len / read_mean)))
    lens = np.clip(rng.normal(read_mean, read_sd, n_reads), 3000, 30000).astype(np.int64)
    lens = np.minimum(lens, region_len - 2)
    starts = (rng.random(n_reads) * (region_len - lens)).astype(np.int64)
    a = plain(starts)
    b = plain(np.minimum(starts + lens - 1, region_len - 1))
    hap_of = rng.integers(0, 2, n_reads)
    records = []
    for k in range(n_reads):
        h = int(hap_of[k])
        ia = np.searchsorted(var_end, a[k], "right")     # variants wholly before a
        ib = np.searchsorted(var_end, b[k], "right")
        s = haps[h][int(a[k] + shifts[h][ia]):int(b[k] + shifts[h][ib]) + 1].copy()
        m = np.flatnonzero(rng.random(len(s)) < noise)
        if len(m):   # substitution noise: another base
            lut = np.zeros(256, np.uint8)
            lut[_ACGT] = np.arange(4, dtype=np.uint8)
            s[m] = _ACGT[(lut[s[m]] + rng.integers(1, 4, len(m))) % 4]
        records.append(AlignedRecord(f"b{block_index}r{k}", int(a[k]), int(b[k]), s.tobytes()))
    truth = truth_all[is_het]
    return BlockSpec(block_index, ref.tobytes(), hets, homs, records), truth


def synth_wgs_like_mix(seed, total_hets, max_hets=2000, **kw):
    """Blocks with the heavy-tailed size distribution of a WGS run (docs/user_guide.md:257: median 15 hets per block,
    mean ~220, max ~4000), until `total_hets` hets are reached. -> [BlockSpec]"""
    rng = np.random.default_rng(seed)
    blocks, hets = [], 0
    while hets < total_hets:
        n = int(np.clip(np.exp(rng.normal(np.log(15.0), 2.2)), 2, max_hets))
        n = min(n, max(2, total_hets - hets))
        blocks.append(synth_read_block(seed * 1000003 + len(blocks), n, block_index=len(blocks), **kw)[0])
        hets += n
    return blocks
