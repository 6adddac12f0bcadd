// hp_synth_reads.cpp — deterministic synthetic READ-BEARING phase blocks (SURVEY.md §8(d) "WFA synthetic", widened to block sets):
// the workload of bench.py's whole-path legs and of the block-level tests, produced straight in the C layout that
// hp_solve_blocks / hp_blockstream_submit / the oracle's hpo_solve_block take (no marshalling in between).
//
// Host-only, no device code; compiled into libhiphase_gpu.so and into the test oracle alike (like hp_synth.cpp). It stands in
// for what BASELINE.json configs[2-4] would read from a BAM + VCF + FASTA (none of which exist in this image): per block a random
// reference, het + hom calls at human-like density in the type mix of SURVEY.md §8(d) (SNV .85, indel .12, SV .01, tandem repeat
// .02; the base qualities of reference src/read_parsing.rs:18-22 hang on the type), two haplotypes, and HiFi-like reads
// (length ~ N(15 kb, 3 kb), 30x) carrying one haplotype with uniform EDIT noise (substitutions, insertions, deletions), a noisy
// tail that runs into max_edit_distance (-> Err(MaxEditDistance) -> local re-alignment, read_parsing.rs:564-575), and
// supplementary records (two records of one read name -> ReadSegment::collapse, read_segments.rs:71-121). Every record comes
// with the CIGAR view local re-alignment needs, and with its bases either as ASCII or in the BAM's own 4-bit encoding.
//
// One splitmix64 stream per block, seeded by (seed, block index): blocks are generated on host threads, the set is a pure
// function of the spec. Block sizes: lognormal, median 15 hets, capped (docs/user_guide.md:257-260: median 15, max 4 165).
#include "../../include/hiphase_gpu.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t x;
    explicit Rng(uint64_t seed) : x(seed) {}
    uint64_t next() {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double u01() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(u01() * n); }                      // [0, n)
    uint32_t range(uint32_t lo, uint32_t hi) { return lo + below(hi - lo + 1); }    // [lo, hi]
    double normal() { const double u = std::max(u01(), 1e-300), v = u01(); return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v); }
    double expo(double mean) { return -mean * std::log(std::max(1.0 - u01(), 1e-300)); }
};

const char ACGT[5] = "ACGT";
inline uint8_t other_base(uint8_t b, Rng& r) {
    const uint32_t k = b == 'A' ? 0 : b == 'C' ? 1 : b == 'G' ? 2 : 3;
    return (uint8_t)ACGT[(k + 1 + r.below(3)) & 3];
}
inline uint8_t bam4_code(uint8_t c) {   // htslib's seq_nt16_table restricted to what the generator emits
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
}

enum { T_SNV = 0, T_INS = 1, T_DEL = 2, T_SVINS = 4, T_SVDEL = 5, T_TR = 9 };   // VariantType repr (variants.rs:10-33)
const uint32_t PAD = 15;   // reference bases on either side of a het's alleles for local re-alignment (phaser.rs:236-293)

struct Var {
    int64_t pos = 0;
    uint32_t ref_len = 1, type = T_SNV;
    bool het = true, multi = false;      // multi: 1|2 genotype, allele0 is itself an ALT (index_allele0 != 0)
    std::string a0, a1;                  // truncated alleles; a0 == reference span unless multi
};

struct BlockData {
    std::vector<uint8_t> reference;
    std::vector<Var> vars;               // position order
    // flat stores the C structs point into (never resized after finalize())
    std::vector<uint8_t> bytes;          // allele strings (truncated + padded)
    std::vector<uint8_t> reads;          // read bases (ASCII, or 4 bits per base)
    std::vector<uint32_t> cigars;
    std::vector<hp_wfa_variant> hets, homs;
    std::vector<hp_local_variant> local_hets;
    std::vector<uint8_t> het_types, truth;
    std::vector<hp_block_record> records;
    std::vector<hp_local_read> locals;
    uint32_t n_qnames = 0;
    uint64_t read_bases = 0, seg_cell_cap = 0;
};

struct RecTmp {          // a record before the stores are laid out
    int64_t a, b;        // first / last reference base (inclusive)
    uint32_t qname;
    std::vector<uint8_t> seq;
    std::vector<uint32_t> cigar;
};

void cigar_push(std::vector<uint32_t>& c, uint32_t op, uint32_t n) {
    if (!n) return;
    if (!c.empty() && (c.back() & 15u) == op) c.back() += n << 4;
    else c.push_back((n << 4) | op);
}

// one record: the bases of haplotype `hap` over reference [a, b] with edit noise, and its CIGAR against the reference
void make_record(const BlockData& B, const std::vector<uint8_t>& carries_alt /* per var, this haplotype */, int64_t a, int64_t b, double noise, double hp_share, Rng& r, RecTmp& out) {
    out.a = a; out.b = b;
    out.seq.clear(); out.cigar.clear();
    enum { M = 0, I = 1, D = 2 };
    const uint8_t* ref = B.reference.data();
    bool first = true;
    // HiFi-shaped errors (hp_share > 0): that share of the errors sits on bases that CONTINUE a homopolymer run (a quarter of a random
    // template's bases) as an insertion / deletion of the run's base; the other bases take the rest, uniform sub / ins / del
    const double f_run = 0.25, p_run = hp_share > 0 ? noise * hp_share / f_run : noise, p_other = hp_share > 0 ? noise * (1.0 - hp_share) / (1.0 - f_run) : noise;
    uint8_t prev_base = 0;
    // emits one template base: kind M (consumes a reference base) or I (read only); `last` = the record's final base
    auto emit = [&](uint8_t base, int kind, bool last) {
        const bool clean = first || last;   // a record begins and ends on an aligned base (min / max position, read_parsing.rs:672-685)
        first = false;
        const bool in_run = hp_share > 0 && base == prev_base;
        prev_base = base;
        if (in_run) {
            if (!clean && r.u01() < p_run) {
                if (r.below(2) == 0) { out.seq.push_back(base); cigar_push(out.cigar, kind == M ? M : I, 1); out.seq.push_back(base); cigar_push(out.cigar, I, 1); }   // the run one longer
                else { if (kind == M) cigar_push(out.cigar, D, 1); }                                                                                                    // ... or one shorter
                return;
            }
            out.seq.push_back(base);
            cigar_push(out.cigar, kind == M ? M : I, 1);
            return;
        }
        if (!clean && r.u01() < p_other) {
            const uint32_t k = r.below(3);
            if (k == 0) { out.seq.push_back(other_base(base, r)); cigar_push(out.cigar, kind == M ? M : I, 1); }     // substitution
            else if (k == 1) { out.seq.push_back(base); cigar_push(out.cigar, kind == M ? M : I, 1);                      // insertion after it
                               out.seq.push_back((uint8_t)ACGT[r.below(4)]); cigar_push(out.cigar, I, 1); }
            else { if (kind == M) cigar_push(out.cigar, D, 1); }                                                         // deletion
            return;
        }
        out.seq.push_back(base);
        cigar_push(out.cigar, kind == M ? M : I, 1);
    };
    int64_t cur = a;
    // variants wholly inside [a, b] (a and b never fall inside a variant's reference span)
    size_t vi = (size_t)(std::lower_bound(B.vars.begin(), B.vars.end(), a, [](const Var& v, int64_t p) { return v.pos < p; }) - B.vars.begin());
    for (; vi < B.vars.size() && B.vars[vi].pos + (int64_t)B.vars[vi].ref_len - 1 <= b; ++vi) {
        const Var& v = B.vars[vi];
        for (; cur < v.pos; ++cur) emit(ref[cur], M, false);
        const std::string& al = carries_alt[vi] ? v.a1 : v.a0;
        const uint32_t common = (uint32_t)std::min<size_t>(al.size(), v.ref_len);
        for (uint32_t k = 0; k < common; ++k) emit((uint8_t)al[k], M, false);
        for (size_t k = common; k < al.size(); ++k) emit((uint8_t)al[k], I, false);
        if (v.ref_len > common) cigar_push(out.cigar, D, v.ref_len - common);
        cur = v.pos + v.ref_len;
    }
    for (; cur <= b; ++cur) emit(ref[cur], M, cur == b);
}

void build_block(const hp_synth_reads_spec& S, uint64_t block_index, uint32_t n_hets, BlockData& B) {
    Rng r(S.seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull * (block_index + 1));
    const uint32_t n_homs = (uint32_t)std::llround(n_hets * S.hom_ratio);
    const uint32_t n_var = n_hets + n_homs;
    // ---- variants: which are het, type, lengths, positions ----
    B.vars.assign(n_var, Var{});
    {
        std::vector<uint32_t> idx(n_var);
        for (uint32_t i = 0; i < n_var; ++i) idx[i] = i;
        for (uint32_t i = 0; i < n_var; ++i) { const uint32_t j = i + r.below(n_var - i); std::swap(idx[i], idx[j]); }
        for (uint32_t i = 0; i < n_var; ++i) B.vars[i].het = false;
        for (uint32_t i = 0; i < n_hets; ++i) B.vars[idx[i]].het = true;
    }
    struct Plan { uint32_t unit = 0, ref_copies = 0, alt_copies = 0, alt2_copies = 0, k = 0; };
    std::vector<Plan> plan(n_var);
    const double gap_mean = S.het_spacing / (1.0 + S.hom_ratio);
    int64_t pos = 3000;
    for (uint32_t i = 0; i < n_var; ++i) {
        Var& v = B.vars[i];
        const double u = r.u01();
        Plan& p = plan[i];
        if (u < S.frac_snv) v.type = T_SNV;
        else if (u < S.frac_snv + S.frac_indel) { v.type = r.u01() < 0.5 ? T_INS : T_DEL; p.k = r.range(1, 10); }
        else if (u < S.frac_snv + S.frac_indel + S.frac_sv) { v.type = r.u01() < 0.5 ? T_SVINS : T_SVDEL; p.k = r.range(50, 500); }
        else {
            v.type = T_TR;
            p.unit = r.range(2, 6); p.ref_copies = r.range(5, 30);
            do { p.alt_copies = r.range(std::max(1u, p.ref_copies - 4), p.ref_copies + 4); } while (p.alt_copies == p.ref_copies);
            if (v.het && r.u01() < S.frac_multiallelic) {
                v.multi = true;
                do { p.alt2_copies = r.range(std::max(1u, p.ref_copies - 4), p.ref_copies + 4); } while (p.alt2_copies == p.ref_copies || p.alt2_copies == p.alt_copies);
            }
        }
        v.ref_len = v.type == T_DEL || v.type == T_SVDEL ? p.k + 1 : v.type == T_TR ? p.unit * p.ref_copies : 1;
        pos += 2 * PAD + 5 + (int64_t)r.expo(gap_mean);
        v.pos = pos;
        pos += v.ref_len;
    }
    const int64_t region_len = pos + 3000;
    B.reference.resize((size_t)region_len);
    for (auto& c : B.reference) c = (uint8_t)ACGT[r.below(4)];
    for (uint32_t i = 0; i < n_var; ++i) {
        Var& v = B.vars[i];
        const Plan& p = plan[i];
        uint8_t* ref = B.reference.data() + v.pos;
        auto rnd = [&](uint32_t n) { std::string s(n, 'A'); for (auto& c : s) c = ACGT[r.below(4)]; return s; };
        if (v.type == T_TR) {
            const std::string unit = rnd(p.unit);
            for (uint32_t k = 0; k < v.ref_len; ++k) ref[k] = (uint8_t)unit[k % p.unit];
            auto tract = [&](uint32_t copies) { std::string s; for (uint32_t c = 0; c < copies; ++c) s += unit; return s; };
            v.a0 = v.multi ? tract(p.alt2_copies) : tract(p.ref_copies);
            v.a1 = tract(p.alt_copies);
        } else if (v.type == T_SNV) {
            v.a0.assign(1, (char)ref[0]); v.a1.assign(1, (char)other_base(ref[0], r));
        } else if (v.type == T_INS || v.type == T_SVINS) {
            v.a0.assign(1, (char)ref[0]); v.a1 = v.a0 + rnd(p.k);
        } else {
            v.a0.assign(reinterpret_cast<const char*>(ref), v.ref_len); v.a1.assign(1, (char)ref[0]);
        }
    }
    // ---- haplotypes: which allele each carries ----
    std::vector<uint8_t> carries[2];
    carries[0].resize(n_var); carries[1].resize(n_var);
    for (uint32_t i = 0; i < n_var; ++i) {
        const uint32_t t = r.below(2);
        carries[0][i] = B.vars[i].het ? (uint8_t)t : 1;
        carries[1][i] = B.vars[i].het ? (uint8_t)(t ^ 1u) : 1;
        if (B.vars[i].het) B.truth.push_back((uint8_t)t);
    }
    // ---- reads ----
    auto plain = [&](int64_t x) {   // move a coordinate off a variant's reference span (to the base before it)
        auto it = std::upper_bound(B.vars.begin(), B.vars.end(), x, [](int64_t p, const Var& v) { return p < v.pos; });
        if (it == B.vars.begin()) return x;
        --it;
        return (x < it->pos + (int64_t)it->ref_len) ? it->pos - 1 : x;
    };
    const uint32_t n_reads = std::max<uint32_t>(2, (uint32_t)std::ceil(S.coverage * (double)region_len / S.read_mean));
    std::vector<RecTmp> recs;
    recs.reserve(n_reads + n_reads / 16 + 2);
    uint32_t qn = 0;
    for (uint32_t k = 0; k < n_reads; ++k) {
        int64_t len = (int64_t)std::llround(S.read_mean + S.read_sd * r.normal());
        len = std::min<int64_t>(std::max<int64_t>(len, 3000), 30000);
        len = std::min<int64_t>(len, region_len - 2);
        const int64_t start = (int64_t)(r.u01() * (double)(region_len - len));
        const int64_t a = plain(start), b = std::max(plain(std::min(start + len - 1, region_len - 1)), a);
        const uint32_t hap = r.below(2);
        double noise = r.u01() < S.noisy_fraction ? S.noisy_noise : S.edit_noise;
        if (S.hifi_sigma > 0 && noise == S.edit_noise)   // per-read rate: lognormal around the median (one normal draw per read)
            noise = std::min(0.04, std::max(S.edit_noise / 20.0, S.edit_noise * std::exp(S.hifi_sigma * r.normal())));
        // wrong-haplotype cells (allele_switch > 0; no draw otherwise: older sets stay byte-for-byte): the hets this read spans, flipped one by one
        std::vector<uint8_t> switched;
        if (S.allele_switch > 0) {
            switched = carries[hap];
            size_t vi = (size_t)(std::lower_bound(B.vars.begin(), B.vars.end(), a, [](const Var& v, int64_t p) { return v.pos < p; }) - B.vars.begin());
            for (; vi < B.vars.size() && B.vars[vi].pos <= b; ++vi) if (B.vars[vi].het && r.u01() < S.allele_switch) switched[vi] ^= 1u;
        }
        const std::vector<uint8_t>& carried = S.allele_switch > 0 ? switched : carries[hap];
        const bool split = r.u01() < S.supplementary_fraction && b - a > 4000;
        if (split) {
            const int64_t mid = plain((a + b) / 2);
            if (mid > a + 10 && mid < b - 10) {
                recs.emplace_back(); make_record(B, carried, a, mid, noise, S.homopolymer_share, r, recs.back()); recs.back().qname = qn;
                // (the second record starts on the next plain base: mid + 1 may sit on a variant's first base)
                int64_t a2 = mid + 1;
                while (plain(a2) != a2 || [&] { auto it = std::lower_bound(B.vars.begin(), B.vars.end(), a2, [](const Var& v, int64_t p) { return v.pos < p; }); return it != B.vars.end() && it->pos == a2; }()) ++a2;
                if (a2 < b) { recs.emplace_back(); make_record(B, carried, a2, b, noise, S.homopolymer_share, r, recs.back()); recs.back().qname = qn; }
                ++qn;
                continue;
            }
        }
        recs.emplace_back();
        make_record(B, carried, a, b, noise, S.homopolymer_share, r, recs.back());
        recs.back().qname = qn++;
    }
    std::stable_sort(recs.begin(), recs.end(), [](const RecTmp& x, const RecTmp& y) { return x.a < y.a; });   // BAM order
    // read names are numbered in first-seen order
    {
        std::vector<uint32_t> remap(qn, UINT32_MAX);
        uint32_t next = 0;
        for (auto& rc : recs) { if (remap[rc.qname] == UINT32_MAX) remap[rc.qname] = next++; rc.qname = remap[rc.qname]; }
        B.n_qnames = next;
    }
    // ---- lay the stores out, then the C structs over them ----
    const bool bam4 = S.seq_format == HP_SEQ_BAM4;
    size_t nbytes = 0, ncig = 0, nread = 0;
    for (const Var& v : B.vars) nbytes += v.a0.size() + v.a1.size() + (v.het ? v.a0.size() + v.a1.size() + 4 * PAD : 0);
    for (const RecTmp& rc : recs) { ncig += rc.cigar.size(); nread += (bam4 ? (rc.seq.size() + 1) / 2 : rc.seq.size()) + 16; B.read_bases += rc.seq.size(); }
    B.bytes.resize(nbytes + 16); B.cigars.resize(ncig + 1); B.reads.resize(nread + 64);
    size_t ob = 0;
    auto put = [&](const uint8_t* p, size_t n) { uint8_t* d = B.bytes.data() + ob; if (n) std::memcpy(d, p, n); ob += n; return d; };
    B.hets.reserve(n_hets); B.homs.reserve(n_homs); B.local_hets.reserve(n_hets); B.het_types.reserve(n_hets);
    for (const Var& v : B.vars) {
        hp_wfa_variant w{};
        w.position = v.pos; w.ref_len = v.ref_len; w.flags = v.multi ? 2u : 0u;
        w.allele0 = put(reinterpret_cast<const uint8_t*>(v.a0.data()), v.a0.size()); w.allele0_len = (uint32_t)v.a0.size();
        w.allele1 = put(reinterpret_cast<const uint8_t*>(v.a1.data()), v.a1.size()); w.allele1_len = (uint32_t)v.a1.size();
        if (!v.het) { B.homs.push_back(w); continue; }
        B.hets.push_back(w);
        B.het_types.push_back((uint8_t)v.type);
        hp_local_variant lv{};
        lv.position = v.pos; lv.ref_len = v.ref_len; lv.variant_type = v.type; lv.prefix_len = PAD; lv.postfix_len = PAD;
        const uint8_t* pre = B.reference.data() + v.pos - PAD;
        const uint8_t* post = B.reference.data() + v.pos + v.ref_len;
        lv.allele0 = B.bytes.data() + ob; put(pre, PAD); put(reinterpret_cast<const uint8_t*>(v.a0.data()), v.a0.size()); put(post, PAD);
        lv.allele0_len = (uint32_t)(v.a0.size() + 2 * PAD);
        lv.allele1 = B.bytes.data() + ob; put(pre, PAD); put(reinterpret_cast<const uint8_t*>(v.a1.data()), v.a1.size()); put(post, PAD);
        lv.allele1_len = (uint32_t)(v.a1.size() + 2 * PAD);
        B.local_hets.push_back(lv);
    }
    B.records.resize(recs.size()); B.locals.resize(recs.size());
    size_t orr = 0, oc = 0;
    // per read name: the hull of its records' het ranges bounds the cells of its collapsed segment
    std::vector<uint32_t> qlo(B.n_qnames, UINT32_MAX), qhi(B.n_qnames, 0);
    for (size_t i = 0; i < recs.size(); ++i) {
        const RecTmp& rc = recs[i];
        uint8_t* dst = B.reads.data() + orr;
        if (bam4) {
            for (size_t k = 0; k + 1 < rc.seq.size(); k += 2) dst[k >> 1] = (uint8_t)((bam4_code(rc.seq[k]) << 4) | bam4_code(rc.seq[k + 1]));
            if (rc.seq.size() & 1) dst[rc.seq.size() >> 1] = (uint8_t)(bam4_code(rc.seq.back()) << 4);
            orr += (rc.seq.size() + 1) / 2;
        } else { std::memcpy(dst, rc.seq.data(), rc.seq.size()); orr += rc.seq.size(); }
        orr = (orr + 15) & ~(size_t)15;
        std::memcpy(B.cigars.data() + oc, rc.cigar.data(), rc.cigar.size() * 4);
        hp_local_read& lr = B.locals[i];
        lr = hp_local_read{};
        lr.pos = rc.a; lr.cigar = B.cigars.data() + oc; lr.n_cigar = (uint32_t)rc.cigar.size();
        lr.seq_len = (uint32_t)rc.seq.size(); lr.seq = dst; lr.qual = nullptr;   // (qual: the set's shared pattern, set by the caller)
        lr.seq_format = S.seq_format;
        oc += rc.cigar.size();
        hp_block_record& R = B.records[i];
        R = hp_block_record{};
        R.min_position = rc.a; R.max_position = rc.b; R.read_align = dst; R.read_len = (uint32_t)rc.seq.size(); R.qname_id = rc.qname;
        R.local = &lr; R.read_offset = 0;
        const uint32_t f = (uint32_t)(std::lower_bound(B.hets.begin(), B.hets.end(), rc.a, [](const hp_wfa_variant& v, int64_t p) { return v.position < p; }) - B.hets.begin());
        const uint32_t l = (uint32_t)(std::upper_bound(B.hets.begin(), B.hets.end(), rc.b, [](int64_t p, const hp_wfa_variant& v) { return p < v.position; }) - B.hets.begin());
        if (l > f) { qlo[rc.qname] = std::min(qlo[rc.qname], f); qhi[rc.qname] = std::max(qhi[rc.qname], l); }
    }
    for (uint32_t q = 0; q < B.n_qnames; ++q) if (qhi[q] > qlo[q]) B.seg_cell_cap += qhi[q] - qlo[q];
    B.seg_cell_cap += 16;
}

}  // namespace

struct hp_synth_set {
    hp_synth_reads_spec spec{};
    std::vector<std::unique_ptr<BlockData>> blocks;
    std::vector<hp_block_input> inputs;
    std::vector<uint8_t> qual;   // one quality string shared by every record (a pattern of Q20..Q50)
    uint64_t hets = 0, records = 0, read_bases = 0, qnames = 0, input_bytes = 0;
    uint8_t* arena = nullptr;    // hp_synth_reads_relocate: every block's read bases, back to back
    void (*arena_free)(void*) = nullptr;
    ~hp_synth_set() { if (arena && arena_free) arena_free(arena); }
};

extern "C" void hp_synth_reads_defaults(hp_synth_reads_spec* s) {
    if (!s) return;
    std::memset(s, 0, sizeof *s);
    s->seed = 20250929; s->total_hets = 60000; s->max_block_hets = 4165;
    s->coverage = 30.0; s->read_mean = 15000.0; s->read_sd = 3000.0; s->het_spacing = 1000.0; s->hom_ratio = 0.6;
    s->frac_snv = 0.85; s->frac_indel = 0.12; s->frac_sv = 0.01; s->frac_multiallelic = 0.25;   // the rest (.02): tandem repeats
    s->edit_noise = 0.005; s->noisy_fraction = 0.003; s->noisy_noise = 0.05; s->supplementary_fraction = 0.02;
    s->seq_format = HP_SEQ_BAM4; s->threads = 0;
    s->hifi_sigma = 0.0; s->homopolymer_share = 0.0; s->allele_switch = 0.0;
}

// The same workload with errors shaped like a HiFi run's (docs/performance.md:59-82 quotes HG002 HiFi data): per-read rate lognormal
// around 0.2 % (sigma 0.8: 2 % of the reads beyond 1 %, 0.2 % beyond 2 %, clamp 4 %), half of the errors homopolymer-run indels;
// the separate 5 % "noisy" class is switched off - the lognormal's own tail is the noisy tail.
extern "C" void hp_synth_reads_hifi(hp_synth_reads_spec* s) {
    if (!s) return;
    hp_synth_reads_defaults(s);
    s->edit_noise = 0.002; s->hifi_sigma = 0.8; s->homopolymer_share = 0.5;
    s->noisy_fraction = 0.0;
}

// BASELINE.json configs[4]'s shape (60x HiFi, no down-sampling, multi-allelic sites, the A* frontier under stress) for the WHOLE path:
// twice the rows per het, 15 % of the cells carrying the other haplotype's allele (conflicting rows: every large block prunes,
// astar_phaser.rs:564-585), every tandem-repeat het a 1|2 genotype whose allele0 is itself an ALT (wfa_graph.rs:216-231: an
// allele0 node per site; 22 % of the hets), 1 % of the reads past max_edit_distance. A third of the default set's hets: the
// rows per het double and the search is several times the default's per het.
extern "C" void hp_synth_reads_deep60(hp_synth_reads_spec* s) {
    if (!s) return;
    hp_synth_reads_defaults(s);
    s->total_hets = 20000; s->coverage = 60.0;
    s->frac_snv = 0.62; s->frac_indel = 0.14; s->frac_sv = 0.02; s->frac_multiallelic = 1.0;   // the rest (.22): tandem repeats, all multi-allelic
    s->noisy_fraction = 0.01;
    s->allele_switch = 0.15;
}

extern "C" hp_synth_set* hp_synth_reads_create(const hp_synth_reads_spec* spec, int* status) {
    auto fail = [&](int rc) -> hp_synth_set* { if (status) *status = rc; return nullptr; };
    if (!spec || spec->total_hets < 2 || spec->max_block_hets < 2 || !(spec->coverage > 0) || !(spec->read_mean >= 3000) || !(spec->het_spacing > 0) ||
        spec->frac_snv + spec->frac_indel + spec->frac_sv > 1.0 + 1e-9 || (spec->seq_format != HP_SEQ_ASCII && spec->seq_format != HP_SEQ_BAM4) ||
        !(spec->hifi_sigma >= 0) || !(spec->homopolymer_share >= 0 && spec->homopolymer_share <= 1.0) || !(spec->allele_switch >= 0 && spec->allele_switch <= 0.5))
        return fail(HP_ERR_ARG);
    auto set = std::unique_ptr<hp_synth_set>(new hp_synth_set());
    set->spec = *spec;
    // block sizes: lognormal, median 15, sigma 2.2, capped - until total_hets are reached
    std::vector<uint32_t> sizes;
    {
        Rng r(spec->seed ^ 0xA5A5A5A5DEADBEEFull);
        uint64_t acc = 0;
        while (acc < spec->total_hets) {
            double x = std::exp(std::log(15.0) + 2.2 * r.normal());
            uint32_t n = (uint32_t)std::min<double>(std::max(x, 2.0), (double)spec->max_block_hets);
            n = (uint32_t)std::min<uint64_t>(n, std::max<uint64_t>(2, spec->total_hets - acc));
            sizes.push_back(n);
            acc += n;
        }
    }
    set->blocks.resize(sizes.size());
    set->qual.resize(32768 + 64);
    for (size_t i = 0; i < set->qual.size(); ++i) set->qual[i] = (uint8_t)(20 + (i * 7 + (i >> 5)) % 31);
    unsigned nt = spec->threads ? spec->threads : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    nt = (unsigned)std::min<size_t>(nt, sizes.size());
    std::atomic<size_t> next{0};
    // largest blocks first (they take the longest)
    std::vector<size_t> order(sizes.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return sizes[a] > sizes[b]; });
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= order.size()) return;
            const size_t b = order[k];
            set->blocks[b].reset(new BlockData());
            build_block(set->spec, b, sizes[b], *set->blocks[b]);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    set->inputs.resize(sizes.size());
    for (size_t b = 0; b < sizes.size(); ++b) {
        BlockData& B = *set->blocks[b];
        for (auto& lr : B.locals) lr.qual = set->qual.data();
        hp_block_input& I = set->inputs[b];
        I = hp_block_input{};
        I.block_index = b; I.reference = B.reference.data(); I.ref_base = 0;
        I.n_hets = (uint32_t)B.hets.size(); I.n_homs = (uint32_t)B.homs.size(); I.n_records = (uint32_t)B.records.size(); I.n_qnames = B.n_qnames;
        I.hets = B.hets.data(); I.het_types = B.het_types.data(); I.local_hets = B.local_hets.data(); I.homs = B.homs.data(); I.records = B.records.data();
        I.seq_format = spec->seq_format;
        set->hets += I.n_hets; set->records += I.n_records; set->read_bases += B.read_bases; set->qnames += B.n_qnames;
        set->input_bytes += B.reads.size() + B.reference.size();
    }
    if (status) *status = HP_OK;
    return set.release();
}

extern "C" const hp_block_input* hp_synth_reads_inputs(const hp_synth_set* s, size_t* n_blocks) {
    if (!s) return nullptr;
    if (n_blocks) *n_blocks = s->inputs.size();
    return s->inputs.data();
}

extern "C" void hp_synth_reads_info(const hp_synth_set* s, uint64_t out[8]) {
    if (!s || !out) return;
    out[0] = s->inputs.size(); out[1] = s->hets; out[2] = s->records; out[3] = s->read_bases; out[4] = s->qnames; out[5] = s->input_bytes;
    uint32_t mx = 0;
    for (auto& I : s->inputs) mx = std::max(mx, I.n_hets);
    out[6] = mx; out[7] = 0;
}

extern "C" const uint8_t* hp_synth_reads_truth(const hp_synth_set* s, size_t block) {
    return (s && block < s->blocks.size()) ? s->blocks[block]->truth.data() : nullptr;
}

extern "C" void hp_synth_reads_destroy(hp_synth_set* s) { delete s; }

extern "C" int hp_synth_reads_relocate(hp_synth_set* s, void* (*alloc)(size_t), void (*dealloc)(void*)) {
    if (!s || !alloc || !dealloc || s->arena) return HP_ERR_ARG;
    size_t total = 0;
    for (auto& b : s->blocks) total += (b->reads.size() + 63) & ~(size_t)63;
    uint8_t* a = static_cast<uint8_t*>(alloc(total + 64));
    if (!a) return HP_ERR_OOM;
    size_t off = 0;
    for (auto& bp : s->blocks) {
        BlockData& B = *bp;
        const uint8_t* old = B.reads.data();
        std::memcpy(a + off, old, B.reads.size());
        for (auto& R : B.records) R.read_align = a + off + (R.read_align - old);
        for (auto& L : B.locals) L.seq = a + off + (L.seq - old);
        off += (B.reads.size() + 63) & ~(size_t)63;
        std::vector<uint8_t>().swap(B.reads);   // (the bases live in the arena from here on)
    }
    s->arena = a; s->arena_free = dealloc;
    return HP_OK;
}

