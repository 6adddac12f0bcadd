// hp_local.hip — hp_local_realign_batch: the reference's `local_realignment` (src/read_parsing.rs:121-503) for a
// batch of records. The order-dependent, branchy per-variant coordinate logic stays on the host (threads over
// records); what the reference spends its time on in this mode — two Levenshtein distances per inexact allele
// (Variant::closest_allele_clip, src/data_types/variants.rs:624-641) — is collected for the WHOLE batch and
// solved by one hp_edit_distance_batch launch (one wavefront per pair, hp_edit.hip).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "hp_common.h"

#include <ctime>

namespace hp {
namespace {

// VariantType repr (variants.rs:10-33)
enum : uint32_t { VT_SNV = 0, VT_INS = 1, VT_DEL = 2, VT_INDEL = 3, VT_SVINS = 4, VT_SVDEL = 5, VT_TR = 9, VT_UNKNOWN = 10 };
// AlleleType repr (read_segments.rs:5-16)
enum : uint8_t { A_REF = 0, A_ALT = 1, A_AMBIGUOUS = 2, A_NOOVERLAP = 3 };
// read_parsing.rs:18-22
constexpr double SNV_QUAL = 80, TR_QUAL = 40, SV_INDEL_QUAL = 20, INDEL_QUAL = 10;
constexpr uint8_t MISSING_QUAL = 0;

constexpr uint8_t F_EXACT = 1, F_OVERLAPS = 2, F_PENDING = 4;

// Rust `f64 as u8`: saturating, NaN -> 0
inline uint8_t as_u8(double x) {
    if (!(x == x)) return 0;
    if (x <= 0.0) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)x;
}

struct Pending { uint32_t read, var, ss, se, head, tail; };

struct Worker {
    std::vector<Pending> pending;
    std::vector<int32_t> lookup;
    int rc = HP_OK;
    std::string err;
};

// One record: everything of local_realignment except the edit distances. `al`, `ql`, `fl` are the record's rows.
int realign_one(const hp_local_read& rd, uint32_t ri, const hp_local_variant* vars, size_t nv, uint8_t* al, uint8_t* ql,
                uint8_t* fl, Worker& w) {
    // reference coordinate -> read coordinate (read_parsing.rs:136-147); rust-htslib 0.39.5 `aligned_pairs`
    // yields (qpos, rpos) for M/=/X only, I/S advance the read, D/N the reference, H nothing, P panics.
    uint64_t ref_span = 0;
    for (uint32_t c = 0; c < rd.n_cigar; ++c) {
        const uint32_t op = rd.cigar[c] & 0xF, len = rd.cigar[c] >> 4;
        if (op == 0 || op == 7 || op == 8 || op == 2 || op == 3) ref_span += len;
        else if (op == 6) { w.err = "record " + std::to_string(ri) + ": CIGAR Pad is not supported (rust-htslib aligned_pairs panics)"; return HP_ERR_UNSUPPORTED; }
        else if (op > 8) { w.err = "record " + std::to_string(ri) + ": invalid CIGAR op"; return HP_ERR_ARG; }
    }
    w.lookup.assign((size_t)std::max<uint64_t>(ref_span, 1), -1);
    const int64_t min_position = rd.pos;
    int64_t max_position = rd.pos;
    {
        uint64_t q = 0, r = 0;
        for (uint32_t c = 0; c < rd.n_cigar; ++c) {
            const uint32_t op = rd.cigar[c] & 0xF, len = rd.cigar[c] >> 4;
            if (op == 0 || op == 7 || op == 8) {
                if (q + len > rd.seq_len) { w.err = "record " + std::to_string(ri) + ": CIGAR consumes more bases than the sequence holds"; return HP_ERR_ARG; }
                for (uint32_t k = 0; k < len; ++k) w.lookup[r + k] = (int32_t)(q + k);
                if (len) max_position = std::max<int64_t>(max_position, min_position + (int64_t)(r + len - 1));
                q += len; r += len;
            } else if (op == 1 || op == 4) q += len;
            else if (op == 2 || op == 3) r += len;
        }
    }
    const int32_t* lk = w.lookup.data();
    auto get = [&](int64_t c) -> int32_t { return (c < min_position || c > max_position) ? -1 : lk[c - min_position]; };
    auto in_range = [&](int64_t c) { return c >= min_position && c < max_position + 1; };  // aligned_range (:150)

    int64_t last_deletion_end = 0;
    for (size_t vi = 0; vi < nv; ++vi) {
        const hp_local_variant& v = vars[vi];
        const int64_t vpos = v.position;
        uint8_t allele, qual = MISSING_QUAL, flags = 0;
        if (v.flags & HP_VAR_IGNORED) {                      // :180-186
            allele = A_NOOVERLAP;
        } else if (vpos < last_deletion_end) {               // :187-195
            allele = A_AMBIGUOUS; flags = F_OVERLAPS;
        } else if (v.variant_type == VT_SVDEL) {             // :354-452
            if (in_range(vpos)) {
                const int64_t last_start = vpos + 1, first_end = vpos + (int64_t)v.ref_len;
                if (in_range(first_end)) {
                    const int64_t expected_deleted = first_end - last_start;
                    int64_t start_anchor = last_start;
                    while (get(start_anchor) < 0) {
                        if (start_anchor <= min_position) break;     // :369-374
                        start_anchor -= 1;
                    }
                    int64_t end_anchor = first_end;
                    while (get(end_anchor) < 0) {
                        end_anchor += 1;
                        if (end_anchor >= max_position + 1) break;   // :380-384
                    }
                    int64_t deleted = 0;
                    for (int64_t dc = start_anchor; dc < end_anchor; ++dc) deleted += get(dc) < 0 ? 1 : 0;
                    const double match_window = 0.33;
                    const double ratio = (double)deleted / (double)expected_deleted;
                    if (ratio < match_window) {
                        allele = A_REF;
                        qual = as_u8(std::fmax(SV_INDEL_QUAL * (1.0 - ratio), 1.0));
                        if (ratio == 0.0) flags |= F_EXACT;
                    } else if (std::fabs(1.0 - ratio) < match_window) {
                        allele = A_ALT;
                        qual = as_u8(std::fmax(SV_INDEL_QUAL * (1.0 - std::fabs(1.0 - ratio)), 1.0));
                        if (ratio == 1.0) flags |= F_EXACT;
                        last_deletion_end = first_end;               // :433
                    } else {
                        allele = A_AMBIGUOUS;
                    }
                } else {
                    allele = A_AMBIGUOUS;                            // partial overlap (:441-447)
                }
                flags |= F_OVERLAPS;
            } else {
                allele = A_NOOVERLAP;
            }
        } else if (v.variant_type == VT_SNV || v.variant_type == VT_INS || v.variant_type == VT_DEL ||
                   v.variant_type == VT_INDEL || v.variant_type == VT_SVINS || v.variant_type == VT_TR) {  // :197-353
            const int64_t prefix = v.prefix_len, postfix = v.postfix_len;
            const int64_t first_start = vpos - prefix, last_start = vpos + 1;
            const int64_t first_end = vpos + (int64_t)v.ref_len, last_end = first_end + postfix + 1;
            int64_t closest_start = -1, closest_end = -1;
            // nothing of the window is aligned: every lookup below misses (the common case for a block's far variants)
            if (last_end > min_position && first_start <= max_position) {
                for (int64_t sc = last_start - 1; sc >= first_start; --sc) { const int32_t si = get(sc); if (si >= 0) { closest_start = si; break; } }
                for (int64_t ec = first_end; ec < last_end; ++ec) { const int32_t ei = get(ec); if (ei >= 0) { closest_end = ei; break; } }
            }
            int64_t start_c = -1, end_c = -1, start_clip = 0, end_clip = 0;
            if (closest_start >= 0 && closest_end >= 0) {
                for (int64_t sc = first_start; sc < last_start; ++sc) {
                    start_clip += 1;
                    const int32_t si = get(sc);
                    if (si < 0) continue;
                    if (closest_start - si > 2 * prefix) continue;       // too far away (:245)
                    start_c = si;
                    for (int64_t ec = last_end - 1; ec >= first_end; --ec) {
                        end_clip += 1;
                        const int32_t ni = get(ec);
                        if (ni < 0) continue;
                        if (ni - closest_end > 2 * postfix) continue;    // :259
                        end_c = ni;
                        break;
                    }
                    break;
                }
            }
            if (start_c >= 0) {
                if (end_c >= 0) {
                    const uint32_t ss = (uint32_t)start_c, se = (uint32_t)end_c;
                    const uint32_t ol = se - ss;
                    const uint8_t* obs = rd.seq + ss;
                    if (ol == v.allele0_len && std::memcmp(obs, v.allele0, ol) == 0) { allele = A_REF; flags |= F_EXACT; }        // match_allele (variants.rs:598)
                    else if (ol == v.allele1_len && std::memcmp(obs, v.allele1, ol) == 0) { allele = A_ALT; flags |= F_EXACT; }
                    else {
                        allele = A_AMBIGUOUS;   // decided by closest_allele_clip once the distances are back
                        flags |= F_PENDING;
                        w.pending.push_back(Pending{ri, (uint32_t)vi, ss, se, (uint32_t)(start_clip - 1), (uint32_t)(end_clip - 1)});
                    }
                    // harmonic mean of the base qualities scales the baseline (:293-327)
                    double inv = 0.0;
                    for (uint32_t k = ss; k < se; ++k) inv += 1.0 / (double)rd.qual[k];
                    const double harmonic = (double)ol / inv;
                    const double factor = std::fmin(harmonic / 40.0, 1.0);
                    double base;
                    switch (v.variant_type) {
                        case VT_SNV: base = SNV_QUAL; break;
                        case VT_DEL: case VT_INS: case VT_INDEL: base = INDEL_QUAL; break;
                        case VT_SVINS: base = SV_INDEL_QUAL; break;
                        default: base = TR_QUAL; break;
                    }
                    qual = as_u8(std::fmax(base * factor, 1.0));
                    flags |= F_OVERLAPS;
                } else {
                    allele = A_AMBIGUOUS; flags = F_OVERLAPS;         // :331-337
                }
            } else if (in_range(vpos)) {
                allele = A_AMBIGUOUS; flags = F_OVERLAPS;             // :340-343
            } else {
                allele = A_NOOVERLAP;
            }
        } else {
            w.err = "variant " + std::to_string(vi) + ": unhandled variant type " + std::to_string(v.variant_type) + " (read_parsing.rs:455 panics)";
            return HP_ERR_INVARIANT;
        }
        al[vi] = allele;
        ql[vi] = qual;
        fl[vi] = flags;
    }
    return HP_OK;
}

}  // namespace
}  // namespace hp

using namespace hp;

// local_realignment for the records of SEVERAL blocks at once (hp_block.hip: the fallbacks of a whole block set): the
// coordinate logic per (group, record) on host threads, then ONE Levenshtein launch for every inexact allele of every group.
int hp::local_realign_groups(LocalGroup* groups, size_t n_groups, int device_id) {
    if (n_groups == 0) return HP_OK;
    if (!groups) { set_error("null argument"); return HP_ERR_ARG; }
    const bool verbose = std::getenv("HP_DEBUG") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t0 = now_ms();
    // ---- validation; records handed over in the BAM's own 4-bit encoding (HP_SEQ_BAM4) are decoded here, exactly as
    // read.seq().as_bytes() does (read_parsing.rs:151): a few records per block (the fallbacks), or everything in local mode ----
    // (the decode itself happens per record on the host threads below: a few hundred 15-kb reads per block set were 3 ms of one thread)
    struct GroupState { std::vector<hp_local_read> reads; std::vector<std::unique_ptr<uint8_t[]>> seqs; const hp_local_read* rd = nullptr; std::vector<uint8_t> flags; size_t first_item = 0; };
    std::vector<GroupState> gs(n_groups);
    size_t n_items = 0;
    for (size_t g = 0; g < n_groups; ++g) {
        LocalGroup& G = groups[g];
        GroupState& S = gs[g];
        S.first_item = n_items;
        n_items += G.n_reads;
        if (G.n_reads == 0) continue;
        if (!G.reads || (G.n_variants && (!G.variants || !G.alleles || !G.quals))) { set_error("null argument"); return HP_ERR_ARG; }
        if (G.n_reads > 0x7FFFFFFFull || G.n_variants > 0x7FFFFFFFull) { set_error("batch too large"); return HP_ERR_ARG; }
        S.rd = G.reads;
        for (size_t r = 0; r < G.n_reads; ++r) {
            const hp_local_read& R = G.reads[r];
            if (R.seq_format == HP_SEQ_ASCII) continue;
            if (R.seq_format != HP_SEQ_BAM4) { set_error("record %zu: unknown seq_format %u", r, R.seq_format); return HP_ERR_ARG; }
            if (S.reads.empty()) { S.reads.assign(G.reads, G.reads + G.n_reads); S.seqs.resize(G.n_reads); }
            if (R.seq_len && !R.seq) { set_error("record %zu: null buffer", r); return HP_ERR_ARG; }
            S.seqs[r].reset(new uint8_t[(size_t)R.seq_len + 1]);   // (filled by the thread that takes the record)
        }
        if (!S.reads.empty()) S.rd = S.reads.data();
        for (size_t i = 0; i < G.n_variants; ++i) {
            const hp_local_variant& v = G.variants[i];
            if (v.variant_type > VT_UNKNOWN) { set_error("variant %zu: invalid variant_type %u", i, v.variant_type); return HP_ERR_ARG; }
            if ((uint64_t)v.prefix_len + v.postfix_len > std::min(v.allele0_len, v.allele1_len) || v.position < (int64_t)v.prefix_len ||
                (v.allele0_len && !v.allele0) || (v.allele1_len && !v.allele1)) {
                set_error("variant %zu: alleles shorter than prefix + postfix, prefix reaching below coordinate 0, or null allele", i);
                return HP_ERR_ARG;
            }
        }
        for (size_t r = 0; r < G.n_reads; ++r) {
            if ((S.rd[r].n_cigar && !S.rd[r].cigar) || (S.rd[r].seq_len && (!S.rd[r].seq || !S.rd[r].qual)) || S.rd[r].pos < 0) {
                set_error("record %zu: null buffer or negative position", r);
                return HP_ERR_ARG;
            }
        }
        S.flags.assign(G.n_reads * G.n_variants, 0);
    }
    if (n_items == 0) return HP_OK;
    // ---- coordinates: host threads over (group, record) ----
    unsigned nt = host_threads(16u);
    if (const char* e = std::getenv("HP_LOCAL_HOST_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
    nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)nt, (size_t)16, n_items / 32 + 1}));
    struct GWorker { Worker w; std::vector<uint32_t> pend_group; };   // group of each pending entry, in order
    std::vector<GWorker> workers(nt);
    auto body = [&](unsigned t) {
        GWorker& W = workers[t];
        const size_t lo = n_items * t / nt, hi = n_items * (t + 1) / nt;
        for (size_t it = lo; it < hi && W.w.rc == HP_OK; ++it) {
            // the group of item `it`: the last one that starts at or before it (empty groups share their start with the next)
            size_t a = 0, b = n_groups;
            while (b - a > 1) { const size_t m = (a + b) / 2; if (gs[m].first_item <= it) a = m; else b = m; }
            const size_t g = a, r = it - gs[g].first_item;
            const LocalGroup& G = groups[g];
            if (!gs[g].seqs.empty() && gs[g].seqs[r]) {   // a record in the BAM's 4-bit codes: decoded here, by the thread that needs it
                hp_local_read& R = gs[g].reads[r];
                decode_bam4(R.seq, 0, R.seq_len, gs[g].seqs[r].get());
                R.seq = gs[g].seqs[r].get();
                R.seq_format = HP_SEQ_ASCII;
            }
            W.w.rc = realign_one(gs[g].rd[r], (uint32_t)r, G.variants, G.n_variants, G.alleles + r * G.n_variants, G.quals + r * G.n_variants,
                                 gs[g].flags.data() + r * G.n_variants, W.w);
            W.pend_group.resize(W.w.pending.size(), (uint32_t)g);
        }
    };
    if (nt == 1) body(0);
    else WorkerPool::get().run(nt, body);
    const double t1 = now_ms();
    size_t n_pending = 0;
    for (auto& W : workers) {
        if (W.w.rc != HP_OK) { set_error("%s", W.w.err.c_str()); return W.w.rc; }
        n_pending += W.w.pending.size();
    }
    // ---- every inexact allele of every group: d0, d1 = edit_distance(obs, allele{0,1}[head .. len - tail]) in one launch ----
    if (n_pending) {
        std::vector<hp_ed_pair> pairs;
        pairs.reserve(2 * n_pending);
        for (auto& W : workers)
            for (size_t k = 0; k < W.w.pending.size(); ++k) {
                const Pending& p = W.w.pending[k];
                const uint32_t g = W.pend_group[k];
                const hp_local_variant& v = groups[g].variants[p.var];
                const uint8_t* obs = gs[g].rd[p.read].seq + p.ss;
                pairs.push_back(hp_ed_pair{obs, v.allele0 + p.head, p.se - p.ss, v.allele0_len - p.head - p.tail});
                pairs.push_back(hp_ed_pair{obs, v.allele1 + p.head, p.se - p.ss, v.allele1_len - p.head - p.tail});
            }
        std::vector<uint64_t> dist(pairs.size());
        const int rc = hp_edit_distance_batch(pairs.data(), pairs.size(), dist.data(), device_id);
        if (rc != HP_OK) return rc;
        size_t k2 = 0;
        for (auto& W : workers)
            for (size_t k = 0; k < W.w.pending.size(); ++k) {
                const Pending& p = W.w.pending[k];
                const LocalGroup& G = groups[W.pend_group[k]];
                const uint64_t d0 = dist[k2], d1 = dist[k2 + 1];
                k2 += 2;
                G.alleles[(size_t)p.read * G.n_variants + p.var] = d0 < d1 ? A_REF : (d0 > d1 ? A_ALT : A_AMBIGUOUS);  // variants.rs:633-640
            }
    }
    const double t2 = now_ms();
    // ---- statistics (read_parsing.rs:460-499) ----
    for (size_t g = 0; g < n_groups; ++g) {
        const LocalGroup& G = groups[g];
        if (!G.stats) continue;
        for (size_t r = 0; r < G.n_reads; ++r) {
            hp_read_stats s{};
            uint64_t overlaps = 0;
            for (size_t vi = 0; vi < G.n_variants; ++vi) {
                const uint8_t f = gs[g].flags[r * G.n_variants + vi], a = G.alleles[r * G.n_variants + vi];
                if (!(f & F_OVERLAPS)) continue;
                const uint32_t t = G.variants[vi].variant_type;
                if (a == A_AMBIGUOUS) { s.failed_matches[t] += 1; continue; }
                if (f & F_EXACT) s.exact_matches[t] += 1; else s.inexact_matches[t] += 1;
                if (a == A_REF) s.allele0_matches[t] += 1; else s.allele1_matches[t] += 1;
                overlaps += 1;
            }
            s.num_alleles = overlaps;
            s.skipped_reads = overlaps == 0 ? 1 : 0;
            s.local_aligned = 1 - s.skipped_reads;
            G.stats[r] = s;
        }
    }
    if (verbose) { fprintf(stderr, "[hp] local re-alignment of %zu records in %zu groups on %u threads: coordinates %.2f ms, %zu inexact alleles on the device %.2f ms, statistics %.2f ms\n", n_items, n_groups, nt, t1 - t0, n_pending, t2 - t1, now_ms() - t2); fflush(stderr); }
    return HP_OK;
}

extern "C" int hp_local_realign_batch(const hp_local_read* reads, size_t n_reads, const hp_local_variant* variants,
                                      size_t n_variants, uint8_t* alleles, uint8_t* quals, hp_read_stats* stats, int device_id) {
    if (n_reads == 0) return HP_OK;
    LocalGroup g{reads, n_reads, variants, n_variants, alleles, quals, stats};
    return local_realign_groups(&g, 1, device_id);
}
