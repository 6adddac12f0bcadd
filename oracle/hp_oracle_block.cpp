// hp_oracle_block.cpp — CPU restatement of `phaser::solve_block` from the decoded records on (reference src/phaser.rs:513-630),
// i.e. of the WHOLE hot path for one phase block: read_parsing::load_full_read_segments (read_parsing.rs:520-637; or
// load_read_segments, :47-113, without global re-alignment) with global_realignment per record (:652-867), the fallback to
// local_realignment and the order-dependent `global_disabled` switch (:556-600), ReadSegment::new / collapse / the
// min_matched_alleles split (:611-629), astar_solver (phaser.rs:541-543), get_solution_span_counts (:546) and haplotag_reads
// for the solver segments and the phasable ones (:614-630).
//
// TEST INFRASTRUCTURE ONLY (see hp_oracle.h): the checker of tests/ and the `cpu_baseline` leg of bench.py; same C structs in
// and out as the product's hp_solve_blocks so that the two can be compared field by field. It is assembled from the pinned
// pieces of this oracle (hpo_wfa_assign, hpo_local_realignment, hpo_read_segment_*, hpo_astar_solve, hpo_solution_span_counts,
// hpo_haplotag_reads); what it adds is the reference's control flow around them, statement by statement.
// PARITY UNPINNED upstream for that control flow: read_parsing.rs holds no tests and solve_block has no read-bearing fixture.
//
// Two things the reference leaves to a hash map are fixed here the way the product fixes them: segments come out in first-seen
// read-name order (`read_groups` is a HashMap<String, _>, read_parsing.rs:542,612 - its iteration order only decides the
// insertion order of an interval tree whose users are order-independent), and a read name is its qname_id.
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "hp_oracle.h"

namespace {

const uint8_t SNV_QUAL = 80, TR_QUAL = 40, SV_INDEL_QUAL = 20, INDEL_QUAL = 10;   // read_parsing.rs:18-22
enum { SNV = 0, INSERTION = 1, DELETION = 2, INDEL = 3, SV_INSERTION = 4, SV_DELETION = 5, TANDEM_REPEAT = 9 };
enum { REFERENCE = 0, ALTERNATE = 1, AMBIGUOUS = 2, NO_OVERLAP = 3 };

struct Row {             // a ReadSegment before / after collapse: full-length vectors + region (read_segments.rs:19-37)
    std::vector<uint8_t> alleles, quals;
    size_t start = 0, end = 0;
};

// ReadSegment::new (read_segments.rs:40-62)
Row row_new(std::vector<uint8_t> alleles, std::vector<uint8_t> quals) {
    Row r;
    hpo_read_segment_new(alleles.data(), alleles.size(), &r.start, &r.end);
    r.alleles = std::move(alleles);
    r.quals = std::move(quals);
    return r;
}

// `read.seq().as_bytes()` (read_parsing.rs:738) for a record handed over as HP_SEQ_BAM4: rust-htslib's DECODE_BASE table
std::vector<uint8_t> decode_seq(const uint8_t* p, uint32_t seq_format, uint64_t first, uint64_t n) {
    std::vector<uint8_t> out(n);
    if (seq_format == HP_SEQ_BAM4) {
        static const char tab[17] = "=ACMGRSVTWYHKDBN";
        for (uint64_t k = 0; k < n; ++k) {
            const uint64_t b = first + k;
            out[k] = (uint8_t)tab[(b & 1u) ? (p[b >> 1] & 15u) : (p[b >> 1] >> 4)];
        }
    } else if (n) std::memcpy(out.data(), p + first, n);
    return out;
}

struct RecResult { std::vector<uint8_t> alleles, quals; bool skipped = false; uint64_t local_aligned = 0; uint64_t wfa_score = 0; int rc = HP_OK;
                   hp_read_stats st{};   // the record's ReadStats (num_alleles and the five per-type arrays; read_parsing.rs:853-862 / :496-502)
};

// local_realignment(&read, variant_calls) (read_parsing.rs:121-503) through the pinned piece
RecResult local_realignment(const hp_block_input* B, uint32_t idx) {
    RecResult r;
    const hp_block_record& rec = B->records[idx];
    if (!rec.local || !B->local_hets) { r.rc = HP_ERR_ARG; return r; }
    hp_local_read lr = *rec.local;
    std::vector<uint8_t> ascii;
    if (lr.seq_format == HP_SEQ_BAM4) { ascii = decode_seq(lr.seq, HP_SEQ_BAM4, 0, lr.seq_len); lr.seq = ascii.data(); lr.seq_format = HP_SEQ_ASCII; }
    r.alleles.assign(B->n_hets, NO_OVERLAP);
    r.quals.assign(B->n_hets, 0);
    hp_read_stats st{};
    const int rc = hpo_local_realignment(&lr, B->local_hets, B->n_hets, r.alleles.data(), r.quals.data(), &st);
    if (rc != 0) { r.rc = HP_ERR_INVARIANT; return r; }
    r.skipped = st.skipped_reads != 0;
    r.local_aligned = st.local_aligned;
    r.st = st;
    return r;
}

// global_realignment (read_parsing.rs:652-867): Ok((alleles, quals, stats, score)) or Err(MaxEditDistance) -> rc = HP_WFA_MAX_ED
RecResult global_realignment(const hp_block_input* B, const hp_block_params* P, uint32_t idx) {
    RecResult r;
    const hp_block_record& rec = B->records[idx];
    const size_t num_variants = B->n_hets;
    const int64_t min_position = rec.min_position, max_position = rec.max_position;   // :672-685 (the caller ran aligned_pairs)
    if (!(max_position >= min_position)) { r.rc = HP_ERR_INVARIANT; return r; }
    // :688-700
    size_t num_overlaps = 0, first_overlap = 0, last_overlap = 0;
    bool have_first = false;
    for (size_t i = 0; i < num_variants; ++i) {
        const int64_t p = B->hets[i].position;
        if (p >= min_position && p < max_position + 1) {
            if (!have_first) { first_overlap = i; have_first = true; }
            last_overlap = i + 1;
            num_overlaps += 1;
        }
    }
    if (num_overlaps == 0) { r.skipped = true; return r; }   // :703-712
    if (num_overlaps != last_overlap - first_overlap) { r.rc = HP_ERR_INVARIANT; return r; }   // :715
    // :718-731
    size_t first_hom = 0, last_hom = 0;
    bool have_hom = false;
    for (size_t i = 0; i < B->n_homs; ++i) {
        const int64_t p = B->homs[i].position;
        if (p >= min_position && p < max_position + 1) {
            if (!have_hom) { first_hom = i; have_hom = true; }
            last_hom = i + 1;
        }
    }
    if (!have_hom) first_hom = 0;
    // :734-742: the part of the read that aligns
    const std::vector<uint8_t> read_align = decode_seq(rec.read_align, B->seq_format, rec.read_offset, rec.read_len);
    // :769-780 graph + WFA, :790-800 traversed nodes -> alleles (hpo_wfa_assign is exactly that for the slice)
    hp_wfa_job job{};
    job.reference = B->reference; job.ref_base = B->ref_base;
    job.ref_start = (uint64_t)min_position; job.ref_end = (uint64_t)max_position + 1;
    job.hets = B->hets + first_overlap; job.n_hets = (uint32_t)(last_overlap - first_overlap);
    job.homs = last_hom > first_hom ? B->homs + first_hom : nullptr; job.n_homs = last_hom > first_hom ? (uint32_t)(last_hom - first_hom) : 0;
    job.read = read_align.data(); job.read_len = (uint32_t)read_align.size();
    hp_wfa_result wr{};
    std::vector<uint8_t> slice(job.n_hets + 1, NO_OVERLAP);
    const int rc = hpo_wfa_assign(&job, P->wfa_prune_distance, P->max_edit_distance, &wr, slice.data());
    if (rc < 0) { r.rc = rc; return r; }
    if (wr.status == HP_WFA_MAX_ED) { r.rc = HP_WFA_MAX_ED; r.wfa_score = wr.score; return r; }   // the `?` of :780
    r.alleles.assign(num_variants, NO_OVERLAP);
    for (size_t k = 0; k < job.n_hets; ++k) r.alleles[first_overlap + k] = slice[k];
    // :803-835
    r.quals.assign(num_variants, 0);
    for (size_t i = 0; i < num_variants; ++i) {
        const uint8_t a = r.alleles[i];
        const size_t vt_index = B->het_types[i];
        if (a == NO_OVERLAP) continue;                                                   // :807-808
        if (a == AMBIGUOUS) { if (vt_index < HP_N_VARIANT_TYPES) r.st.failed_matches[vt_index] += 1; continue; }   // :809-811
        uint8_t q;
        switch (B->het_types[i]) {
            case SNV: q = SNV_QUAL; break;
            case DELETION: case INSERTION: case INDEL: q = INDEL_QUAL; break;
            case SV_DELETION: case SV_INSERTION: q = SV_INDEL_QUAL; break;
            case TANDEM_REPEAT: q = TR_QUAL; break;
            default: r.rc = HP_ERR_INVARIANT; return r;   // panic!("No implementation for matching ...")
        }
        r.quals[i] = (uint8_t)(2 * q);
        r.st.inexact_matches[vt_index] += 1;                                             // :838-843: `exact_allele` is false
        if (a == REFERENCE) r.st.allele0_matches[vt_index] += 1; else r.st.allele1_matches[vt_index] += 1;   // :844-848
        r.st.num_alleles += 1;                                                           // :849
    }
    r.wfa_score = wr.score;
    r.local_aligned = 0;   // ReadStats::new(..., 1, 0): global_aligned 1, local_aligned 0 (:857-862)
    return r;
}

}  // namespace

extern "C" int hpo_solve_block(const hp_block_input* B, const hp_block_params* P, hp_block_output* O) {
    if (!B || !P || !O || B->n_hets == 0) return HP_ERR_ARG;
    const size_t N = B->n_hets;
    // ---- load_full_read_segments (read_parsing.rs:520-637) / load_read_segments (:47-113) ----
    std::vector<std::vector<Row>> read_groups(B->n_qnames);
    std::vector<uint32_t> order;                     // first-seen read names
    std::vector<uint8_t> seen(B->n_qnames, 0);
    uint64_t num_reads = 0, skipped_reads = 0, global_aligned = 0, local_aligned = 0;
    hp_read_stats joint{};                            // num_alleles + the per-type arrays of joint_stats
    std::vector<uint64_t> edit_distances;
    bool global_disabled = false;
    double num_global_failures = 0.0, total_parsed = 0.0;
    for (uint32_t idx = 0; idx < B->n_records; ++idx) {
        const hp_block_record& rec = B->records[idx];
        if (rec.qname_id >= B->n_qnames) return HP_ERR_ARG;
        RecResult r;
        if (!P->global_realignment) r = local_realignment(B, idx);          // :76
        else if (global_disabled) {                                         // :556-559
            r = local_realignment(B, idx);
            r.wfa_score = P->max_edit_distance;
        } else {
            r = global_realignment(B, P, idx);                              // :562
            if (r.rc == HP_WFA_MAX_ED) {                                    // :564-575: Err(MaxEditDistance { distance })
                const uint64_t distance = r.wfa_score;
                r = local_realignment(B, idx);
                r.wfa_score = distance;
            }
        }
        if (r.rc != HP_OK) return r.rc;
        joint.num_alleles += r.st.num_alleles;                              // :607 / :88: joint_stats += read_stats, skipped or not
        for (int t = 0; t < HP_N_VARIANT_TYPES; ++t) {
            joint.exact_matches[t] += r.st.exact_matches[t]; joint.inexact_matches[t] += r.st.inexact_matches[t]; joint.failed_matches[t] += r.st.failed_matches[t];
            joint.allele0_matches[t] += r.st.allele0_matches[t]; joint.allele1_matches[t] += r.st.allele1_matches[t];
        }
        if (!r.skipped) {                                                   // :583-600
            if (!seen[rec.qname_id]) { seen[rec.qname_id] = 1; order.push_back(rec.qname_id); }
            read_groups[rec.qname_id].push_back(row_new(std::move(r.alleles), std::move(r.quals)));
            if (P->global_realignment) {
                edit_distances.push_back(r.wfa_score);
                num_global_failures += (double)r.local_aligned;
                total_parsed += 1.0;
                if (!global_disabled && num_global_failures >= (double)P->global_failure_minimum && num_global_failures / total_parsed >= P->global_failure_ratio)
                    global_disabled = true;
            }
            local_aligned += r.local_aligned;
            global_aligned += 1 - r.local_aligned;
        } else skipped_reads += 1;                                          // :602-608 (joint_stats += read_stats)
    }
    // ---- collapse + split (:611-629) ----
    struct Seg { uint32_t qname; Row row; bool solver; };
    std::vector<Seg> segs;
    for (uint32_t q : order) {
        const std::vector<Row>& grp = read_groups[q];
        Row col;
        if (grp.size() == 1) col = grp[0];                                  // read_segments.rs:72-75
        else {
            std::vector<uint8_t> al(grp.size() * N), ql(grp.size() * N);
            for (size_t k = 0; k < grp.size(); ++k) { std::memcpy(al.data() + k * N, grp[k].alleles.data(), N); std::memcpy(ql.data() + k * N, grp[k].quals.data(), N); }
            col.alleles.assign(N, NO_OVERLAP); col.quals.assign(N, 0);
            if (hpo_read_segment_collapse(al.data(), ql.data(), grp.size(), N, col.alleles.data(), col.quals.data(), &col.start, &col.end) != 0) return HP_ERR_INVARIANT;
        }
        size_t num_set = 0;
        for (size_t i = 0; i < N; ++i) num_set += col.alleles[i] < AMBIGUOUS;   // get_num_set (read_segments.rs:141-146)
        if (num_set >= P->min_matched_alleles) { num_reads += grp.size(); segs.push_back(Seg{q, std::move(col), true}); }
        else {
            skipped_reads += grp.size();
            if (num_set > 0) segs.push_back(Seg{q, std::move(col), false});
        }
    }
    // ---- the solver matrix (phaser.rs:514-533 hands the interval tree over) as hp_block_view ----
    auto make_view = [&](bool solver, std::vector<uint32_t>& rs, std::vector<uint32_t>& re, std::vector<uint64_t>& ro, std::vector<uint8_t>& a2,
                         std::vector<uint8_t>& qv, std::vector<uint8_t>& flags, hp_block_view& v) {
        uint64_t cells = 0;
        ro.push_back(0);
        for (const Seg& s : segs) {
            if (s.solver != solver) continue;
            rs.push_back((uint32_t)s.row.start); re.push_back((uint32_t)s.row.end);
            for (size_t i = s.row.start; i < s.row.end; ++i, ++cells) {
                if ((cells >> 2) >= a2.size()) a2.push_back(0);
                a2[cells >> 2] |= (uint8_t)(s.row.alleles[i] << (2 * (cells & 3)));
                qv.push_back(s.row.quals[i]);
            }
            ro.push_back(cells);
        }
        if (a2.empty()) a2.push_back(0);
        if (qv.empty()) qv.push_back(0);
        if (rs.empty()) { rs.push_back(0); re.push_back(0); }
        flags.resize(N);
        for (size_t i = 0; i < N; ++i) flags[i] = (uint8_t)(((B->hets[i].flags & 1u) ? HP_VAR_IGNORED : 0) | (B->het_types[i] == SNV ? HP_VAR_SNV : 0));
        v.n_variants = (uint32_t)N; v.n_reads = (uint32_t)(ro.size() - 1);
        v.read_start = rs.data(); v.read_end = re.data(); v.row_off = ro.data(); v.alleles_2bit = a2.data(); v.quals = qv.data(); v.var_flags = flags.data();
    };
    std::vector<uint32_t> rs, re, prs, pre;
    std::vector<uint64_t> ro, pro;
    std::vector<uint8_t> a2, qv, fl, pa2, pqv, pfl;
    hp_block_view view{}, pview{};
    make_view(true, rs, re, ro, a2, qv, fl, view);
    make_view(false, prs, pre, pro, pa2, pqv, pfl, pview);
    // ---- astar_solver (phaser.rs:541-543), span counts (:546), haplotags (:614-630) ----
    std::vector<uint8_t> h1(N), h2(N);
    hp_astar_params ap = P->astar;
    ap.block_index = B->block_index;
    hp_phase_stats stats{};
    int rc = hpo_astar_solve(&view, &ap, h1.data(), h2.data(), &stats, nullptr, nullptr);
    if (rc != HP_OK) return rc;
    std::vector<uint64_t> spans(N > 1 ? N - 1 : 1, 0);
    if (N > 1 && (rc = hpo_solution_span_counts(&view, h1.data(), h2.data(), spans.data())) != HP_OK) return rc;
    std::vector<uint64_t> ident(N);
    for (size_t i = 0; i < N; ++i) ident[i] = i;     // block_tags[i] := i, so that phase_block comes back as the first resolved het
    std::vector<uint8_t> tag(view.n_reads + 1, 2), ptag(pview.n_reads + 1, 2);
    std::vector<uint64_t> first(view.n_reads + 1, 0), pfirst(pview.n_reads + 1, 0);
    if (view.n_reads && (rc = hpo_haplotag_reads(&view, h1.data(), h2.data(), ident.data(), tag.data(), first.data())) != HP_OK) return rc;
    if (pview.n_reads && (rc = hpo_haplotag_reads(&pview, h1.data(), h2.data(), ident.data(), ptag.data(), pfirst.data())) != HP_OK) return rc;
    // ---- outputs (hp_block_output, as hp_solve_blocks fills it) ----
    if (O->h1) std::memcpy(O->h1, h1.data(), N);
    if (O->h2) std::memcpy(O->h2, h2.data(), N);
    O->stats = stats;
    if (O->span_counts && N > 1) std::memcpy(O->span_counts, spans.data(), (N - 1) * 8);
    O->n_segments = (uint32_t)segs.size();
    O->n_solver = view.n_reads;
    O->num_reads = num_reads; O->skipped_reads = skipped_reads; O->global_aligned = global_aligned; O->local_aligned = local_aligned;
    O->num_alleles = joint.num_alleles;
    for (int t = 0; t < HP_N_VARIANT_TYPES; ++t) {
        O->exact_matches[t] = joint.exact_matches[t]; O->inexact_matches[t] = joint.inexact_matches[t]; O->failed_matches[t] = joint.failed_matches[t];
        O->allele0_matches[t] = joint.allele0_matches[t]; O->allele1_matches[t] = joint.allele1_matches[t];
    }
    O->n_edit_distances = edit_distances.size();
    if (O->edit_distances && !edit_distances.empty()) std::memcpy(O->edit_distances, edit_distances.data(), edit_distances.size() * 8);
    uint64_t cells = 0;
    uint32_t ks = 0, kp = 0;
    for (size_t k = 0; k < segs.size(); ++k) {
        const Seg& s = segs[k];
        if (O->seg_qname) O->seg_qname[k] = s.qname;
        if (O->seg_start) O->seg_start[k] = (uint32_t)s.row.start;
        if (O->seg_end) O->seg_end[k] = (uint32_t)s.row.end;
        if (O->seg_solver) O->seg_solver[k] = s.solver ? 1 : 0;
        const uint8_t ht = s.solver ? tag[ks] : ptag[kp];
        const uint64_t fh = s.solver ? first[ks] : pfirst[kp];
        if (s.solver) ++ks; else ++kp;
        if (O->seg_haplotag) O->seg_haplotag[k] = ht;
        if (O->seg_first_het) O->seg_first_het[k] = ht == 2 ? UINT32_MAX : (uint32_t)fh;
        if (O->seg_row_off) O->seg_row_off[k] = cells;
        const uint64_t len = s.row.end - s.row.start;
        if (O->seg_alleles || O->seg_quals) {
            if (cells + len > O->seg_cell_cap) return HP_ERR_ARG;
            if (O->seg_alleles && len) std::memcpy(O->seg_alleles + cells, s.row.alleles.data() + s.row.start, len);
            if (O->seg_quals && len) std::memcpy(O->seg_quals + cells, s.row.quals.data() + s.row.start, len);
        }
        cells += len;
    }
    if (O->seg_row_off) O->seg_row_off[segs.size()] = cells;
    O->status = HP_OK;
    return HP_OK;
}
