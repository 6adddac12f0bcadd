// hp_oracle_local.cpp — CPU restatement of `local_realignment` (reference src/read_parsing.rs:121-503) and of the
// `Variant` methods it calls (src/data_types/variants.rs:598-641). TEST INFRASTRUCTURE ONLY (see hp_oracle.h).
//
// PARITY UNPINNED upstream: read_parsing.rs holds no tests. What IS pinned: the closest_allele / match_allele
// triples of variants.rs:780-846 (tests/golden/sequence_alignment.json) through hpo_closest_allele_clip below.
// The BAM record is replaced by its (pos, CIGAR, seq, qual) view; `aligned_pairs` follows rust-htslib 0.39.5
// (Cargo.lock:1069; source not under /root/reference): M/=/X yield pairs, I/S advance the read, D/N the reference,
// H advances nothing, P panics.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "hp_oracle.h"

namespace {

struct VariantView {  // variants.rs:67-94 (the fields local_realignment touches)
    const hp_local_variant* v;
    int64_t position() const { return v->position; }
    uint32_t get_type() const { return v->variant_type; }
    size_t get_ref_len() const { return v->ref_len; }
    size_t get_prefix_len() const { return v->prefix_len; }
    size_t get_postfix_len() const { return v->postfix_len; }
    bool is_ignored() const { return (v->flags & HP_VAR_IGNORED) != 0; }
    // variants.rs:598-606
    uint8_t match_allele(const uint8_t* a, size_t n) const {
        if (n == v->allele0_len && std::memcmp(a, v->allele0, n) == 0) return 0;
        if (n == v->allele1_len && std::memcmp(a, v->allele1, n) == 0) return 1;
        return 2;
    }
    // variants.rs:624-641 -> (allele, min distance, other distance); -1 on a failed assert
    int closest_allele_clip(const uint8_t* a, size_t n, size_t head_clip, size_t tail_clip, uint64_t* dmin, uint64_t* dother) const {
        if (head_clip > v->prefix_len || tail_clip > v->postfix_len) return -1;
        const uint64_t d0 = hpo_edit_distance(a, n, v->allele0 + head_clip, v->allele0_len - tail_clip - head_clip);
        const uint64_t d1 = hpo_edit_distance(a, n, v->allele1 + head_clip, v->allele1_len - tail_clip - head_clip);
        if (d0 < d1) { *dmin = d0; *dother = d1; return 0; }
        if (d0 > d1) { *dmin = d1; *dother = d0; return 1; }
        *dmin = d0; *dother = d1;
        return 2;
    }
};

// Rust `as u8` from f64: saturating, NaN -> 0
uint8_t f64_as_u8(double x) {
    if (std::isnan(x)) return 0;
    if (x < 0.0) return 0;
    if (x > 255.0) return 255;
    return (uint8_t)x;
}

enum { SNV = 0, INSERTION = 1, DELETION = 2, INDEL = 3, SV_INSERTION = 4, SV_DELETION = 5, TANDEM_REPEAT = 9 };
enum { REFERENCE = 0, ALTERNATE = 1, AMBIGUOUS = 2, NO_OVERLAP = 3 };
const uint8_t SNV_QUAL = 80, TR_QUAL = 40, SV_INDEL_QUAL = 20, INDEL_QUAL = 10, MISSING_QUAL = 0;

}  // namespace

extern "C" {

int hpo_closest_allele_clip(const hp_local_variant* variant, const uint8_t* allele, size_t len, size_t head_clip,
                            size_t tail_clip, uint64_t* dmin, uint64_t* dother) {
    return VariantView{variant}.closest_allele_clip(allele, len, head_clip, tail_clip, dmin, dother);
}

int hpo_match_allele(const hp_local_variant* variant, const uint8_t* allele, size_t len) {
    return VariantView{variant}.match_allele(allele, len);
}

// read_parsing.rs:121-503. Returns 0, or -3 where the reference would panic.
int hpo_local_realignment(const hp_local_read* read, const hp_local_variant* variant_calls, size_t num_variants,
                          uint8_t* alleles, uint8_t* quals, hp_read_stats* out_stats) {
    uint64_t num_alleles = 0;
    hp_read_stats st;
    std::memset(&st, 0, sizeof st);

    // reference coordinate -> sequence coordinate (:136-147)
    std::unordered_map<int64_t, int64_t> coordinate_lookup;
    const int64_t min_position = read->pos;
    int64_t max_position = read->pos;
    {
        int64_t qpos = 0, rpos = read->pos;
        for (uint32_t c = 0; c < read->n_cigar; ++c) {
            const uint32_t op = read->cigar[c] & 0xF, len = read->cigar[c] >> 4;
            switch (op) {
                case 0: case 7: case 8:
                    for (uint32_t k = 0; k < len; ++k) {
                        coordinate_lookup[rpos] = qpos;
                        if (rpos > max_position) max_position = rpos;
                        ++qpos; ++rpos;
                    }
                    break;
                case 1: case 4: qpos += len; break;
                case 2: case 3: rpos += len; break;
                case 5: break;
                default: return -3;  // Cigar::Pad panics
            }
        }
    }
    if (max_position < min_position) return -3;
    const int64_t range_start = min_position, range_end = max_position + 1;  // aligned_range (:150)
    auto range_contains = [&](int64_t x) { return x >= range_start && x < range_end; };
    auto lookup = [&](int64_t c, int64_t* out) {
        auto it = coordinate_lookup.find(c);
        if (it == coordinate_lookup.end()) return false;
        *out = it->second;
        return true;
    };
    const uint8_t* read_sequence = read->seq;
    const uint8_t* read_qualities = read->qual;

    size_t num_overlaps = 0;
    size_t last_deletion_end = 0;
    for (size_t vi = 0; vi < num_variants; ++vi) {
        const VariantView variant{&variant_calls[vi]};
        const int64_t variant_pos = variant.position();
        const uint32_t variant_type = variant.get_type();
        const size_t vt_index = variant_type;
        uint8_t allele, qual;
        bool exact_allele, overlaps_allele;

        if (variant.is_ignored()) {
            allele = NO_OVERLAP; qual = MISSING_QUAL; exact_allele = false; overlaps_allele = false;
        } else if (variant_pos < (int64_t)last_deletion_end) {
            allele = AMBIGUOUS; qual = MISSING_QUAL; exact_allele = false; overlaps_allele = true;
        } else if (variant_type == SNV || variant_type == INSERTION || variant_type == DELETION || variant_type == INDEL ||
                   variant_type == SV_INSERTION || variant_type == TANDEM_REPEAT) {
            const size_t ref_allele_len = variant.get_ref_len();
            const size_t prefix_len = variant.get_prefix_len();
            const size_t postfix_len = variant.get_postfix_len();
            if ((size_t)variant_pos < prefix_len) return -3;  // usize underflow
            const size_t first_start_coordinate = (size_t)variant_pos - prefix_len;
            const size_t last_start_coordinate = (size_t)variant_pos + 1;
            const size_t first_end_coordinate = (size_t)variant_pos + ref_allele_len;
            const size_t last_end_coordinate = (size_t)variant_pos + ref_allele_len + postfix_len + 1;

            bool has_closest_start = false, has_closest_end = false;
            size_t closest_start = 0, closest_end = 0;
            for (size_t sc = last_start_coordinate; sc-- > first_start_coordinate;) {
                int64_t si;
                if (lookup((int64_t)sc, &si)) { closest_start = (size_t)si; has_closest_start = true; break; }
            }
            for (size_t ec = first_end_coordinate; ec < last_end_coordinate; ++ec) {
                int64_t ei;
                if (lookup((int64_t)ec, &ei)) { closest_end = (size_t)ei; has_closest_end = true; break; }
            }

            bool has_start = false, has_end = false;
            size_t start_coordinate = 0, end_coordinate = 0, start_clip = 0, end_clip = 0;
            if (has_closest_start && has_closest_end) {
                for (size_t sc = first_start_coordinate; sc < last_start_coordinate; ++sc) {
                    start_clip += 1;
                    int64_t segment_index;
                    if (lookup((int64_t)sc, &segment_index)) {
                        if (closest_start - (size_t)segment_index > 2 * prefix_len) continue;
                        start_coordinate = (size_t)segment_index; has_start = true;
                        for (size_t ec = last_end_coordinate; ec-- > first_end_coordinate;) {
                            end_clip += 1;
                            int64_t next_index;
                            if (lookup((int64_t)ec, &next_index)) {
                                if ((size_t)next_index - closest_end > 2 * postfix_len) continue;
                                end_coordinate = (size_t)next_index; has_end = true;
                                break;
                            }
                        }
                        break;
                    }
                }
            }

            if (has_start) {
                if (has_end) {
                    const size_t ss = start_coordinate, se = end_coordinate;
                    if (se < ss || se > read->seq_len) return -3;  // slice panic
                    allele = variant.match_allele(read_sequence + ss, se - ss);
                    if (allele == AMBIGUOUS) {
                        uint64_t dmin, dother;
                        const int r = variant.closest_allele_clip(read_sequence + ss, se - ss, start_clip - 1, end_clip - 1, &dmin, &dother);
                        if (r < 0) return -3;
                        allele = (uint8_t)r;
                        exact_allele = false;
                    } else {
                        exact_allele = true;
                    }
                    const double max_qual_credit = 40.0;
                    double sum = 0.0;
                    for (size_t k = ss; k < se; ++k) sum += 1.0 / (double)read_qualities[k];
                    const double harmonic_qual = (double)(se - ss) / sum;
                    const double qual_factor = std::fmin(harmonic_qual / max_qual_credit, 1.0);  // f64::min ignores NaN
                    uint8_t baseline_quality;
                    switch (variant_type) {
                        case SNV: baseline_quality = SNV_QUAL; break;
                        case DELETION: case INSERTION: case INDEL: baseline_quality = INDEL_QUAL; break;
                        case SV_DELETION: case SV_INSERTION: baseline_quality = SV_INDEL_QUAL; break;
                        case TANDEM_REPEAT: baseline_quality = TR_QUAL; break;
                        default: return -3;
                    }
                    qual = f64_as_u8(std::fmax((double)baseline_quality * qual_factor, 1.0));
                    overlaps_allele = true;
                } else {
                    allele = AMBIGUOUS; qual = MISSING_QUAL; exact_allele = false; overlaps_allele = true;
                }
            } else {
                if (range_contains(variant_pos)) { overlaps_allele = true; allele = AMBIGUOUS; }
                else { overlaps_allele = false; allele = NO_OVERLAP; }
                qual = MISSING_QUAL;
                exact_allele = false;
            }
        } else if (variant_type == SV_DELETION) {
            const size_t ref_allele_len = variant.get_ref_len();
            if (range_contains(variant_pos)) {
                const size_t last_start_coordinate = (size_t)variant_pos + 1;
                const size_t first_end_coordinate = (size_t)variant_pos + ref_allele_len;
                if (range_contains((int64_t)first_end_coordinate)) {
                    if (first_end_coordinate < last_start_coordinate) return -3;  // usize underflow
                    const size_t expected_deleted = first_end_coordinate - last_start_coordinate;
                    size_t start_anchor = last_start_coordinate;
                    while (!coordinate_lookup.count((int64_t)start_anchor)) {
                        if (start_anchor <= (size_t)range_start) break;
                        start_anchor -= 1;
                    }
                    size_t end_anchor = first_end_coordinate;
                    while (!coordinate_lookup.count((int64_t)end_anchor)) {
                        end_anchor += 1;
                        if (end_anchor >= (size_t)range_end) break;
                    }
                    size_t deleted_count = 0;
                    for (size_t dc = start_anchor; dc < end_anchor; ++dc)
                        if (!coordinate_lookup.count((int64_t)dc)) deleted_count += 1;
                    const double match_window_size = 0.33;
                    const double deleted_ratio = (double)deleted_count / (double)expected_deleted;
                    if (deleted_ratio < match_window_size) {
                        allele = REFERENCE;
                        qual = f64_as_u8(std::fmax((double)SV_INDEL_QUAL * (1.0 - deleted_ratio), 1.0));
                        exact_allele = deleted_ratio == 0.0;
                    } else if (std::fabs(1.0 - deleted_ratio) < match_window_size) {
                        allele = ALTERNATE;
                        const double qual_frac = 1.0 - std::fabs(1.0 - deleted_ratio);
                        qual = f64_as_u8(std::fmax((double)SV_INDEL_QUAL * qual_frac, 1.0));
                        exact_allele = deleted_ratio == 1.0;
                        last_deletion_end = first_end_coordinate;
                    } else {
                        allele = AMBIGUOUS; qual = MISSING_QUAL; exact_allele = false;
                    }
                    overlaps_allele = true;
                } else {
                    allele = AMBIGUOUS; qual = MISSING_QUAL; exact_allele = false; overlaps_allele = true;
                }
            } else {
                allele = NO_OVERLAP; qual = MISSING_QUAL; exact_allele = false; overlaps_allele = false;
            }
        } else {
            return -3;  // panic!("Unhandled variant type")
        }

        // :460-483
        if (overlaps_allele) {
            if (allele > AMBIGUOUS) return -3;
            if (allele == AMBIGUOUS) {
                st.failed_matches[vt_index] += 1;
            } else {
                if (exact_allele) st.exact_matches[vt_index] += 1; else st.inexact_matches[vt_index] += 1;
                if (allele == REFERENCE) st.allele0_matches[vt_index] += 1; else st.allele1_matches[vt_index] += 1;
                num_overlaps += 1;
                num_alleles += 1;
            }
        } else if (allele != NO_OVERLAP) {
            return -3;
        }
        alleles[vi] = allele;
        quals[vi] = qual;
    }
    st.skipped_reads = num_overlaps == 0 ? 1 : 0;
    st.local_aligned = 1 - st.skipped_reads;
    st.num_alleles = num_alleles;
    if (out_stats) *out_stats = st;
    return 0;
}

}  // extern "C"
