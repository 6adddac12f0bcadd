// hiphase_gpu.hpp — C++17 host side ABOVE the C ABI (include/hiphase_gpu.h): the reference's own interfaces for
// the accelerated path, with the reference's names, argument meaning and error behaviour, so that C++ callers (and
// the parity tests in tests/cpp/) read like the Rust they stand in for. Header-only; link with -lhiphase_gpu.
//
// The reference is Rust and this image has no Rust toolchain, so the caller-side logic that INTEGRATION.md patches
// into phaser.rs / read_parsing.rs is written here in C++ (hiphase_amd/*.py is the same mirror for the Python
// tests). Every function cites the reference lines it follows. Nothing in this header computes on the CPU what the
// library computes on the GPU: scoring, A*, graph-WFA, Levenshtein and local re-alignment all go through the C ABI
// and fail with hiphase::Error when there is no device (the library has no CPU fallback).
//
//   read_segments.rs   AlleleType, ReadSegment (new / collapse / allele / qual / get_num_set / score_*_haplotype)
//   astar_phaser.rs    astar_solver -> AstarResult{haplotype_1, haplotype_2, PhaseStats}
//   variants.rs        VariantType, Variant (constructors, truncated / padded alleles, match_allele), closest_allele_clip
//   wfa_graph.rs +     global_realignment_batch: WFAGraph::from_reference_variants_with_hom +
//   read_parsing.rs      edit_distance_with_pruning + the node->allele mapping, for all records of a block at once;
//                      local_realignment_batch; sequence_alignment::edit_distance
//   phaser.rs          get_solution_span_counts, haplotag_reads; solve_block (from decoded records on) = ONE call into the
//                      library (hp_solve_blocks): load_full_read_segments / load_read_segments with the fallback replay,
//                      quality assignment, collapse, A*, span counts and haplotags run behind the C ABI
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hiphase_gpu.h"

namespace hiphase {

// The solver never returns an error in the reference: an inconsistency is a panic! that ends the process
// (astar_phaser.rs:631; main.rs:401-405). Here a negative C-ABI status becomes this exception.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc, const char* where) {
    if (rc < 0) throw Error(rc, std::string(where) + ": " + hp_last_error());
}
// the reference's assert!s on caller input
inline void require(bool ok, const char* what) {
    if (!ok) throw std::invalid_argument(what);
}

using Bytes = std::vector<uint8_t>;
inline Bytes bytes(const std::string& s) { return Bytes(s.begin(), s.end()); }

// ---- read_segments.rs ----------------------------------------------------------------------------------------
enum class AlleleType : uint8_t { Reference = 0, Alternate = 1, Ambiguous = 2, NoOverlap = 3 };  // read_segments.rs:5-16

class ReadSegment {  // read_segments.rs:19-62 — one clipped matrix row
public:
    // ReadSegment::new: clips to [first set allele, last set allele + 1); no set allele -> len..len
    ReadSegment(std::string read_name, const Bytes& alleles, const Bytes& quals) : read_name_(std::move(read_name)) {
        require(alleles.size() == quals.size(), "alleles and quals differ in length");
        const size_t n = alleles.size();
        size_t first = n, last = n;
        for (size_t i = 0; i < n; ++i) if (alleles[i] < (uint8_t)AlleleType::Ambiguous) { first = i; break; }
        for (size_t i = n; i-- > 0;) if (alleles[i] < (uint8_t)AlleleType::Ambiguous) { last = i + 1; break; }
        start_ = first;
        end_ = last;
        alleles_.assign(alleles.begin() + first, alleles.begin() + last);
        quals_.assign(quals.begin() + first, quals.begin() + last);
    }
    const std::string& read_name() const { return read_name_; }
    std::pair<size_t, size_t> region() const { return {start_, end_}; }
    size_t start() const { return start_; }
    size_t end() const { return end_; }
    const Bytes& alleles() const { return alleles_; }
    const Bytes& quals() const { return quals_; }
    uint8_t allele(size_t i) const {  // read_segments.rs:128-134
        return (i >= start_ && i < end_) ? alleles_[i - start_] : (uint8_t)AlleleType::NoOverlap;
    }
    uint8_t qual(size_t i) const {  // read_segments.rs:137-143
        return (i >= start_ && i < end_) ? quals_[i - start_] : 0;
    }
    // score_partial_haplotype (read_segments.rs:177-206): qualities of the cells where the haplotype (starting at
    // variant `offset`) is set and differs from the read; 0 when they do not overlap. On the solver path this is what
    // the kernels compute on the device; the host form is for callers and tests, as in the reference.
    uint64_t score_partial_haplotype(const Bytes& haplotype, size_t offset) const {
        if (haplotype.size() + offset <= start_ || offset >= end_) return 0;
        const size_t lo = std::max(start_, offset), hi = std::min(end_, offset + haplotype.size());
        uint64_t sum = 0;
        for (size_t i = lo; i < hi; ++i) {
            const uint8_t h = haplotype[i - offset];
            if (h < (uint8_t)AlleleType::Ambiguous && allele(i) != h) sum += qual(i);
        }
        return sum;
    }
    uint64_t score_haplotype(const Bytes& haplotype) const {  // read_segments.rs:161-168
        require(end_ <= haplotype.size(), "assert!(region.end <= haplotype.len()) (read_segments.rs:163)");
        return score_partial_haplotype(haplotype, 0);
    }
    size_t get_num_set() const {  // read_segments.rs:151-155
        size_t n = 0;
        for (uint8_t a : alleles_) n += a < (uint8_t)AlleleType::Ambiguous;
        return n;
    }
    // read_segments.rs:71-121: merge the mappings of one read; first non-NoOverlap wins, equal alleles keep the
    // larger quality (which must be > 0), different alleles -> Ambiguous with quality 0; then re-clip
    static ReadSegment collapse(const std::vector<ReadSegment>& rs) {
        require(!rs.empty(), "collapse of nothing");
        if (rs.size() == 1) return rs[0];
        size_t min_start = rs[0].start_, max_end = rs[0].end_;
        for (const auto& r : rs) { min_start = std::min(min_start, r.start_); max_end = std::max(max_end, r.end_); }
        Bytes alleles(max_end, (uint8_t)AlleleType::NoOverlap), quals(max_end, 0);
        for (const auto& r : rs) {
            require(r.read_name_ == rs[0].read_name_, "collapse of different reads");
            for (size_t i = min_start; i < max_end; ++i) {
                const uint8_t a = r.allele(i), q = r.qual(i);
                if (a == (uint8_t)AlleleType::NoOverlap) continue;
                if (alleles[i] == (uint8_t)AlleleType::NoOverlap) { alleles[i] = a; quals[i] = q; }
                else if (alleles[i] == (uint8_t)AlleleType::Ambiguous) {}
                else if (alleles[i] == a) {
                    quals[i] = std::max(quals[i], q);
                    require(quals[i] > 0, "assert!(quals[i] > 0) (read_segments.rs:105)");
                } else { alleles[i] = (uint8_t)AlleleType::Ambiguous; quals[i] = 0; }
            }
        }
        return ReadSegment(rs[0].read_name_, alleles, quals);
    }
    bool operator==(const ReadSegment& o) const {
        return read_name_ == o.read_name_ && start_ == o.start_ && end_ == o.end_ && alleles_ == o.alleles_ && quals_ == o.quals_;
    }

private:
    std::string read_name_;
    size_t start_ = 0, end_ = 0;
    Bytes alleles_, quals_;
};

// The solver's IntervalTree<usize, ReadSegment> (phaser.rs:514-533) as the CSR view the C ABI takes.
struct BlockMatrix {
    std::vector<uint32_t> read_start, read_end;
    std::vector<uint64_t> row_off;
    Bytes alleles_2bit, quals, var_flags;
    BlockMatrix(const std::vector<ReadSegment>& segs, const Bytes& flags) : var_flags(flags) {
        row_off.push_back(0);
        Bytes cells;
        for (const auto& s : segs) {
            read_start.push_back((uint32_t)s.start());
            read_end.push_back((uint32_t)s.end());
            row_off.push_back(row_off.back() + (s.end() - s.start()));
            cells.insert(cells.end(), s.alleles().begin(), s.alleles().end());
            quals.insert(quals.end(), s.quals().begin(), s.quals().end());
        }
        alleles_2bit.assign((cells.size() + 3) / 4 + 1, 0);
        for (size_t i = 0; i < cells.size(); ++i) alleles_2bit[i >> 2] |= (uint8_t)(cells[i] << (2 * (i & 3)));
        if (quals.empty()) quals.push_back(0);
    }
    hp_block_view view() const {
        hp_block_view v{};
        v.n_variants = (uint32_t)var_flags.size();
        v.n_reads = (uint32_t)read_start.size();
        v.read_start = read_start.data();
        v.read_end = read_end.data();
        v.row_off = row_off.data();
        v.alleles_2bit = alleles_2bit.data();
        v.quals = quals.data();
        v.var_flags = var_flags.data();
        return v;
    }
};

// ---- variants.rs ---------------------------------------------------------------------------------------------
enum class VariantType : uint32_t {  // variants.rs:10-33
    Snv = 0, Insertion, Deletion, Indel, SvInsertion, SvDeletion, SvDuplication, SvInversion, SvBreakend, TandemRepeat, Unknown
};
// read_parsing.rs:18-22: base qualities per type; global re-alignment doubles them (read_parsing.rs:815)
inline uint8_t base_quality(VariantType t) {
    switch (t) {
        case VariantType::Snv: return 80;
        case VariantType::Insertion: case VariantType::Deletion: case VariantType::Indel: return 10;
        case VariantType::SvInsertion: case VariantType::SvDeletion: return 20;
        case VariantType::TandemRepeat: return 40;
        default: throw std::invalid_argument("variant type without a base quality (read_parsing.rs:18-22)");
    }
}

struct Variant {  // what the path reads from a `Variant` (variants.rs:67-94)
    uint32_t vcf_index = 0;
    VariantType variant_type = VariantType::Unknown;
    int64_t position = 0;
    uint32_t ref_len = 0;
    Bytes allele0, allele1;            // truncated alleles (variants.rs:581-591): what the WFA graph uses
    uint32_t index_allele0 = 0, index_allele1 = 1;
    bool is_ignored = false;
    Bytes prefix, postfix;             // +-reference_buffer padding (variants.rs:497-539), local re-alignment only

    Bytes get_allele0() const { Bytes a = prefix; a.insert(a.end(), allele0.begin(), allele0.end()); a.insert(a.end(), postfix.begin(), postfix.end()); return a; }
    Bytes get_allele1() const { Bytes a = prefix; a.insert(a.end(), allele1.begin(), allele1.end()); a.insert(a.end(), postfix.begin(), postfix.end()); return a; }
    uint8_t match_allele(const Bytes& a) const {  // variants.rs:598-606
        return a == get_allele0() ? 0 : (a == get_allele1() ? 1 : 2);
    }
    // constructors with the reference's validation essentials (variants.rs:109-492)
    static Variant make(uint32_t vi, VariantType t, int64_t pos, uint32_t rl, Bytes a0, Bytes a1, uint32_t i0, uint32_t i1) {
        Variant v;
        v.vcf_index = vi; v.variant_type = t; v.position = pos; v.ref_len = rl;
        v.allele0 = std::move(a0); v.allele1 = std::move(a1); v.index_allele0 = i0; v.index_allele1 = i1;
        return v;
    }
    static Variant new_snv(uint32_t vi, int64_t pos, Bytes a0, Bytes a1, uint32_t i0, uint32_t i1) {
        require(i0 < i1 && a0.size() == 1 && a1.size() == 1, "new_snv");
        return make(vi, VariantType::Snv, pos, 1, std::move(a0), std::move(a1), i0, i1);
    }
    static Variant new_deletion(uint32_t vi, int64_t pos, uint32_t rl, Bytes a0, Bytes a1, uint32_t i0, uint32_t i1) {
        require(i0 < i1 && rl > 1 && a1.size() == 1 && a0.size() == (i0 == 0 ? rl : 1u), "new_deletion");
        return make(vi, VariantType::Deletion, pos, rl, std::move(a0), std::move(a1), i0, i1);
    }
    static Variant new_insertion(uint32_t vi, int64_t pos, Bytes a0, Bytes a1, uint32_t i0, uint32_t i1) {
        require(i0 < i1 && !a1.empty() && (i0 == 0 ? a0.size() == 1 : !a0.empty()), "new_insertion");
        return make(vi, VariantType::Insertion, pos, 1, std::move(a0), std::move(a1), i0, i1);
    }
    static Variant new_indel(uint32_t vi, int64_t pos, uint32_t rl, Bytes a0, Bytes a1, uint32_t i0, uint32_t i1) {
        require(i0 < i1 && rl > 1 && !a1.empty() && (i0 == 0 ? a0.size() == rl : !a0.empty()), "new_indel");
        return make(vi, VariantType::Indel, pos, rl, std::move(a0), std::move(a1), i0, i1);
    }
    static Variant new_sv_deletion(uint32_t vi, int64_t pos, uint32_t rl, Bytes a0, Bytes a1) {
        require(a0.size() == rl && !a1.empty() && a1.size() <= a0.size(), "new_sv_deletion");
        return make(vi, VariantType::SvDeletion, pos, rl, std::move(a0), std::move(a1), 0, 1);
    }
    static Variant new_sv_insertion(uint32_t vi, int64_t pos, uint32_t rl, Bytes a0, Bytes a1) {
        require(a0.size() == rl && !a0.empty() && a1.size() >= a0.size(), "new_sv_insertion");
        return make(vi, VariantType::SvInsertion, pos, rl, std::move(a0), std::move(a1), 0, 1);
    }
    static Variant new_tandem_repeat(uint32_t vi, int64_t pos, uint32_t rl, Bytes a0, Bytes a1, uint32_t i0, uint32_t i1) {
        require(i0 < i1 && !a0.empty() && !a1.empty() && (i0 != 0 || a0.size() == rl), "new_tandem_repeat");
        return make(vi, VariantType::TandemRepeat, pos, rl, std::move(a0), std::move(a1), i0, i1);
    }
};

// ---- astar_phaser.rs -----------------------------------------------------------------------------------------
struct PhaseStats {  // writers/phase_stats.rs:131-173 (the solver's fields)
    uint64_t pruned_solutions = 0, estimated_cost = 0, actual_cost = 0, phased_variants = 0, phased_snvs = 0,
             homozygous_variants = 0, skipped_variants = 0;
    bool operator==(const PhaseStats& o) const {
        return pruned_solutions == o.pruned_solutions && estimated_cost == o.estimated_cost && actual_cost == o.actual_cost &&
               phased_variants == o.phased_variants && phased_snvs == o.phased_snvs &&
               homozygous_variants == o.homozygous_variants && skipped_variants == o.skipped_variants;
    }
};
struct AstarResult {  // astar_phaser.rs:408-415
    Bytes haplotype_1, haplotype_2;
    PhaseStats statistics;
};
inline Bytes variant_flags(const std::vector<Variant>& variants) {  // what astar_solver reads from &[Variant]
    Bytes f(variants.size());
    for (size_t i = 0; i < variants.size(); ++i)
        f[i] = (uint8_t)((variants[i].is_ignored ? HP_VAR_IGNORED : 0) | (variants[i].variant_type == VariantType::Snv ? HP_VAR_SNV : 0));
    return f;
}
// astar_solver(phase_block, variants, read_segments, min_queue_size, queue_increment) (astar_phaser.rs:426-429):
// from the PhaseBlock only the index is read (diagnostics), from each Variant is_ignored() and get_type() == Snv
inline AstarResult astar_solver(uint64_t block_index, const Bytes& var_flags, const std::vector<ReadSegment>& read_segments,
                                uint64_t min_queue_size = 1000, uint64_t queue_increment = 3) {
    const BlockMatrix m(read_segments, var_flags);
    const hp_block_view v = m.view();
    hp_astar_params p{min_queue_size, queue_increment, 0, block_index};
    AstarResult r;
    r.haplotype_1.assign(var_flags.size(), 0);
    r.haplotype_2.assign(var_flags.size(), 0);
    hp_phase_stats st{};
    check(hp_astar_solve(&v, &p, r.haplotype_1.data(), r.haplotype_2.data(), &st), "hp_astar_solve");
    r.statistics = PhaseStats{st.pruned_solutions, st.estimated_cost, st.actual_cost, st.phased_variants, st.phased_snvs,
                              st.homozygous_variants, st.skipped_variants};
    return r;
}
inline AstarResult astar_solver(uint64_t block_index, const std::vector<Variant>& variants, const std::vector<ReadSegment>& read_segments,
                                uint64_t min_queue_size = 1000, uint64_t queue_increment = 3) {
    return astar_solver(block_index, variant_flags(variants), read_segments, min_queue_size, queue_increment);
}

// ---- sequence_alignment.rs -----------------------------------------------------------------------------------
inline uint64_t edit_distance(const Bytes& v1, const Bytes& v2) {  // sequence_alignment.rs:7-38
    static const uint8_t none = 0;
    hp_ed_pair pr{v1.empty() ? &none : v1.data(), v2.empty() ? &none : v2.data(), (uint32_t)v1.size(), (uint32_t)v2.size()};
    uint64_t out = 0;
    check(hp_edit_distance_batch(&pr, 1, &out, -1), "hp_edit_distance_batch");
    return out;
}

// Variant::closest_allele_clip (variants.rs:624-641): the allele (0 / 1 / 2 = Ambiguous on a tie) nearer to `allele`
// in edit distance once `head_clip` / `tail_clip` bases are removed from both padded alleles; both distances ride along
struct ClosestAllele { uint8_t allele; uint64_t min_ed, other_ed; };
inline ClosestAllele closest_allele_clip(const Variant& v, const Bytes& allele, size_t head_clip = 0, size_t tail_clip = 0) {
    const Bytes f0 = v.get_allele0(), f1 = v.get_allele1();
    require(head_clip + tail_clip <= f0.size() && head_clip + tail_clip <= f1.size(), "clip longer than an allele");
    const Bytes a0(f0.begin() + head_clip, f0.end() - tail_clip), a1(f1.begin() + head_clip, f1.end() - tail_clip);
    static const uint8_t none = 0;
    auto ptr = [](const Bytes& b) { return b.empty() ? &none : b.data(); };
    hp_ed_pair pr[2] = {{ptr(allele), ptr(a0), (uint32_t)allele.size(), (uint32_t)a0.size()},
                        {ptr(allele), ptr(a1), (uint32_t)allele.size(), (uint32_t)a1.size()}};
    uint64_t d[2] = {0, 0};
    check(hp_edit_distance_batch(pr, 2, d, -1), "hp_edit_distance_batch");
    if (d[0] < d[1]) return {0, d[0], d[1]};
    if (d[0] > d[1]) return {1, d[1], d[0]};
    return {2, d[0], d[1]};
}

// ---- read_parsing.rs -----------------------------------------------------------------------------------------
struct GlobalRealignmentConfig {  // read_parsing.rs:25-34 (defaults: cli.rs:189-210)
    uint64_t max_edit_distance = 500, wfa_prune_distance = 500;
    double global_failure_ratio = 0.5;
    uint64_t global_failure_minimum = 50;
};
struct LocalRecord {  // what local_realignment reads from a bam::Record (read_parsing.rs:121-160)
    std::string qname;
    int64_t pos = 0;
    std::vector<uint32_t> cigar;   // BAM encoding: len << 4 | op (MIDNSHP=X = 0..8)
    Bytes seq, qual;
};
struct AlignedRecord {  // what global_realignment needs from one record (read_parsing.rs:672-742)
    std::string qname;
    int64_t min_position = 0, max_position = 0;   // first / last reference base of the alignment (inclusive)
    Bytes read_align;                             // seq[read_start..=read_end]
    bool has_local = false;
    LocalRecord local;                            // CIGAR view of the same record (needed when it falls back)
};
struct LoadStats {   // ReadStats (writers/phase_stats.rs:12-33) + the edit distances the loader logs
    uint64_t num_reads = 0, skipped_reads = 0, global_aligned = 0, local_aligned = 0;
    uint64_t num_alleles = 0;
    std::array<uint64_t, HP_N_VARIANT_TYPES> exact_matches{}, inexact_matches{}, failed_matches{}, allele0_matches{}, allele1_matches{};
    std::vector<uint64_t> edit_distances;
};
struct WfaOutcome {  // Ok(WFAResult) mapped to per-het alleles (read_parsing.rs:790-800) | Err(MaxEditDistance)
    bool max_edit_distance = false;
    uint64_t score = 0;
    uint32_t num_nodes = 0;
    Bytes alleles;   // one AlleleType per het variant handed to the job
};
struct WfaJob {  // one record's WFAGraph::from_reference_variants_with_hom + edit_distance_with_pruning
    const Bytes* reference = nullptr;   // chromosome (or slice) with (*reference)[0] at coordinate ref_base
    uint64_t ref_base = 0, ref_start = 0, ref_end = 0;
    const Variant* hets = nullptr; size_t n_hets = 0;
    const Variant* homs = nullptr; size_t n_homs = 0;
    const Bytes* read = nullptr;
};
namespace detail {
inline void pack_variants(const Variant* v, size_t n, std::vector<hp_wfa_variant>& out) {
    static const uint8_t none = 0;
    for (size_t i = 0; i < n; ++i) {
        hp_wfa_variant w{};
        w.position = v[i].position;
        w.ref_len = v[i].ref_len;
        w.flags = (v[i].is_ignored ? 1u : 0u) | (v[i].index_allele0 != 0 ? 2u : 0u);
        w.allele0 = v[i].allele0.empty() ? &none : v[i].allele0.data();
        w.allele1 = v[i].allele1.empty() ? &none : v[i].allele1.data();
        w.allele0_len = (uint32_t)v[i].allele0.size();
        w.allele1_len = (uint32_t)v[i].allele1.size();
        out.push_back(w);
    }
}
// indices [first, last) of the variants with lo <= position <= hi (read_parsing.rs:688-700, 721-730)
inline bool overlap_range(const std::vector<Variant>& v, int64_t lo, int64_t hi, size_t& first, size_t& last) {
    bool any = false;
    for (size_t i = 0; i < v.size(); ++i)
        if (lo <= v[i].position && v[i].position <= hi) { if (!any) { first = i; any = true; } last = i + 1; }
    return any;
}
}  // namespace detail

// WFAGraph (wfa_graph.rs:93-117), add_node (:298-331), edit_distance_with_pruning (:350-650) -> WFAResult (:654-670) for a
// caller-built graph: the layer under from_reference_variants_with_hom (hp_wfa_align_graphs)
struct WFAResult {
    uint64_t score = 0;
    std::vector<size_t> traversed_nodes;   // ascending
};
class WFAGraph {
    std::vector<Bytes> seqs_;
    std::vector<std::vector<uint32_t>> parents_;
public:
    // returns the node's index; the reference's asserts (:305-312) are reported by the alignment call
    size_t add_node(const Bytes& sequence, std::vector<size_t> parents) {
        std::sort(parents.begin(), parents.end());
        seqs_.push_back(sequence);
        parents_.emplace_back(parents.begin(), parents.end());
        return seqs_.size() - 1;
    }
    size_t get_num_nodes() const { return seqs_.size(); }
    // Err(MaxEditDistance) <=> `max_edit_distance_reached` is set (the result is then empty)
    WFAResult edit_distance_with_pruning(const Bytes& other, uint64_t prune_distance, uint64_t max_edit_distance, bool* max_edit_distance_reached = nullptr) const {
        std::vector<hp_graph_node> nodes(seqs_.size());
        static const uint8_t none8 = 0;
        static const uint32_t none32 = 0;
        for (size_t k = 0; k < seqs_.size(); ++k) {
            nodes[k].seq = seqs_[k].empty() ? &none8 : seqs_[k].data(); nodes[k].seq_len = (uint32_t)seqs_[k].size();
            nodes[k].parents = parents_[k].empty() ? &none32 : parents_[k].data(); nodes[k].n_parents = (uint32_t)parents_[k].size();
        }
        hp_graph_job job{};
        job.nodes = nodes.data(); job.n_nodes = (uint32_t)nodes.size();
        job.read = other.empty() ? &none8 : other.data(); job.read_len = (uint32_t)other.size();
        std::vector<uint32_t> set((nodes.size() + 31) / 32 + 1, 0);
        uint32_t* ptr = set.data();
        hp_graph_result res{};
        check(hp_wfa_align_graphs(&job, 1, prune_distance == 0 ? UINT64_MAX : prune_distance, max_edit_distance, &res, &ptr, -1), "hp_wfa_align_graphs");
        WFAResult r;
        if (max_edit_distance_reached) *max_edit_distance_reached = res.status == HP_WFA_MAX_ED;
        if (res.status != HP_OK) return r;
        r.score = res.score;
        for (size_t k = 0; k < nodes.size(); ++k) if ((set[k >> 5] >> (k & 31)) & 1u) r.traversed_nodes.push_back(k);
        return r;
    }
};

// read_parsing.rs:769-800 for a batch of records: one hp_wfa_assign_batch call
inline std::vector<WfaOutcome> global_realignment_batch(const std::vector<WfaJob>& jobs, uint64_t prune_distance, uint64_t max_edit_distance) {
    const size_t n = jobs.size();
    std::vector<WfaOutcome> out(n);
    if (n == 0) return out;
    std::vector<std::vector<hp_wfa_variant>> hv(n), mv(n);
    std::vector<hp_wfa_job> cj(n);
    std::vector<uint8_t*> ptrs(n);
    static const uint8_t none = 0;
    for (size_t i = 0; i < n; ++i) {
        detail::pack_variants(jobs[i].hets, jobs[i].n_hets, hv[i]);
        detail::pack_variants(jobs[i].homs, jobs[i].n_homs, mv[i]);
        hp_wfa_job j{};
        j.reference = jobs[i].reference->data();
        j.ref_base = jobs[i].ref_base; j.ref_start = jobs[i].ref_start; j.ref_end = jobs[i].ref_end;
        j.hets = hv[i].data(); j.n_hets = (uint32_t)hv[i].size();
        j.homs = mv[i].data(); j.n_homs = (uint32_t)mv[i].size();
        j.read = jobs[i].read->empty() ? &none : jobs[i].read->data();
        j.read_len = (uint32_t)jobs[i].read->size();
        cj[i] = j;
        out[i].alleles.assign(std::max<size_t>(jobs[i].n_hets, 1), (uint8_t)AlleleType::NoOverlap);
        ptrs[i] = out[i].alleles.data();
    }
    std::vector<hp_wfa_result> res(n);
    check(hp_wfa_assign_batch(cj.data(), n, prune_distance == 0 ? UINT64_MAX : prune_distance, max_edit_distance, res.data(), ptrs.data(), -1),
          "hp_wfa_assign_batch");
    for (size_t i = 0; i < n; ++i) {
        out[i].alleles.resize(jobs[i].n_hets);
        out[i].max_edit_distance = res[i].status == HP_WFA_MAX_ED;
        out[i].score = res[i].score;
        out[i].num_nodes = res[i].n_nodes;
    }
    return out;
}

struct LocalResult { Bytes alleles, quals; hp_read_stats stats; };
// `local_realignment(read, variant_calls)` (read_parsing.rs:121-503) for a list of records: hp_local_realign_batch
inline std::vector<LocalResult> local_realignment_batch(const std::vector<const LocalRecord*>& records, const std::vector<Variant>& variant_calls) {
    const size_t nr = records.size(), nv = variant_calls.size();
    std::vector<LocalResult> out(nr);
    if (nr == 0) return out;
    static const uint8_t none = 0;
    static const uint32_t none32 = 0;
    std::vector<Bytes> a0(nv), a1(nv);
    std::vector<hp_local_variant> vs(std::max<size_t>(nv, 1));
    for (size_t i = 0; i < nv; ++i) {
        const Variant& v = variant_calls[i];
        a0[i] = v.get_allele0();
        a1[i] = v.get_allele1();
        hp_local_variant w{};
        w.position = v.position; w.ref_len = v.ref_len; w.variant_type = (uint32_t)v.variant_type;
        w.prefix_len = (uint32_t)v.prefix.size(); w.postfix_len = (uint32_t)v.postfix.size();
        w.allele0 = a0[i].empty() ? &none : a0[i].data(); w.allele1 = a1[i].empty() ? &none : a1[i].data();
        w.allele0_len = (uint32_t)a0[i].size(); w.allele1_len = (uint32_t)a1[i].size();
        w.flags = v.is_ignored ? HP_VAR_IGNORED : 0;
        vs[i] = w;
    }
    std::vector<hp_local_read> rs(nr);
    for (size_t i = 0; i < nr; ++i) {
        const LocalRecord& r = *records[i];
        require(r.seq.size() == r.qual.size(), "assert_eq!(sequence length, quality length) (read_parsing.rs:155)");
        hp_local_read lr{};
        lr.pos = r.pos;
        lr.cigar = r.cigar.empty() ? &none32 : r.cigar.data();
        lr.n_cigar = (uint32_t)r.cigar.size();
        lr.seq_len = (uint32_t)r.seq.size();
        lr.seq = r.seq.empty() ? &none : r.seq.data();
        lr.qual = r.qual.empty() ? &none : r.qual.data();
        rs[i] = lr;
    }
    const size_t stride = std::max<size_t>(nv, 1);
    Bytes alleles(nr * stride, 0), quals(nr * stride, 0);
    std::vector<hp_read_stats> stats(nr);
    check(hp_local_realign_batch(rs.data(), nr, vs.data(), nv, alleles.data(), quals.data(), stats.data(), -1), "hp_local_realign_batch");
    for (size_t i = 0; i < nr; ++i) {
        out[i].alleles.assign(alleles.begin() + i * stride, alleles.begin() + i * stride + nv);
        out[i].quals.assign(quals.begin() + i * stride, quals.begin() + i * stride + nv);
        out[i].stats = stats[i];
    }
    return out;
}

// ---- phaser.rs -----------------------------------------------------------------------------------------------
// get_solution_span_counts (phaser.rs:350-388): per juncture, the reads spanning it after trimming the ends of the
// read that the solution leaves homozygous
inline std::vector<uint64_t> get_solution_span_counts(const std::vector<ReadSegment>& read_segments, const Bytes& h1, const Bytes& h2) {
    std::vector<uint64_t> counts(h1.empty() ? 0 : h1.size() - 1, 0);
    for (const auto& rs : read_segments) {
        if (rs.end() == rs.start()) continue;
        size_t js = rs.start(), je = rs.end() - 1;
        while (js < je && h1[js] == h2[js]) ++js;
        while (js < je && h1[je] == h2[je]) --je;
        for (size_t j = js; j < je; ++j) counts[j] += 1;
    }
    return counts;
}
struct Haplotag { std::string read_name; int64_t phase_block = 0; uint8_t haplotag = 0; };
// haplotag_reads (phaser.rs:714-750): argmin of the two haplotype scores, ties untagged; PS = the tag of the first
// het the read resolves. Rows in read_segments order (the reference's hash map is keyed by read name).
inline std::vector<Haplotag> haplotag_reads(const std::vector<ReadSegment>& read_segments, const Bytes& h1, const Bytes& h2,
                                            const std::vector<int64_t>& block_tags) {
    std::vector<Haplotag> out;
    for (const auto& rs : read_segments) {
        uint64_t s1 = 0, s2 = 0;
        for (size_t i = rs.start(); i < rs.end(); ++i) {   // score_haplotype (read_segments.rs:161-206)
            const uint8_t a = rs.allele(i), q = rs.qual(i);
            if (h1[i] < 2 && a != h1[i]) s1 += q;
            if (h2[i] < 2 && a != h2[i]) s2 += q;
        }
        if (s1 == s2) continue;
        size_t first = rs.start();
        while (h1[first] == h2[first] || rs.allele(first) >= (uint8_t)AlleleType::Ambiguous) ++first;
        out.push_back(Haplotag{rs.read_name(), block_tags[first], (uint8_t)(s1 < s2 ? 0 : 1)});
    }
    return out;
}

struct PhaseResult {  // phaser.rs:326-343 (the solver-facing fields)
    Bytes haplotype_1, haplotype_2;
    std::vector<int64_t> block_ids;                    // PS tag per variant
    std::vector<std::vector<size_t>> sub_phase_blocks; // phased hets per sub-block
    PhaseStats statistics;
    std::vector<Haplotag> haplotags;
    LoadStats load_stats;
    std::vector<ReadSegment> read_segments;
};
// solve_block (phaser.rs:406-649) from the decoded records on. ONE call into the library (hp_solve_blocks): graph-WFA for
// every record with overlaps, the local fallback and the order-dependent `global_disabled` replay (read_parsing.rs:
// 556-600), quality assignment, collapse per read name, the min_matched_alleles split, A*, span counts and haplotags
// all run behind the C ABI; what is left here is marshalling, the PS tag walk (phaser.rs:557-565) and the sub-block
// split (:569-611), which need the variant positions this side owns.
inline PhaseResult solve_block(uint64_t block_index, const std::vector<AlignedRecord>& records, const std::vector<Variant>& variant_calls,
                               const std::vector<Variant>& hom_calls, const Bytes& reference, uint64_t ref_base = 0,
                               size_t min_matched_alleles = 2, uint64_t min_queue_size = 1000, uint64_t queue_increment = 3,
                               const GlobalRealignmentConfig* global_config = nullptr, bool global_realignment = true) {
    require(!variant_calls.empty(), "solve_block needs at least one variant");
    const GlobalRealignmentConfig cfg = global_config ? *global_config : GlobalRealignmentConfig();
    const size_t N = variant_calls.size(), R = records.size();
    static const uint8_t none = 0;
    static const uint32_t none32 = 0;
    std::vector<hp_wfa_variant> hv, mv;
    detail::pack_variants(variant_calls.data(), N, hv);
    detail::pack_variants(hom_calls.data(), hom_calls.size(), mv);
    Bytes types(N);
    std::vector<Bytes> a0(N), a1(N);
    std::vector<hp_local_variant> lv(N);
    for (size_t i = 0; i < N; ++i) {
        const Variant& v = variant_calls[i];
        types[i] = (uint8_t)v.variant_type;
        a0[i] = v.get_allele0(); a1[i] = v.get_allele1();
        hp_local_variant w{};
        w.position = v.position; w.ref_len = v.ref_len; w.variant_type = (uint32_t)v.variant_type;
        w.prefix_len = (uint32_t)v.prefix.size(); w.postfix_len = (uint32_t)v.postfix.size();
        w.allele0 = a0[i].empty() ? &none : a0[i].data(); w.allele1 = a1[i].empty() ? &none : a1[i].data();
        w.allele0_len = (uint32_t)a0[i].size(); w.allele1_len = (uint32_t)a1[i].size();
        w.flags = v.is_ignored ? HP_VAR_IGNORED : 0;
        lv[i] = w;
    }
    std::map<std::string, uint32_t> ids;
    std::vector<std::string> names;
    std::vector<hp_block_record> recs(std::max<size_t>(R, 1));
    std::vector<hp_local_read> locs(std::max<size_t>(R, 1));
    for (size_t i = 0; i < R; ++i) {
        const AlignedRecord& r = records[i];
        auto it = ids.find(r.qname);
        if (it == ids.end()) { it = ids.emplace(r.qname, (uint32_t)names.size()).first; names.push_back(r.qname); }
        hp_block_record& o = recs[i];
        o = hp_block_record{};
        o.qname_id = it->second;
        if (global_realignment) {
            o.min_position = r.min_position; o.max_position = r.max_position;
            o.read_align = r.read_align.empty() ? &none : r.read_align.data(); o.read_len = (uint32_t)r.read_align.size();
        } else {
            require(r.has_local, "local mode needs the CIGAR view of every record");
            o.min_position = o.max_position = r.local.pos;
            o.read_align = &none;
        }
        if (r.has_local) {
            const LocalRecord& l = r.local;
            require(l.seq.size() == l.qual.size(), "assert_eq!(sequence length, quality length) (read_parsing.rs:155)");
            hp_local_read& lr = locs[i];
            lr = hp_local_read{};
            lr.pos = l.pos; lr.cigar = l.cigar.empty() ? &none32 : l.cigar.data(); lr.n_cigar = (uint32_t)l.cigar.size();
            lr.seq_len = (uint32_t)l.seq.size(); lr.seq = l.seq.empty() ? &none : l.seq.data(); lr.qual = l.qual.empty() ? &none : l.qual.data();
            o.local = &lr;
        }
    }
    hp_block_input in{};
    in.block_index = block_index; in.reference = reference.empty() ? &none : reference.data(); in.ref_base = ref_base;
    in.n_hets = (uint32_t)N; in.n_homs = (uint32_t)hom_calls.size(); in.n_records = (uint32_t)R; in.n_qnames = (uint32_t)names.size();
    in.hets = hv.data(); in.het_types = types.data(); in.local_hets = lv.data(); in.homs = mv.empty() ? nullptr : mv.data(); in.records = recs.data();
    hp_block_params prm{};
    prm.astar = hp_astar_params{min_queue_size, queue_increment, 0, block_index};
    prm.wfa_prune_distance = cfg.wfa_prune_distance == 0 ? UINT64_MAX : cfg.wfa_prune_distance;
    prm.max_edit_distance = cfg.max_edit_distance;
    prm.global_failure_ratio = cfg.global_failure_ratio; prm.global_failure_minimum = cfg.global_failure_minimum;
    prm.min_matched_alleles = min_matched_alleles; prm.global_realignment = global_realignment ? 1u : 0u;
    const size_t Q = std::max<size_t>(names.size(), 1);
    PhaseResult pr;
    pr.haplotype_1.assign(N, 0); pr.haplotype_2.assign(N, 0);
    std::vector<uint64_t> spans(std::max<size_t>(N, 2) - 1), row_off(Q + 1), eds(std::max<size_t>(R, 1));
    std::vector<uint32_t> qn(Q), st(Q), en(Q), fh(Q);
    Bytes so(Q), ht(Q), al(Q * N + 1), ql(Q * N + 1);
    hp_block_output out{};
    out.h1 = pr.haplotype_1.data(); out.h2 = pr.haplotype_2.data(); out.span_counts = spans.data();
    out.seg_qname = qn.data(); out.seg_start = st.data(); out.seg_end = en.data(); out.seg_solver = so.data();
    out.seg_haplotag = ht.data(); out.seg_first_het = fh.data(); out.seg_row_off = row_off.data();
    out.seg_alleles = al.data(); out.seg_quals = ql.data(); out.seg_cell_cap = Q * N + 1; out.edit_distances = eds.data();
    check(hp_solve_blocks(1, &in, &prm, &out, -1), "hp_solve_blocks");
    require(out.status == HP_OK, "block outside the device solver's limits (HP_BLOCK_UNSUPPORTED): solve it with the reference's own solve_block");
    pr.statistics = PhaseStats{out.stats.pruned_solutions, out.stats.estimated_cost, out.stats.actual_cost, out.stats.phased_variants,
                               out.stats.phased_snvs, out.stats.homozygous_variants, out.stats.skipped_variants};
    int64_t cur = variant_calls[0].position;
    for (size_t i = 0; i < N; ++i) {   // phaser.rs:557-565
        if (i > 0 && spans[i - 1] == 0) cur = variant_calls[i].position;
        pr.block_ids.push_back(cur);
    }
    std::vector<size_t> block;                            // phaser.rs:569-611
    int64_t cur_tag = pr.block_ids[0];
    for (size_t i = 0; i < N; ++i) {
        const uint8_t a = pr.haplotype_1[i], b = pr.haplotype_2[i];
        if (a < 2 && b < 2 && a != b) {
            if (cur_tag != pr.block_ids[i]) {
                if (!block.empty()) { pr.sub_phase_blocks.push_back(block); block.clear(); }
                cur_tag = pr.block_ids[i];
            }
            block.push_back(i);
        }
    }
    if (!block.empty()) pr.sub_phase_blocks.push_back(block);
    // solver segments first, then the phasable-only ones (the order haplotag_reads is called in, phaser.rs:614-630)
    for (int pass = 0; pass < 2; ++pass)
        for (uint32_t k = 0; k < out.n_segments; ++k) {
            if ((so[k] != 0) != (pass == 0)) continue;
            const Bytes a(al.begin() + row_off[k], al.begin() + row_off[k + 1]), q(ql.begin() + row_off[k], ql.begin() + row_off[k + 1]);
            if (pass == 0) {
                Bytes fa(N, (uint8_t)AlleleType::NoOverlap), fq(N, 0);
                std::copy(a.begin(), a.end(), fa.begin() + st[k]);
                std::copy(q.begin(), q.end(), fq.begin() + st[k]);
                pr.read_segments.push_back(ReadSegment(names[qn[k]], fa, fq));
            }
            if (ht[k] != 2) pr.haplotags.push_back(Haplotag{names[qn[k]], pr.block_ids[fh[k]], ht[k]});
        }
    pr.load_stats.num_reads = out.num_reads; pr.load_stats.skipped_reads = out.skipped_reads;
    pr.load_stats.global_aligned = out.global_aligned; pr.load_stats.local_aligned = out.local_aligned;
    pr.load_stats.num_alleles = out.num_alleles;
    for (int t = 0; t < HP_N_VARIANT_TYPES; ++t) {
        pr.load_stats.exact_matches[t] = out.exact_matches[t]; pr.load_stats.inexact_matches[t] = out.inexact_matches[t]; pr.load_stats.failed_matches[t] = out.failed_matches[t];
        pr.load_stats.allele0_matches[t] = out.allele0_matches[t]; pr.load_stats.allele1_matches[t] = out.allele1_matches[t];
    }
    pr.load_stats.edit_distances.assign(eds.begin(), eds.begin() + out.n_edit_distances);
    return pr;
}

}  // namespace hiphase
