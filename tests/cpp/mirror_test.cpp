// mirror_test.cpp — parity tests of the C++ host mirror (include/hiphase_gpu.hpp) written the way the reference's own
// Rust tests read: the same calls, the same known answers (data transcribed from the reference's #[test]s; the JSON
// fixtures under tests/golden/ hold the full sets), plus GPU == oracle checks through the mirror.
//
// TEST INFRASTRUCTURE: links oracle/liboracle.so (the CPU restatement) as the checker. Needs a GPU.
//   mirror_test                 run the built-in tests
//   mirror_test --case FILE     solve_block on a decoded block written by tests/test_cpp_mirror.py; prints the result
//                               in a canonical text form that the Python mirror + oracle pipeline must reproduce
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/hiphase_gpu.hpp"
#include "../../oracle/hp_oracle.h"

using namespace hiphase;

static int g_checks = 0, g_failed = 0;
#define CHECK(...)                                                                         \
    do {                                                                                   \
        ++g_checks;                                                                        \
        if (!(__VA_ARGS__)) { ++g_failed; std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #__VA_ARGS__); } \
    } while (0)

// ---- read_segments.rs:213-308 ------------------------------------------------------------------------------------
static void test_read_segment_constructor() {   // fn test_constructor
    const ReadSegment rs("read_name", {3, 0, 1, 0, 0, 1, 2, 2, 3, 3}, {0, 1, 2, 3, 4, 5, 6, 7, 0, 0});
    CHECK(rs.region() == std::pair<size_t, size_t>(1, 6));
    CHECK(rs.alleles() == Bytes({0, 1, 0, 0, 1}));
    CHECK(rs.quals() == Bytes({1, 2, 3, 4, 5}));
}
static void test_read_segment_collapse() {   // fn test_collapse
    const ReadSegment rs1("read_name", {3, 1, 0, 2, 1, 3, 3}, {0, 2, 1, 0, 2, 0, 0});
    const ReadSegment rs2("read_name", {3, 3, 0, 1, 0, 1, 1}, {0, 0, 1, 2, 2, 1, 1});
    const ReadSegment c = ReadSegment::collapse({rs1, rs2});
    CHECK(c.region() == std::pair<size_t, size_t>(1, 7));
    CHECK(c.alleles() == Bytes({1, 0, 2, 2, 1, 1}));
    CHECK(c.quals() == Bytes({2, 1, 0, 0, 1, 1}));
}
static void test_read_segment_empty_and_num_set() {
    const ReadSegment none("r", {3, 2, 3}, {0, 0, 0});   // no set allele: empty region len..len (read_segments.rs:48-55)
    CHECK(none.region() == std::pair<size_t, size_t>(3, 3));
    CHECK(none.get_num_set() == 0);
    const ReadSegment rs("r", {3, 0, 1, 0, 0, 1, 2, 1, 3, 3}, {0, 1, 2, 3, 4, 5, 6, 7, 0, 0});
    CHECK(rs.get_num_set() == 6);
    CHECK(rs.allele(0) == 3 && rs.allele(7) == 1 && rs.allele(8) == 3 && rs.qual(8) == 0);
}

static void test_score_haplotype() {   // fn test_score_haplotype / fn test_score_partial_haplotype
    const ReadSegment rs("read_name", {3, 0, 1, 0, 0, 1, 2, 1, 3, 3}, {0, 1, 2, 3, 4, 5, 6, 7, 0, 0});
    CHECK(rs.region() == std::pair<size_t, size_t>(1, 8));
    CHECK(rs.score_haplotype({0, 0, 1, 0, 0, 1, 1, 1, 0, 0}) == 6);
    CHECK(rs.score_haplotype({2, 2, 2, 2, 2, 2, 2, 2, 2, 2}) == 0);
    CHECK(rs.score_haplotype({1, 1, 0, 1, 1, 0, 0, 0, 1, 1}) == 28);
    const ReadSegment rp("read_name", {2, 0, 1, 0, 0, 1, 2, 1, 2, 2}, {0, 1, 2, 3, 4, 5, 6, 7, 0, 0});
    CHECK(rp.score_partial_haplotype({0, 1, 0, 0, 1, 1, 1}, 1) == 6);
    CHECK(rp.score_partial_haplotype({2, 2, 2, 2, 2, 2, 2}, 2) == 0);
    CHECK(rp.score_partial_haplotype({1, 0, 1, 1, 0, 0, 0}, 1) == 28);
    CHECK(rp.score_partial_haplotype({0, 1, 1, 0, 0, 0}, 2) == 27);
    CHECK(rp.score_partial_haplotype({1, 1, 0, 0, 0}, 3) == 25);
    CHECK(rp.score_partial_haplotype({1, 0, 0, 0}, 4) == 22);
    CHECK(rp.score_partial_haplotype({0, 0, 0}, 5) == 18);
    CHECK(rp.score_partial_haplotype({0, 0}, 6) == 13);
    CHECK(rp.score_partial_haplotype({0}, 7) == 7);
}

// ---- variants.rs:800-846 fn test_closest_allele ---------------------------------------------------------------------
static void test_closest_allele() {
    Variant v = Variant::new_insertion(0, 10, bytes("A"), bytes("AGT"), 0, 1);
    v.prefix = bytes("AC");     // add_reference_prefix / add_reference_postfix (variants.rs:497-539)
    v.postfix = bytes("GGC");
    CHECK(v.get_allele0() == bytes("ACAGGC") && v.get_allele1() == bytes("ACAGTGGC"));
    struct K { const char* a; uint8_t allele; uint64_t lo, hi; };
    for (const K& k : {K{"A", 0, 5, 7}, K{"AGT", 0, 4, 5}, K{"AG", 0, 4, 6}, K{"ACAGGC", 0, 0, 2}, K{"ACAGTGGC", 1, 0, 2}, K{"ACAGGGC", 2, 1, 1}}) {
        const ClosestAllele c = closest_allele_clip(v, bytes(k.a));
        CHECK(c.allele == k.allele && c.min_ed == k.lo && c.other_ed == k.hi);
    }
    CHECK(v.match_allele(bytes("ACAGGC")) == 0 && v.match_allele(bytes("ACAGTGGC")) == 1 && v.match_allele(bytes("A")) == 2);
}

// ---- sequence_alignment.rs:45-76 ---------------------------------------------------------------------------------
static void test_edit_distance() {   // fn test_edit_distance (the HIP Levenshtein kernel behind the same signature)
    const Bytes v1{0, 1, 2, 4, 5}, v2{0, 1, 3, 4, 5}, v3{1, 2, 3, 5}, v4{};
    CHECK(edit_distance(v1, v1) == 0);
    CHECK(edit_distance(v1, v2) == 1);
    CHECK(edit_distance(v1, v3) == 2);
    CHECK(edit_distance(v1, v4) == 5);
    CHECK(edit_distance(v2, v3) == 3);
    CHECK(edit_distance(v3, v4) == 4);
    CHECK(edit_distance(v4, v4) == 0);
    CHECK(edit_distance(bytes("AAAAAAAAAAAAAAAAACAAA"), bytes("AAAAAAAAAAAAAAAACAAA")) == 1);
}

// ---- wfa_graph.rs:842 fn test_simple_snv, observed through the allele mapping of read_parsing.rs:790-800 ----------
static void test_simple_snv() {
    const Bytes reference = bytes("AAA");
    const std::vector<Variant> variants{Variant::new_snv(0, 1, bytes("A"), bytes("C"), 0, 1)};
    const Bytes q0 = bytes("AAA"), q1 = bytes("ACA"), q2 = bytes("AA");
    std::vector<WfaJob> jobs(3);
    const Bytes* reads[3] = {&q0, &q1, &q2};
    for (int i = 0; i < 3; ++i) {
        jobs[i].reference = &reference; jobs[i].ref_start = 0; jobs[i].ref_end = 3;
        jobs[i].hets = variants.data(); jobs[i].n_hets = 1; jobs[i].read = reads[i];
    }
    const auto r = global_realignment_batch(jobs, 0, 1000);
    CHECK(!r[0].max_edit_distance && r[0].score == 0 && r[0].num_nodes == 4 && r[0].alleles == Bytes({0}));   // nodes {0,2,3}
    CHECK(!r[1].max_edit_distance && r[1].score == 0 && r[1].alleles == Bytes({1}));                           // nodes {0,1,3}
    CHECK(!r[2].max_edit_distance && r[2].score == 1 && r[2].alleles == Bytes({2}));   // both branches tie -> Ambiguous
    const auto capped = global_realignment_batch({jobs[2]}, 0, 0);   // Err(MaxEditDistance) (wfa_graph.rs:645-648)
    CHECK(capped[0].max_edit_distance);
}

// ---- phaser.rs:756-804 -------------------------------------------------------------------------------------------
static void test_span_counts_and_haplotags() {
    {   // fn test_get_solution_span_counts
        const std::vector<ReadSegment> reads{
            ReadSegment("r1", {0, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 1}),
            ReadSegment("r2", {3, 3, 3, 1, 1, 3}, {0, 0, 0, 1, 1, 0}),
            ReadSegment("r3", {1, 1, 1, 1, 3, 3}, {1, 1, 1, 1, 0, 0}),
            ReadSegment("r4", {3, 1, 1, 1, 1, 1}, {0, 1, 1, 1, 1, 1}),
        };
        const auto c = get_solution_span_counts(reads, {0, 1, 1, 0, 0, 0}, {1, 1, 1, 1, 0, 1});
        CHECK((c == std::vector<uint64_t>{2, 2, 2, 2, 2}));
    }
    {   // fn test_haplotag_reads
        const std::vector<ReadSegment> reads{
            ReadSegment("r1", {0, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 1}),
            ReadSegment("r2", {2, 2, 2, 1, 1, 2}, {0, 0, 0, 1, 1, 0}),
            ReadSegment("r3", {2, 2, 2, 1, 0, 2}, {0, 0, 0, 1, 1, 0}),
            ReadSegment("r4", {2, 2, 2, 1, 0, 1}, {0, 0, 0, 1, 1, 1}),
            ReadSegment("r5", {2, 2, 2, 1, 0, 2}, {0, 0, 0, 2, 1, 0}),
        };
        const auto t = haplotag_reads(reads, {0, 0, 0, 0, 0, 0}, {1, 1, 1, 1, 1, 1}, {0, 0, 0, 3, 3, 5});
        CHECK(t.size() == 4);   // r3 ties: untagged
        CHECK(t[0].read_name == "r1" && t[0].phase_block == 0 && t[0].haplotag == 0);
        CHECK(t[1].read_name == "r2" && t[1].phase_block == 3 && t[1].haplotag == 1);
        CHECK(t[2].read_name == "r4" && t[2].phase_block == 3 && t[2].haplotag == 1);
        CHECK(t[3].read_name == "r5" && t[3].phase_block == 3 && t[3].haplotag == 1);
    }
}

// ---- astar_solver: no known-answer test exists upstream (only node costs are pinned); GPU == oracle ----------------
static std::vector<ReadSegment> synth_segments(uint32_t n, uint32_t c, uint32_t s, double e, uint64_t seed, Bytes& flags) {
    hp_synth_spec spec{n, c, s, 0, e, 0.02, seed};
    uint64_t n_cells = 0;
    const uint32_t R = hp_synth_block_size(&spec, &n_cells);
    std::vector<uint32_t> rs(R), re(R);
    std::vector<uint64_t> off(R + 1);
    Bytes a2((n_cells + 3) / 4 + 1), q(n_cells + 1), truth(n);
    flags.assign(n, 0);
    if (hp_synth_block(&spec, rs.data(), re.data(), off.data(), a2.data(), q.data(), flags.data(), truth.data()) != 0) throw std::runtime_error("synth");
    std::vector<ReadSegment> segs;
    for (uint32_t r = 0; r < R; ++r) {   // rows as the reference holds them: full-length, clipped by ReadSegment::new
        Bytes al(n, 3), ql(n, 0);
        for (uint32_t i = rs[r]; i < re[r]; ++i) {
            const uint64_t cell = off[r] + (i - rs[r]);
            al[i] = (a2[cell >> 2] >> (2 * (cell & 3))) & 3;
            ql[i] = q[cell];
        }
        segs.emplace_back("read_" + std::to_string(r), al, ql);
    }
    return segs;
}
static void test_astar_solver_against_oracle() {
    struct Cfg { uint32_t n, c, s; double e; uint64_t seed; };
    for (const Cfg& cf : {Cfg{50, 8, 20, 0.01, 1}, Cfg{400, 30, 20, 0.01, 20250509}, Cfg{300, 30, 20, 0.10, 2}, Cfg{250, 60, 20, 0.15, 3}}) {
        Bytes flags;
        const auto segs = synth_segments(cf.n, cf.c, cf.s, cf.e, cf.seed, flags);
        const AstarResult got = astar_solver(7, flags, segs, 1000, 3);
        const BlockMatrix m(segs, flags);
        const hp_block_view v = m.view();
        hp_astar_params p{1000, 3, 0, 7};
        Bytes h1(cf.n), h2(cf.n);
        hp_phase_stats st{};
        CHECK(hpo_astar_solve(&v, &p, h1.data(), h2.data(), &st, nullptr, nullptr) == 0);
        CHECK(got.haplotype_1 == h1 && got.haplotype_2 == h2);
        CHECK((got.statistics == PhaseStats{st.pruned_solutions, st.estimated_cost, st.actual_cost, st.phased_variants, st.phased_snvs,
                                            st.homozygous_variants, st.skipped_variants}));
        // invariants the reference asserts: the heuristic is a lower bound; counts partition the block
        CHECK(got.statistics.estimated_cost <= got.statistics.actual_cost);
        CHECK(got.statistics.phased_variants + got.statistics.homozygous_variants + got.statistics.skipped_variants == cf.n);
    }
}
// ---- wfa_graph.rs:747-787, 814-838: hand-built graphs through WFAGraph::add_node + edit_distance_with_pruning -------
static void test_wfa_hand_built_graphs() {
    {   // fn test_triple_split
        WFAGraph g;
        const size_t n0 = g.add_node({0, 1}, {});
        const size_t n1 = g.add_node({2, 3}, {n0}), n2 = g.add_node({2, 4}, {n0}), n3 = g.add_node({4}, {n0});
        const size_t n4 = g.add_node({4, 5}, {n1, n2, n3});
        WFAResult r = g.edit_distance_with_pruning({0, 1, 2, 3, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({n0, n1, n4}));
        r = g.edit_distance_with_pruning({0, 1, 2, 4, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({n0, n2, n4}));
        r = g.edit_distance_with_pruning({0, 1, 4, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({n0, n3, n4}));
    }
    {   // fn test_nested_split
        WFAGraph g;
        g.add_node({0, 1}, {}); g.add_node({2, 3}, {0}); g.add_node({2}, {0}); g.add_node({4}, {0, 2}); g.add_node({4, 5}, {1, 3});
        WFAResult r = g.edit_distance_with_pruning({0, 1, 2, 4, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({0, 2, 3, 4}));
        r = g.edit_distance_with_pruning({0, 1, 4, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({0, 3, 4}));
    }
    {   // fn test_overlapping_split
        WFAGraph g;
        g.add_node({0}, {}); g.add_node({1}, {0}); g.add_node({2}, {1}); g.add_node({3}, {0, 2}); g.add_node({4, 5}, {1, 3});
        WFAResult r = g.edit_distance_with_pruning({0, 1, 2, 3, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({0, 1, 2, 3, 4}));
        r = g.edit_distance_with_pruning({0, 3, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({0, 3, 4}));
        r = g.edit_distance_with_pruning({0, 1, 4, 5}, UINT64_MAX, 500);
        CHECK(r.score == 0 && r.traversed_nodes == std::vector<size_t>({0, 1, 4}));
        bool hit = false;
        r = g.edit_distance_with_pruning({7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7}, UINT64_MAX, 3, &hit);
        CHECK(hit && r.traversed_nodes.empty());
    }
}

static void test_errors() {
    bool threw = false;
    try { (void)astar_solver(0, Bytes{}, {}, 1000, 3); } catch (const Error& e) { threw = e.code < 0; }   // N == 0: malformed view
    CHECK(threw);
    threw = false;
    try { (void)Variant::new_snv(0, 5, bytes("AC"), bytes("G"), 0, 1); } catch (const std::invalid_argument&) { threw = true; }
    CHECK(threw);
}

// ---- --case FILE ---------------------------------------------------------------------------------------------------
static Bytes tok_bytes(const std::string& t) { return t == "-" ? Bytes{} : bytes(t); }
static Bytes hex_bytes(const std::string& t) {
    Bytes b;
    if (t == "-") return b;
    for (size_t i = 0; i + 1 < t.size(); i += 2) b.push_back((uint8_t)std::stoi(t.substr(i, 2), nullptr, 16));
    return b;
}
static std::string digits(const Bytes& b) { std::string s; for (uint8_t x : b) s += (char)('0' + x); return s; }
static int run_case(const char* path) {
    std::ifstream f(path);
    if (!f) { std::printf("cannot open %s\n", path); return 2; }
    std::string tag;
    uint64_t ref_base = 0, mma = 2, minq = 1000, qinc = 3, fail_min = 50;
    int global = 1;
    GlobalRealignmentConfig cfg;
    Bytes reference;
    std::vector<Variant> hets, homs;
    std::vector<AlignedRecord> records;
    while (f >> tag) {
        if (tag == "reference") { std::string s; f >> ref_base >> s; reference = bytes(s); }
        else if (tag == "params") { f >> mma >> minq >> qinc >> cfg.max_edit_distance >> cfg.wfa_prune_distance >> cfg.global_failure_ratio >> fail_min >> global; cfg.global_failure_minimum = fail_min; }
        else if (tag == "V" || tag == "H") {
            uint32_t type, ref_len, i0, i1; int64_t pos; int ignored; std::string a0, a1, pre, post;
            f >> type >> pos >> ref_len >> a0 >> a1 >> i0 >> i1 >> ignored >> pre >> post;
            Variant v = Variant::make(0, (VariantType)type, pos, ref_len, tok_bytes(a0), tok_bytes(a1), i0, i1);
            v.is_ignored = ignored != 0; v.prefix = tok_bytes(pre); v.postfix = tok_bytes(post);
            (tag == "V" ? hets : homs).push_back(v);
        } else if (tag == "R") {
            AlignedRecord r; std::string seq; int has_local;
            f >> r.qname >> r.min_position >> r.max_position >> seq >> has_local;
            r.read_align = tok_bytes(seq);
            r.has_local = has_local != 0;
            if (r.has_local) {
                size_t nc; std::string lseq, lqual;
                f >> r.local.pos >> nc;
                r.local.qname = r.qname;
                for (size_t k = 0; k < nc; ++k) { uint32_t c; f >> c; r.local.cigar.push_back(c); }
                f >> lseq >> lqual;
                r.local.seq = tok_bytes(lseq);
                r.local.qual = hex_bytes(lqual);
            }
            records.push_back(std::move(r));
        }
    }
    const PhaseResult pr = solve_block(7, records, hets, homs, reference, ref_base, mma, minq, qinc, &cfg, global != 0);
    std::printf("h1 %s\nh2 %s\n", digits(pr.haplotype_1).c_str(), digits(pr.haplotype_2).c_str());
    const PhaseStats& s = pr.statistics;
    std::printf("stats %llu %llu %llu %llu %llu %llu %llu\n", (unsigned long long)s.pruned_solutions, (unsigned long long)s.estimated_cost,
                (unsigned long long)s.actual_cost, (unsigned long long)s.phased_variants, (unsigned long long)s.phased_snvs,
                (unsigned long long)s.homozygous_variants, (unsigned long long)s.skipped_variants);
    std::printf("block_ids"); for (int64_t b : pr.block_ids) std::printf(" %lld", (long long)b); std::printf("\n");
    std::printf("sub_blocks"); for (auto& b : pr.sub_phase_blocks) { std::printf(" "); for (size_t k = 0; k < b.size(); ++k) std::printf(k ? ",%zu" : "%zu", b[k]); } std::printf("\n");
    std::printf("load %llu %llu %llu %llu\n", (unsigned long long)pr.load_stats.num_reads, (unsigned long long)pr.load_stats.skipped_reads,
                (unsigned long long)pr.load_stats.global_aligned, (unsigned long long)pr.load_stats.local_aligned);
    for (const auto& rs : pr.read_segments) {
        std::string q; char buf[4];
        for (uint8_t x : rs.quals()) { std::snprintf(buf, sizeof buf, "%02x", x); q += buf; }
        std::printf("segment %s %zu %zu %s %s\n", rs.read_name().c_str(), rs.start(), rs.end(), digits(rs.alleles()).c_str(), q.empty() ? "-" : q.c_str());
    }
    for (const auto& t : pr.haplotags) std::printf("haplotag %s %lld %u\n", t.read_name.c_str(), (long long)t.phase_block, (unsigned)t.haplotag);
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc == 3 && std::strcmp(argv[1], "--case") == 0) return run_case(argv[2]);
        if (hp_device_count() < 1) { std::printf("mirror_test needs a GPU: the library has no CPU fallback\n"); return 3; }
        test_read_segment_constructor();
        test_read_segment_collapse();
        test_read_segment_empty_and_num_set();
        test_score_haplotype();
        test_closest_allele();
        test_edit_distance();
        test_simple_snv();
        test_span_counts_and_haplotags();
        test_astar_solver_against_oracle();
        test_wfa_hand_built_graphs();
        test_errors();
    } catch (const std::exception& e) {
        std::printf("EXCEPTION: %s\n", e.what());
        return 1;
    }
    std::printf("mirror_test: %d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
