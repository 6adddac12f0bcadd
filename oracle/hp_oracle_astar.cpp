// hp_oracle_astar.cpp — TEST INFRASTRUCTURE ONLY (see hp_oracle.h).
//
// Line-faithful CPU restatement of
//   reference src/data_types/read_segments.rs  (ReadSegment, collapse, score_partial_haplotype)
//   reference src/astar_phaser.rs              (AstarNode, PQueueHapTracker, calculate_astar_heuristic,
//                                               astar_subsolver, astar_solver)
//   reference src/phaser.rs:350-388,714-750    (get_solution_span_counts, haplotag_reads)
// Behaviour is restated statement by statement (same child order, same priority tuple, same `<`/`<=`,
// same asserts) — including the O(len) haplotype copies per node, so that it can double as the timed
// "port" CPU baseline. No cleverness on purpose.
#include "hp_oracle.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace {

constexpr uint8_t REF = HP_ALLELE_REFERENCE, ALT = HP_ALLELE_ALTERNATE, AMB = HP_ALLELE_AMBIGUOUS,
                  NOOV = HP_ALLELE_NOOVERLAP;

struct InvariantError { const char* what; };
#define ORACLE_ASSERT(c) do { if (!(c)) throw InvariantError{#c}; } while (0)

// ---- read_segments.rs:19-207 -------------------------------------------------------------------
struct ReadSegment {
    std::vector<uint8_t> alleles;  // clipped
    std::vector<uint8_t> quals;    // clipped
    size_t start = 0, end = 0;     // region

    // read_segments.rs:40-62
    static ReadSegment make(const std::vector<uint8_t>& alleles, const std::vector<uint8_t>& quals) {
        ORACLE_ASSERT(alleles.size() == quals.size());
        size_t first_allele = alleles.size();
        for (size_t i = 0; i < alleles.size(); ++i) if (alleles[i] < AMB) { first_allele = i; break; }
        size_t last_allele = alleles.size();
        for (size_t i = alleles.size(); i-- > 0;) if (alleles[i] < AMB) { last_allele = i + 1; break; }
        ReadSegment rs;
        rs.start = first_allele;
        rs.end = last_allele;
        rs.alleles.assign(alleles.begin() + first_allele, alleles.begin() + last_allele);
        rs.quals.assign(quals.begin() + first_allele, quals.begin() + last_allele);
        return rs;
    }
    bool contains(size_t i) const { return i >= start && i < end; }
    // read_segments.rs:128-143
    uint8_t allele(size_t i) const { return contains(i) ? alleles[i - start] : NOOV; }
    uint8_t qual(size_t i) const { return contains(i) ? quals[i - start] : 0; }
    // read_segments.rs:177-206
    uint64_t score_partial_haplotype(const uint8_t* hap, size_t hap_len, size_t offset) const {
        if (hap_len + offset <= start || offset >= end) return 0;
        size_t min_compare = std::max(start, offset);
        size_t max_compare = std::min(end, offset + hap_len);
        uint64_t sum = 0;
        for (size_t index = min_compare; index < max_compare; ++index) {
            uint8_t a = allele(index);
            if (hap[index - offset] < AMB && a != hap[index - offset]) sum += qual(index);
        }
        return sum;
    }
    // read_segments.rs:161-168
    uint64_t score_haplotype(const uint8_t* hap, size_t hap_len) const {
        ORACLE_ASSERT(end <= hap_len);
        return score_partial_haplotype(hap, hap_len, 0);
    }
};

// read_segments.rs:71-121
ReadSegment collapse(const std::vector<ReadSegment>& segs) {
    ORACLE_ASSERT(!segs.empty());
    if (segs.size() == 1) return segs[0];
    size_t min_start = segs[0].start, max_end = segs[0].end;
    for (auto& rs : segs) { min_start = std::min(min_start, rs.start); max_end = std::max(max_end, rs.end); }
    std::vector<uint8_t> alleles(max_end, NOOV), quals(max_end, 0);
    for (auto& rs : segs) {
        for (size_t i = min_start; i < max_end; ++i) {
            uint8_t rsa = rs.allele(i), rsq = rs.qual(i);
            if (rsa != NOOV) {
                if (alleles[i] == NOOV) { alleles[i] = rsa; quals[i] = rsq; }
                else if (alleles[i] == AMB) { /* already ambiguous */ }
                else if (alleles[i] == rsa) { quals[i] = std::max(quals[i], rsq); ORACLE_ASSERT(quals[i] > 0); }
                else { alleles[i] = AMB; quals[i] = 0; }
            }
        }
    }
    return ReadSegment::make(alleles, quals);
}

// The IntervalTree<usize, ReadSegment> of the reference, reduced to what the solver asks of it:
// find(p..p+1) (astar_phaser.rs:92) and find(0..usize::MAX) (astar_phaser.rs:435; phaser.rs:362,720).
struct ReadSet {
    std::vector<ReadSegment> reads;
    std::vector<std::vector<uint32_t>> reads_at;  // per variant: rows whose region contains it
    size_t n_variants = 0;
};

ReadSet load_view(const hp_block_view* v) {
    ReadSet rs;
    rs.n_variants = v->n_variants;
    rs.reads.resize(v->n_reads);
    rs.reads_at.resize(v->n_variants);
    for (uint32_t r = 0; r < v->n_reads; ++r) {
        ReadSegment& s = rs.reads[r];
        s.start = v->read_start[r];
        s.end = v->read_end[r];
        size_t len = s.end - s.start;
        ORACLE_ASSERT(v->row_off[r + 1] - v->row_off[r] == len);
        s.alleles.resize(len);
        s.quals.resize(len);
        for (size_t c = 0; c < len; ++c) {
            uint64_t cell = v->row_off[r] + c;
            s.alleles[c] = (v->alleles_2bit[cell >> 2] >> (2 * (cell & 3))) & 3;
            s.quals[c] = v->quals[cell];
        }
        for (size_t p = s.start; p < s.end; ++p) rs.reads_at[p].push_back(r);
    }
    return rs;
}

// ---- astar_phaser.rs:13-166 --------------------------------------------------------------------
struct AstarNode {
    uint64_t node_index = 0, frozen_cost = 0, fluid_cost = 0, heuristic_cost = 0;
    std::vector<uint8_t> h1, h2;
    uint64_t num_hets = 0;
    uint64_t total() const { return frozen_cost + fluid_cost + heuristic_cost; }
    size_t allele_count() const { return h1.size(); }
    bool identical() const { return h1 == h2; }
};

struct Counters { uint64_t sub_pops = 0, main_pops = 0, evals = 0, cells = 0, nodes = 0; };

// astar_phaser.rs:47-57
std::unique_ptr<AstarNode> node_new(uint64_t max_heuristic) {
    auto n = std::make_unique<AstarNode>();
    n->heuristic_cost = max_heuristic;
    return n;
}

// astar_phaser.rs:69-119
std::unique_ptr<AstarNode> new_extended_node(uint64_t node_index, const AstarNode& parent, uint8_t a1, uint8_t a2,
                                             uint64_t heuristic_cost, const ReadSet& reads, size_t hap_offset,
                                             Counters& ctr) {
    auto n = std::make_unique<AstarNode>();
    n->h1 = parent.h1; n->h1.push_back(a1);
    n->h2 = parent.h2; n->h2.push_back(a2);
    ORACLE_ASSERT(n->h1.size() == n->h2.size());
    n->num_hets = parent.num_hets + (a1 == a2 ? 0 : 1);
    uint64_t frozen = parent.frozen_cost, fluid = 0;
    size_t hap_len = n->h1.size() + hap_offset;
    for (uint32_t r : reads.reads_at[hap_len - 1]) {
        const ReadSegment& rs = reads.reads[r];
        uint64_t c = std::min(rs.score_partial_haplotype(n->h1.data(), n->h1.size(), hap_offset),
                              rs.score_partial_haplotype(n->h2.data(), n->h2.size(), hap_offset));
        if (rs.end <= hap_len) frozen += c; else fluid += c;
        ctr.evals += 1;
        ctr.cells += std::min(rs.end, hap_len) - std::max(rs.start, hap_offset);
    }
    n->node_index = node_index;
    n->frozen_cost = frozen;
    n->fluid_cost = fluid;
    n->heuristic_cost = heuristic_cost;
    ctr.nodes += 1;
    return n;
}

// PriorityQueue<AstarNode, (Reverse<u64>, u64, Reverse<u64>)> (astar_phaser.rs:131-138,316,460).
// Pop order: smallest cost, then largest num_hets, then smallest node_index. node_index is unique,
// so the order is total and the container's internal tie-breaking cannot leak.
struct QEntry {
    uint64_t cost, hets, idx;
    AstarNode* node;
};
struct QLess {  // "a has lower priority than b" for std::*_heap (max-heap on priority)
    bool operator()(const QEntry& a, const QEntry& b) const {
        if (a.cost != b.cost) return a.cost > b.cost;
        if (a.hets != b.hets) return a.hets < b.hets;
        return a.idx > b.idx;
    }
};
struct PQueue {
    std::vector<QEntry> v;
    ~PQueue() { for (auto& e : v) delete e.node; }
    void push(std::unique_ptr<AstarNode> n) {
        QEntry e{n->total(), n->num_hets, n->node_index, n.release()};
        v.push_back(e);
        std::push_heap(v.begin(), v.end(), QLess());
    }
    const QEntry& peek() const { return v.front(); }
    std::unique_ptr<AstarNode> pop() {
        std::pop_heap(v.begin(), v.end(), QLess());
        AstarNode* n = v.back().node;
        v.pop_back();
        return std::unique_ptr<AstarNode>(n);
    }
    size_t len() const { return v.size(); }
};

// astar_phaser.rs:171-231
struct PQueueHapTracker {
    std::vector<uint64_t> length_counts;
    uint64_t total_count = 0;
    size_t threshold = 0;
    explicit PQueueHapTracker(size_t max_hap_length) : length_counts(max_hap_length + 1, 0) {}
    void add_hap(size_t value) { length_counts[value] += 1; if (value >= threshold) total_count += 1; }
    void remove_hap(size_t value) {
        ORACLE_ASSERT(length_counts[value] > 0);
        length_counts[value] -= 1;
        if (value >= threshold) { ORACLE_ASSERT(total_count > 0); total_count -= 1; }
    }
    void increase_threshold(size_t new_threshold) {
        ORACLE_ASSERT(new_threshold >= threshold);
        for (size_t t = threshold; t < new_threshold; ++t) total_count -= length_counts[t];
        threshold = new_threshold;
    }
    uint64_t len() const { return total_count; }
};

const uint8_t HAP_ORDER[4][2] = {{REF, ALT}, {ALT, REF}, {REF, REF}, {ALT, ALT}};  // astar_phaser.rs:367-372,535-540

// astar_phaser.rs:311-405
std::pair<uint64_t, size_t> astar_subsolver(size_t problem_offset, size_t problem_size, const ReadSet& reads,
                                            const std::vector<uint64_t>& heuristic_costs,
                                            const std::vector<uint8_t>& bad_variants, size_t min_queue_size,
                                            size_t queue_increment, Counters& ctr) {
    PQueue pqueue;
    uint64_t next_node_index = 1;
    ORACLE_ASSERT(heuristic_costs[problem_offset] == 0);
    uint64_t initial_estimate = heuristic_costs[problem_offset + 1];
    pqueue.push(node_new(initial_estimate));
    size_t next_expected = 0;
    uint64_t max_cost_so_far = 0;
    size_t max_visits = min_queue_size + queue_increment * problem_size;
    size_t nodes_visited = 0;

    while (pqueue.peek().node->allele_count() < problem_size && nodes_visited < max_visits) {
        auto top_node = pqueue.pop();
        size_t allele_count = top_node->allele_count();
        nodes_visited += 1;
        ctr.sub_pops += 1;
        if (allele_count == next_expected) {
            max_cost_so_far = std::max(max_cost_so_far, top_node->total());
            next_expected += 1;
        }
        if (bad_variants[problem_offset + allele_count]) {
            auto new_node = new_extended_node(next_node_index, *top_node, AMB, AMB,
                                              heuristic_costs[problem_offset + allele_count + 1], reads,
                                              problem_offset, ctr);
            next_node_index += 1;
            ORACLE_ASSERT(top_node->total() == new_node->total());
            pqueue.push(std::move(new_node));
        } else {
            for (auto& ho : HAP_ORDER) {
                if (!(ho[0] == ALT && ho[1] == REF && top_node->identical())) {
                    auto new_node = new_extended_node(next_node_index, *top_node, ho[0], ho[1],
                                                      heuristic_costs[problem_offset + allele_count + 1], reads,
                                                      problem_offset, ctr);
                    next_node_index += 1;
                    pqueue.push(std::move(new_node));
                }
            }
        }
    }
    if (pqueue.peek().node->allele_count() == problem_size) {
        max_cost_so_far = std::max(max_cost_so_far, pqueue.peek().node->total());
        next_expected += 1;
    }
    return {max_cost_so_far, next_expected - 1};
}

// astar_phaser.rs:246-292
std::vector<uint64_t> calculate_astar_heuristic(size_t num_variants, size_t max_segment_size, const ReadSet& reads,
                                                size_t min_queue_size, size_t queue_increment,
                                                std::vector<uint8_t>& bad_variants, Counters& ctr) {
    ORACLE_ASSERT(max_segment_size >= 2);
    std::vector<uint64_t> heuristics(num_variants + 1, 0);
    ORACLE_ASSERT(bad_variants.size() == num_variants);
    const bool bad_variants_enabled = false;
    size_t max_clip_size = 1;
    for (size_t v_index = num_variants; v_index-- > 0;) {
        auto [max_estimate, solve_size] = astar_subsolver(v_index, max_clip_size, reads, heuristics, bad_variants,
                                                         min_queue_size / 10, queue_increment, ctr);
        ORACLE_ASSERT(solve_size >= std::min<size_t>(max_clip_size, 2));
        if (bad_variants_enabled && solve_size < max_clip_size) bad_variants[v_index] = 1;
        if (bad_variants[v_index]) {
            heuristics[v_index] = heuristics[v_index + 1];
        } else {
            ORACLE_ASSERT(max_estimate >= heuristics[v_index + 1]);
            heuristics[v_index] = max_estimate;
        }
        max_clip_size = std::min(solve_size + 1, max_segment_size);
    }
    return heuristics;
}

struct AstarResult {
    std::vector<uint8_t> h1, h2;
    hp_phase_stats stats{};
    std::vector<uint64_t> heuristics;
};

// astar_phaser.rs:426-633
AstarResult astar_solver(const hp_block_view* view, const ReadSet& reads, size_t min_queue_size,
                         size_t queue_increment, size_t max_segment_size, Counters& ctr) {
    const size_t num_variants = view->n_variants;
    // astar_phaser.rs:435-442 sanity check
    for (size_t vi = 0; vi < num_variants; ++vi)
        if (view->var_flags[vi] & HP_VAR_IGNORED)
            for (auto& seg : reads.reads) ORACLE_ASSERT(seg.allele(vi) == NOOV);
    std::vector<uint8_t> bad_variants(num_variants);
    for (size_t i = 0; i < num_variants; ++i) bad_variants[i] = (view->var_flags[i] & HP_VAR_IGNORED) ? 1 : 0;

    size_t curr_queue_size_threshold = min_queue_size;
    const bool full_prune_enabled = true;
    const size_t max_queue_size = 10 * min_queue_size;
    size_t min_progress = 0;
    PQueue pqueue;
    PQueueHapTracker hap_tracker(num_variants);
    size_t next_expected = 0;

    std::vector<uint64_t> heuristic_costs = calculate_astar_heuristic(num_variants, max_segment_size, reads,
                                                                      min_queue_size, queue_increment, bad_variants, ctr);
    for (size_t i = 0; i < num_variants; ++i)
        if (view->var_flags[i] & HP_VAR_IGNORED) ORACLE_ASSERT(bad_variants[i]);

    uint64_t num_pruned = 0;
    const uint64_t estimated_cost = heuristic_costs[0];
    pqueue.push(node_new(heuristic_costs[0]));
    hap_tracker.add_hap(0);
    uint64_t next_node_index = 1;

    while (pqueue.peek().node->allele_count() < num_variants) {
        auto top_node = pqueue.pop();
        size_t allele_count = top_node->allele_count();
        hap_tracker.remove_hap(allele_count);
        ctr.main_pops += 1;
        if (allele_count == next_expected) {
            next_expected += 1;
            if (num_pruned == 0) {
                curr_queue_size_threshold += queue_increment;
                ORACLE_ASSERT(curr_queue_size_threshold == min_queue_size + queue_increment * next_expected);
            }
        }
        if (allele_count < min_progress) {
            if (num_pruned == 0) curr_queue_size_threshold = min_queue_size;
            num_pruned += 1;
            continue;
        }
        if (bad_variants[allele_count]) {
            auto new_node = new_extended_node(next_node_index, *top_node, AMB, AMB, heuristic_costs[allele_count + 1],
                                              reads, 0, ctr);
            next_node_index += 1;
            ORACLE_ASSERT(top_node->total() == new_node->total());
            pqueue.push(std::move(new_node));
            hap_tracker.add_hap(allele_count + 1);
        } else {
            for (auto& ho : HAP_ORDER) {
                if (!(ho[0] == ALT && ho[1] == REF && top_node->identical())) {
                    auto new_node = new_extended_node(next_node_index, *top_node, ho[0], ho[1],
                                                      heuristic_costs[allele_count + 1], reads, 0, ctr);
                    next_node_index += 1;
                    pqueue.push(std::move(new_node));
                    hap_tracker.add_hap(allele_count + 1);
                }
            }
        }
        while (hap_tracker.len() > curr_queue_size_threshold && min_progress < next_expected) {
            min_progress += 1;
            hap_tracker.increase_threshold(min_progress);
            if (full_prune_enabled && pqueue.len() > max_queue_size) {
                // astar_phaser.rs:576-582: iter_mut() rewrite + automatic re-heapify
                for (auto& e : pqueue.v)
                    if (e.node->allele_count() < min_progress) e.cost = 0;  // get_cleared_priority
                std::make_heap(pqueue.v.begin(), pqueue.v.end(), QLess());
            }
        }
    }

    auto top_node = pqueue.pop();
    size_t allele_count = top_node->allele_count();
    hap_tracker.remove_hap(allele_count);
    if (allele_count != num_variants) throw InvariantError{"failed to find solution (astar_phaser.rs:631)"};
    AstarResult res;
    res.h1 = top_node->h1;
    res.h2 = top_node->h2;
    uint64_t actual_cost = top_node->total();
    uint64_t phased = 0, phased_snvs = 0, homozygous = 0, skipped = 0;
    for (size_t i = 0; i < num_variants; ++i) {
        uint8_t a = res.h1[i], b = res.h2[i];
        if (a != b) { phased += 1; if (view->var_flags[i] & HP_VAR_SNV) phased_snvs += 1; }
        else if (a == AMB) skipped += 1;
        else homozygous += 1;
    }
    ORACLE_ASSERT(actual_cost >= estimated_cost);  // phase_stats.rs:163
    res.stats = hp_phase_stats{num_pruned, estimated_cost, actual_cost, phased, phased_snvs, homozygous, skipped};
    res.heuristics = std::move(heuristic_costs);
    return res;
}

void fill_counters(hp_work_counters* out, const Counters& c) {
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    out->sub_pops = c.sub_pops; out->main_pops = c.main_pops; out->evals = c.evals;
    out->cells = c.cells; out->nodes_created = c.nodes;
}
size_t seg_or_default(const hp_astar_params* p) { return p->max_segment_size ? (size_t)p->max_segment_size : 40; }

}  // namespace

extern "C" {

void hpo_read_segment_new(const uint8_t* alleles, size_t len, size_t* start, size_t* end) {
    std::vector<uint8_t> a(alleles, alleles + len), q(len, 0);
    ReadSegment rs = ReadSegment::make(a, q);
    *start = rs.start; *end = rs.end;
}

int hpo_read_segment_collapse(const uint8_t* alleles, const uint8_t* quals, size_t k, size_t len,
                              uint8_t* out_alleles, uint8_t* out_quals, size_t* start, size_t* end) {
    try {
        std::vector<ReadSegment> segs;
        for (size_t i = 0; i < k; ++i) {
            std::vector<uint8_t> a(alleles + i * len, alleles + (i + 1) * len), q(quals + i * len, quals + (i + 1) * len);
            segs.push_back(ReadSegment::make(a, q));
        }
        ReadSegment c = collapse(segs);
        for (size_t i = 0; i < len; ++i) { out_alleles[i] = c.allele(i); out_quals[i] = c.qual(i); }
        *start = c.start; *end = c.end;
        return HP_OK;
    } catch (const InvariantError&) { return HP_ERR_INVARIANT; }
}

uint64_t hpo_score_partial_haplotype(const uint8_t* row_alleles, const uint8_t* row_quals, size_t start, size_t end,
                                     const uint8_t* haplotype, size_t hap_len, size_t offset) {
    ReadSegment rs;
    rs.start = start; rs.end = end;
    rs.alleles.assign(row_alleles, row_alleles + (end - start));
    rs.quals.assign(row_quals, row_quals + (end - start));
    return rs.score_partial_haplotype(haplotype, hap_len, offset);
}

int hpo_astar_node_walk(const hp_block_view* blk, const uint8_t* path1, const uint8_t* path2, size_t len,
                        const uint64_t* heuristic_costs, size_t hap_offset, uint64_t* frozen, uint64_t* total,
                        uint64_t* num_hets) {
    try {
        ReadSet reads = load_view(blk);
        Counters ctr;
        std::unique_ptr<AstarNode> cur = node_new(heuristic_costs[0]);
        for (size_t i = 0; i < len; ++i) {
            auto nxt = new_extended_node(i + 1, *cur, path1[i], path2[i], heuristic_costs[i + 1], reads, hap_offset, ctr);
            frozen[i] = nxt->frozen_cost; total[i] = nxt->total(); num_hets[i] = nxt->num_hets;
            cur = std::move(nxt);
        }
        return HP_OK;
    } catch (const InvariantError&) { return HP_ERR_INVARIANT; }
}

int hpo_hap_tracker_script(size_t max_hap_length, const int32_t* ops, const uint64_t* values, size_t n, uint64_t* out_len) {
    try {
        PQueueHapTracker t(max_hap_length);
        for (size_t i = 0; i < n; ++i) {
            if (ops[i] == 0) t.add_hap(values[i]);
            else if (ops[i] == 1) t.remove_hap(values[i]);
            else t.increase_threshold(values[i]);
            out_len[i] = t.len();
        }
        return HP_OK;
    } catch (const InvariantError&) { return HP_ERR_INVARIANT; }
}

int hpo_astar_heuristic(const hp_block_view* blk, const hp_astar_params* p, uint64_t* heuristics) {
    try {
        ReadSet reads = load_view(blk);
        Counters ctr;
        std::vector<uint8_t> bad(blk->n_variants);
        for (uint32_t i = 0; i < blk->n_variants; ++i) bad[i] = (blk->var_flags[i] & HP_VAR_IGNORED) ? 1 : 0;
        auto h = calculate_astar_heuristic(blk->n_variants, seg_or_default(p), reads, p->min_queue_size,
                                           p->queue_increment, bad, ctr);
        std::memcpy(heuristics, h.data(), h.size() * sizeof(uint64_t));
        return HP_OK;
    } catch (const InvariantError&) { return HP_ERR_INVARIANT; }
}

int hpo_astar_solve(const hp_block_view* blk, const hp_astar_params* p, uint8_t* h1, uint8_t* h2,
                    hp_phase_stats* out, hp_work_counters* counters, uint64_t* heuristics) {
    try {
        if (blk->n_variants == 0) return HP_ERR_ARG;
        ReadSet reads = load_view(blk);
        Counters ctr;
        AstarResult r = astar_solver(blk, reads, p->min_queue_size, p->queue_increment, seg_or_default(p), ctr);
        std::memcpy(h1, r.h1.data(), r.h1.size());
        std::memcpy(h2, r.h2.data(), r.h2.size());
        if (out) *out = r.stats;
        fill_counters(counters, ctr);
        if (heuristics) std::memcpy(heuristics, r.heuristics.data(), r.heuristics.size() * sizeof(uint64_t));
        return HP_OK;
    } catch (const InvariantError&) { return HP_ERR_INVARIANT; }
}

uint64_t hpo_bruteforce_mec(const hp_block_view* blk) {
    ReadSet reads = load_view(blk);
    const size_t N = blk->n_variants;
    uint64_t best = UINT64_MAX;
    std::vector<uint8_t> h1(N), h2(N);
    for (uint64_t m1 = 0; m1 < (1ull << N); ++m1) {
        for (size_t i = 0; i < N; ++i) h1[i] = (m1 >> i) & 1;
        for (uint64_t m2 = 0; m2 < (1ull << N); ++m2) {
            for (size_t i = 0; i < N; ++i) h2[i] = (m2 >> i) & 1;
            uint64_t cost = 0;
            for (auto& rs : reads.reads)
                cost += std::min(rs.score_partial_haplotype(h1.data(), N, 0), rs.score_partial_haplotype(h2.data(), N, 0));
            best = std::min(best, cost);
        }
    }
    return best;
}

// phaser.rs:350-388
int hpo_solution_span_counts(const hp_block_view* blk, const uint8_t* h1, const uint8_t* h2, uint64_t* out) {
    try {
        ReadSet reads = load_view(blk);
        const size_t N = blk->n_variants;
        for (size_t i = 0; i + 1 < N; ++i) out[i] = 0;
        for (auto& rs : reads.reads) {
            if (rs.start == rs.end) continue;  // an empty region cannot be in the interval tree
            size_t js = rs.start, je = rs.end - 1;
            while (js < je && h1[js] == h2[js]) js += 1;
            while (js < je && h1[je] == h2[je]) je -= 1;
            for (size_t j = js; j < je; ++j) out[j] += 1;
        }
        return HP_OK;
    } catch (const InvariantError&) { return HP_ERR_INVARIANT; }
}

// phaser.rs:714-750
int hpo_haplotag_reads(const hp_block_view* blk, const uint8_t* h1, const uint8_t* h2, const uint64_t* block_tags,
                       uint8_t* haplotag, uint64_t* phase_block) {
    try {
        ReadSet reads = load_view(blk);
        const size_t N = blk->n_variants;
        for (size_t r = 0; r < reads.reads.size(); ++r) {
            const ReadSegment& rs = reads.reads[r];
            haplotag[r] = 2; phase_block[r] = 0;
            if (rs.start == rs.end) continue;
            uint64_t a1 = rs.score_haplotype(h1, N), a2 = rs.score_haplotype(h2, N);
            uint8_t tag = a1 < a2 ? 0 : (a1 > a2 ? 1 : 2);
            if (tag != 2) {
                size_t first_variant = rs.start;
                while (h1[first_variant] == h2[first_variant] || rs.allele(first_variant) >= AMB) first_variant += 1;
                haplotag[r] = tag;
                phase_block[r] = block_tags[first_variant];
            }
        }
        return HP_OK;
    } catch (const InvariantError&) { return HP_ERR_INVARIANT; }
}

}  // extern "C"
