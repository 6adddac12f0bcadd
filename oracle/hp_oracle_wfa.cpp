// hp_oracle_wfa.cpp — TEST INFRASTRUCTURE ONLY (see hp_oracle.h).
//
// Line-faithful CPU restatement of
//   reference src/wfa_graph.rs               (WFANode, WFAGraph::add_node, from_reference_variants_with_hom,
//                                             edit_distance_with_pruning, WFAResult)
//   reference src/sequence_alignment.rs:7-38 (edit_distance)
//   reference src/read_parsing.rs:790-800    (traversed nodes -> per-het AlleleType)
// Hash maps are kept as hash maps (std::unordered_map); an optional shuffle randomises every iteration
// the reference performs over a map so tests can show results do not depend on iteration order.
#include "hp_oracle.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <set>
#include <unordered_map>
#include <vector>

namespace {

struct WFANode {
    std::vector<uint8_t> sequence;
    std::vector<size_t> parent_nodes;  // sorted (wfa_graph.rs:36)
};

using Wave = std::pair<size_t, size_t>;                       // (offset, set index)
using DiagMap = std::unordered_map<int64_t, std::vector<Wave>>;

struct Shuffler {
    uint64_t x;
    bool on;
    explicit Shuffler(uint64_t seed) : x(seed), on(seed != 0) {}
    uint64_t next() {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    template <class T> void shuffle(std::vector<T>& v) {
        if (!on) return;
        for (size_t i = v.size(); i > 1; --i) std::swap(v[i - 1], v[next() % i]);
    }
};

}  // namespace

struct hpo_graph {
    std::vector<WFANode> nodes;
    std::vector<std::vector<size_t>> edges;
    uint64_t max_edit_distance = 1000;  // WFAGraph::default (wfa_graph.rs:70-74)
    std::map<size_t, std::vector<std::pair<size_t, uint8_t>>> node_to_alleles;  // NodeAlleleMap (wfa_graph.rs:19)

    // wfa_graph.rs:298-331
    int64_t add_node(std::vector<uint8_t> sequence, std::vector<size_t> parent_nodes) {
        size_t new_index = nodes.size();
        if (new_index == 0) {
            if (!parent_nodes.empty()) return -1;  // "First node must have no parent nodes."
        } else {
            if (parent_nodes.empty()) return -2;   // "All nodes after the first must have at least one parent node."
            for (size_t pn : parent_nodes) if (new_index <= pn) return -3;  // "All parent nodes must come before this node."
        }
        for (size_t p : parent_nodes) edges[p].push_back(new_index);
        std::sort(parent_nodes.begin(), parent_nodes.end());
        nodes.push_back(WFANode{std::move(sequence), std::move(parent_nodes)});
        edges.emplace_back();
        return (int64_t)new_index;
    }
};

namespace {

// wfa_graph.rs:119-284
hpo_graph* graph_from_job(const hp_wfa_job* job, uint64_t max_edit_distance, int* status) {
    auto* g = new hpo_graph();
    g->max_edit_distance = max_edit_distance;
    auto fail = [&](int s) { if (status) *status = s; delete g; return (hpo_graph*)nullptr; };
    auto ref_slice = [&](size_t a, size_t b) {
        return std::vector<uint8_t>(job->reference + (a - job->ref_base), job->reference + (b - job->ref_base));
    };
    const size_t ref_start = job->ref_start, ref_end = job->ref_end;
    size_t previous_end = ref_start;
    int64_t reference_index;
    std::vector<size_t> reference_reconnect;
    std::vector<std::pair<size_t, uint8_t>> reference_alleles;
    // PriorityQueue<usize, Reverse<usize>>: pop smallest reconnect position; ties in any order
    std::set<std::pair<size_t, size_t>> reconnect_queue;  // (reconnect, alt_index)

    struct VarRef { const hp_wfa_variant* v; int64_t index; };
    std::vector<VarRef> all_variants;
    for (uint32_t i = 0; i < job->n_hets; ++i) all_variants.push_back({&job->hets[i], (int64_t)i});
    for (uint32_t i = 0; i < job->n_homs; ++i) all_variants.push_back({&job->homs[i], -1});
    std::stable_sort(all_variants.begin(), all_variants.end(),
                     [](const VarRef& a, const VarRef& b) { return a.v->position < b.v->position; });

    auto drain_one = [&]() -> bool {  // wfa_graph.rs:168-189 / 256-272
        auto it = reconnect_queue.begin();
        size_t alt_reconnect = it->first, alt_index = it->second;
        reconnect_queue.erase(it);
        if (!(alt_reconnect > previous_end)) return false;
        reference_index = g->add_node(ref_slice(previous_end, alt_reconnect), reference_reconnect);
        if (reference_index < 0) return false;
        if (!reference_alleles.empty()) { g->node_to_alleles[reference_index] = reference_alleles; reference_alleles.clear(); }
        previous_end = alt_reconnect;
        reference_reconnect = {(size_t)reference_index, alt_index};
        while (!reconnect_queue.empty() && reconnect_queue.begin()->first == alt_reconnect) {
            reference_reconnect.push_back(reconnect_queue.begin()->second);
            reconnect_queue.erase(reconnect_queue.begin());
        }
        return true;
    };

    for (auto& vr : all_variants) {
        const hp_wfa_variant* variant = vr.v;
        if (variant->flags & 1) continue;  // is_ignored
        size_t variant_pos = (size_t)variant->position;
        size_t ref_len = variant->ref_len;
        if (variant->position < (int64_t)ref_start) continue;
        if (variant_pos + ref_len > ref_end) continue;

        while (!reconnect_queue.empty() && reconnect_queue.begin()->first <= variant_pos)
            if (!drain_one()) return fail(HP_ERR_INVARIANT);

        if (previous_end < variant_pos || g->nodes.empty()) {
            reference_index = g->add_node(ref_slice(previous_end, variant_pos), reference_reconnect);
            if (reference_index < 0) return fail(HP_ERR_INVARIANT);
            if (!reference_alleles.empty()) { g->node_to_alleles[reference_index] = reference_alleles; reference_alleles.clear(); }
            reference_reconnect = {(size_t)reference_index};
            previous_end = variant_pos;
        } else if (previous_end != variant_pos) {
            return fail(HP_ERR_INVARIANT);  // assert!(previous_end == variant_pos)
        }

        if (variant->flags & 2) {  // convert_index(Reference) != 0: allele0 is an ALT too
            int64_t alt_index = g->add_node(std::vector<uint8_t>(variant->allele0, variant->allele0 + variant->allele0_len),
                                            reference_reconnect);
            if (alt_index < 0) return fail(HP_ERR_INVARIANT);
            if (vr.index >= 0) g->node_to_alleles[alt_index] = {{(size_t)vr.index, 0}};
            reconnect_queue.insert({variant_pos + ref_len, (size_t)alt_index});
        } else if (vr.index >= 0) {
            reference_alleles.push_back({(size_t)vr.index, 0});
        }
        int64_t alt_index = g->add_node(std::vector<uint8_t>(variant->allele1, variant->allele1 + variant->allele1_len),
                                        reference_reconnect);
        if (alt_index < 0) return fail(HP_ERR_INVARIANT);
        if (vr.index >= 0) g->node_to_alleles[alt_index] = {{(size_t)vr.index, 1}};
        reconnect_queue.insert({variant_pos + ref_len, (size_t)alt_index});
    }
    while (!reconnect_queue.empty())
        if (!drain_one()) return fail(HP_ERR_INVARIANT);
    if (!(previous_end <= ref_end)) return fail(HP_ERR_INVARIANT);
    if (g->add_node(ref_slice(previous_end, ref_end), reference_reconnect) < 0) return fail(HP_ERR_INVARIANT);
    if (!reference_alleles.empty()) return fail(HP_ERR_INVARIANT);
    if (status) *status = HP_OK;
    return g;
}

using BitSet = std::vector<uint64_t>;

struct SetTable {
    std::map<BitSet, size_t> treeset_to_index;
    std::vector<BitSet> index_to_treeset;
    size_t words;
    explicit SetTable(size_t n_nodes) : words((n_nodes + 63) / 64) {}
    BitSet empty() const { return BitSet(words, 0); }
    size_t intern(const BitSet& s) {
        auto it = treeset_to_index.find(s);
        if (it != treeset_to_index.end()) return it->second;
        index_to_treeset.push_back(s);
        treeset_to_index[s] = index_to_treeset.size() - 1;
        return index_to_treeset.size() - 1;
    }
    size_t union_of(std::vector<size_t> sets) {  // sort + dedup + OR (wfa_graph.rs:486-510,593-618)
        std::sort(sets.begin(), sets.end());
        sets.erase(std::unique(sets.begin(), sets.end()), sets.end());
        if (sets.size() == 1) return sets[0];
        BitSet u = empty();
        for (size_t s : sets) for (size_t w = 0; w < words; ++w) u[w] |= index_to_treeset[s][w];
        return intern(u);
    }
};

// What a dense-band kernel with lanes = a node's diagonals would execute (measurement aid, DESIGN.md 7: does a read's alignment have
// more than one wavefront's worth of diagonals per node?): per (round, node) visit the hull of the diagonals that kept a wave.
// [0] node visits, [1] sum of hull widths, [2] widest hull, [3] visits x ceil(width / 64), [4] visits x ceil(width / 256), [5] rounds,
// [6] most nodes visited in one round, [7] visits with a hull wider than 64
thread_local uint64_t g_hull_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};

// wfa_graph.rs:350-650
int edit_distance_with_pruning(const hpo_graph& g, const uint8_t* other, size_t other_len, uint64_t prune_distance,
                               uint64_t shuffle_seed, uint64_t* score_out, std::vector<size_t>* traversed_out) {
    Shuffler shuf(shuffle_seed);
    const size_t n_nodes = g.nodes.size();
    std::unordered_map<size_t, DiagMap> active_wavefronts, next_wavefronts;
    std::unordered_map<size_t, std::unordered_map<int64_t, size_t>> max_wavefronts;
    SetTable sets(n_nodes);
    {
        BitSet base = sets.empty();
        base[0] |= 1ull;
        sets.intern(base);
    }
    active_wavefronts[0][0].push_back({0, 0});
    size_t edit_distance = 0, farthest_progression = 0, min_progression = 0;

    for (;;) {
        uint64_t visits_this_round = 0;
        for (size_t node_index = 0; node_index < n_nodes; ++node_index) {
            auto act = active_wavefronts.find(node_index);
            if (act == active_wavefronts.end()) continue;
            int64_t hull_lo = INT64_MAX, hull_hi = INT64_MIN;
            const std::vector<uint8_t>& node_sequence = g.nodes[node_index].sequence;
            const size_t node_length = node_sequence.size();
            DiagMap wavefront = std::move(act->second);
            active_wavefronts.erase(act);
            auto& maxfront = max_wavefronts[node_index];

            std::vector<int64_t> diag_keys;
            for (auto& kv : wavefront) diag_keys.push_back(kv.first);
            shuf.shuffle(diag_keys);
            for (int64_t other_start : diag_keys) {
                std::vector<Wave>& vec_waves = wavefront[other_start];
                size_t max_offset = 0;
                for (auto& w : vec_waves) {
                    size_t& offset = w.first;
                    if (other_start + (int64_t)offset < 0) return HP_ERR_INVARIANT;
                    size_t other_position = (size_t)(other_start + (int64_t)offset);
                    while (offset < node_length && other_position < other_len && node_sequence[offset] == other[other_position]) {
                        offset += 1;
                        other_position += 1;
                    }
                    max_offset = std::max(max_offset, offset);
                }
                size_t& maxfront_record = maxfront.emplace(other_start, 0).first->second;
                if (max_offset < maxfront_record || (other_start + (int64_t)max_offset) < (int64_t)min_progression) continue;
                maxfront_record = max_offset;
                hull_lo = std::min(hull_lo, other_start); hull_hi = std::max(hull_hi, other_start);
                farthest_progression = std::max(farthest_progression, (size_t)(other_start + (int64_t)max_offset));

                std::vector<size_t> best_sets;
                for (auto& w : vec_waves) if (w.first == max_offset) best_sets.push_back(w.second);
                shuf.shuffle(best_sets);
                size_t best_set = sets.union_of(best_sets);

                if (max_offset == node_length) {
                    if (node_index == n_nodes - 1) {
                        if ((size_t)(other_start + (int64_t)max_offset) < other_len)
                            next_wavefronts[node_index][other_start + 1].push_back({max_offset, best_set});
                    } else {
                        int64_t new_offset = other_start + (int64_t)max_offset;
                        for (size_t successor_index : g.edges[node_index]) {
                            BitSet new_set = sets.index_to_treeset[best_set];
                            new_set[successor_index / 64] |= 1ull << (successor_index % 64);
                            size_t new_set_index = sets.intern(new_set);
                            active_wavefronts[successor_index][new_offset].push_back({0, new_set_index});
                        }
                    }
                } else {
                    DiagMap& node_wf = next_wavefronts[node_index];
                    node_wf[other_start - 1].push_back({max_offset + 1, best_set});
                    if ((size_t)(other_start + (int64_t)max_offset) < other_len) {
                        node_wf[other_start].push_back({max_offset + 1, best_set});
                        node_wf[other_start + 1].push_back({max_offset, best_set});
                    }
                }
            }

            if (hull_lo <= hull_hi) {
                const uint64_t w = (uint64_t)(hull_hi - hull_lo) + 1;
                g_hull_stats[0] += 1; g_hull_stats[1] += w; g_hull_stats[2] = std::max(g_hull_stats[2], w);
                g_hull_stats[3] += (w + 63) / 64; g_hull_stats[4] += (w + 255) / 256; g_hull_stats[7] += w > 64 ? 1 : 0;
                ++visits_this_round;
            }
            if (node_index == n_nodes - 1) {
                std::vector<size_t> final_hashsets;
                for (auto& kv : wavefront)
                    for (auto& w : kv.second)
                        if (w.first == node_length && (size_t)(kv.first + (int64_t)w.first) == other_len)
                            final_hashsets.push_back(w.second);
                if (!final_hashsets.empty()) {
                    shuf.shuffle(final_hashsets);
                    size_t best_set = sets.union_of(final_hashsets);
                    *score_out = edit_distance;
                    traversed_out->clear();
                    const BitSet& bs = sets.index_to_treeset[best_set];
                    for (size_t i = 0; i < n_nodes; ++i) if ((bs[i / 64] >> (i % 64)) & 1) traversed_out->push_back(i);
                    return HP_OK;
                }
            }
        }
        g_hull_stats[5] += 1; g_hull_stats[6] = std::max(g_hull_stats[6], visits_this_round);
        edit_distance += 1;
        active_wavefronts = std::move(next_wavefronts);
        next_wavefronts.clear();
        if (farthest_progression > prune_distance) min_progression = farthest_progression - (size_t)prune_distance;
        if (edit_distance > g.max_edit_distance) {
            *score_out = g.max_edit_distance;
            return HP_WFA_MAX_ED;
        }
    }
}

}  // namespace

extern "C" {

// the hull statistics above since the last call with reset != 0 (per thread: the calling thread's own alignments)
void hpo_wfa_hull_stats(uint64_t out[8], int reset) {
    for (int i = 0; i < 8; ++i) { out[i] = g_hull_stats[i]; if (reset) g_hull_stats[i] = 0; }
}

// sequence_alignment.rs:7-38
uint64_t hpo_edit_distance(const uint8_t* v1, size_t l1, const uint8_t* v2, size_t l2) {
    std::vector<size_t> row(l1 + 1, 0), prev_row(l1 + 1);
    for (size_t j = 0; j <= l1; ++j) prev_row[j] = j;
    for (size_t i = 0; i < l2; ++i) {
        uint8_t c2 = v2[i];
        row[0] = i + 1;
        for (size_t j = 0; j < l1; ++j) {
            uint8_t c1 = v1[j];
            size_t a = prev_row[j + 1] + 1, b = row[j] + 1, c = prev_row[j] + (c1 == c2 ? 0 : 1);
            row[j + 1] = std::min(a, std::min(b, c));
        }
        std::swap(row, prev_row);
    }
    return prev_row[l1];
}

hpo_graph* hpo_graph_new(uint64_t max_edit_distance) {
    auto* g = new hpo_graph();
    g->max_edit_distance = max_edit_distance;
    return g;
}
void hpo_graph_free(hpo_graph* g) { delete g; }

int64_t hpo_graph_add_node(hpo_graph* g, const uint8_t* seq, size_t len, const uint64_t* parents, size_t n_parents) {
    std::vector<size_t> p(parents, parents + n_parents);
    return g->add_node(std::vector<uint8_t>(seq, seq + len), std::move(p));
}

hpo_graph* hpo_graph_from_job(const hp_wfa_job* job, uint64_t max_edit_distance, int* status) {
    return graph_from_job(job, max_edit_distance, status);
}

uint64_t hpo_graph_num_nodes(const hpo_graph* g) { return g->nodes.size(); }

size_t hpo_graph_node_alleles(const hpo_graph* g, uint64_t node, uint64_t* var_idx, uint8_t* allele, size_t cap) {
    auto it = g->node_to_alleles.find((size_t)node);
    if (it == g->node_to_alleles.end()) return 0;
    size_t n = 0;
    for (auto& pr : it->second) { if (n < cap) { var_idx[n] = pr.first; allele[n] = pr.second; } ++n; }
    return n;
}
size_t hpo_graph_node_seq(const hpo_graph* g, uint64_t node, uint8_t* out, size_t cap) {
    auto& s = g->nodes[node].sequence;
    std::memcpy(out, s.data(), std::min(cap, s.size()));
    return s.size();
}
size_t hpo_graph_node_parents(const hpo_graph* g, uint64_t node, uint64_t* out, size_t cap) {
    auto& p = g->nodes[node].parent_nodes;
    for (size_t i = 0; i < p.size() && i < cap; ++i) out[i] = p[i];
    return p.size();
}
size_t hpo_graph_node_edges(const hpo_graph* g, uint64_t node, uint64_t* out, size_t cap) {
    auto& e = g->edges[node];
    for (size_t i = 0; i < e.size() && i < cap; ++i) out[i] = e[i];
    return e.size();
}

int hpo_graph_edit_distance(const hpo_graph* g, const uint8_t* other, size_t other_len, uint64_t prune_distance,
                            uint64_t shuffle_seed, uint64_t* score, uint64_t* traversed, size_t* n_traversed) {
    std::vector<size_t> tn;
    uint64_t sc = 0;
    int st = edit_distance_with_pruning(*g, other, other_len, prune_distance, shuffle_seed, &sc, &tn);
    *score = sc;
    size_t cap = n_traversed ? *n_traversed : 0;
    for (size_t i = 0; i < tn.size() && i < cap; ++i) traversed[i] = tn[i];
    if (n_traversed) *n_traversed = tn.size();
    return st;
}

// read_parsing.rs:769-800
int hpo_wfa_assign(const hp_wfa_job* job, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* alleles) {
    int status = HP_OK;
    hpo_graph* g = graph_from_job(job, max_ed, &status);
    if (!g) return status;
    std::vector<size_t> tn;
    uint64_t sc = 0;
    int st = edit_distance_with_pruning(*g, job->read, job->read_len, prune_distance, 0, &sc, &tn);
    out->n_nodes = (uint32_t)g->nodes.size();
    out->score = sc;
    out->status = st;
    for (uint32_t i = 0; i < job->n_hets; ++i) alleles[i] = HP_ALLELE_NOOVERLAP;
    if (st == HP_OK) {
        for (size_t node : tn) {
            auto it = g->node_to_alleles.find(node);
            if (it == g->node_to_alleles.end()) continue;
            for (auto& pr : it->second) {
                if (alleles[pr.first] == HP_ALLELE_NOOVERLAP) alleles[pr.first] = pr.second;
                else if (alleles[pr.first] != pr.second) alleles[pr.first] = HP_ALLELE_AMBIGUOUS;
            }
        }
    }
    delete g;
    return st < 0 ? st : HP_OK;
}

}  // extern "C"
