// hp_oracle_brute.cpp — TEST INFRASTRUCTURE ONLY. Independent evidence for the graph-WFA oracle (hp_oracle_wfa.cpp, the restatement
// of reference src/wfa_graph.rs:350-650): the definition the wavefront algorithm implements, computed the slow way. Enumerates every
// path from node 0 to the LAST node of a small graph (the reference ends an alignment on the last node only, wfa_graph.rs:576-629),
// spells each path's sequence, takes the plain Levenshtein distance (hpo_edit_distance, pinned to src/sequence_alignment.rs' own
// tests) to the query and returns the minimum - which `edit_distance_with_pruning` must equal with pruning off. Uses only the
// graph's public accessors: nothing of the wavefront code is shared.
#include "hp_oracle.h"

#include <vector>

namespace {
struct Walk {
    const hpo_graph* g;
    size_t last;
    const uint8_t* other; size_t other_len;
    std::vector<std::vector<uint8_t>> seq;
    std::vector<std::vector<uint64_t>> edges;
    std::vector<uint8_t> spelled;
    uint64_t best = UINT64_MAX, n_paths = 0, n_optimal = 0, union_mask = 0, wfa_mask = 0;
    bool inside = false;
    void go(size_t node, uint64_t mask) {
        const size_t keep = spelled.size();
        spelled.insert(spelled.end(), seq[node].begin(), seq[node].end());
        mask |= 1ull << node;
        if (node == last) {
            ++n_paths;
            const uint64_t d = hpo_edit_distance(spelled.data(), spelled.size(), other, other_len);
            if (d < best) { best = d; n_optimal = 0; union_mask = 0; inside = false; }
            if (d == best) { ++n_optimal; union_mask |= mask; if ((mask & ~wfa_mask) == 0) inside = true; }
        } else {
            for (uint64_t c : edges[node]) go((size_t)c, mask);
        }
        spelled.resize(keep);
    }
};
}  // namespace

/* out[0] = min over root -> last-node paths of Levenshtein(path, other); out[1] = paths; out[2] = optimal paths; out[3] = union of the
 * optimal paths' nodes (bit per node); out[4] = 1 when some optimal path lies inside `wfa_mask` (the traversed set the WFA returned).
 * Graphs of at most 64 nodes; returns -1 otherwise or when the last node cannot be reached. */
extern "C" int hpo_graph_bruteforce(const hpo_graph* g, const uint8_t* other, size_t other_len, uint64_t wfa_mask, uint64_t out[5]) {
    const size_t n = (size_t)hpo_graph_num_nodes(g);
    if (n == 0 || n > 64) return -1;
    Walk w;
    w.g = g; w.last = n - 1; w.other = other; w.other_len = other_len; w.wfa_mask = wfa_mask;
    w.seq.resize(n); w.edges.resize(n);
    for (size_t i = 0; i < n; ++i) {
        std::vector<uint8_t> buf(1 << 12);
        const size_t len = hpo_graph_node_seq(g, i, buf.data(), buf.size());
        w.seq[i].assign(buf.begin(), buf.begin() + (long)len);
        std::vector<uint64_t> e(64);
        const size_t ne = hpo_graph_node_edges(g, i, e.data(), e.size());
        w.edges[i].assign(e.begin(), e.begin() + (long)ne);
    }
    w.go(0, 0);
    if (w.n_paths == 0) return -1;
    out[0] = w.best; out[1] = w.n_paths; out[2] = w.n_optimal; out[3] = w.union_mask; out[4] = w.inside ? 1 : 0;
    return 0;
}
