// wfa2_model.cpp — TEST INFRASTRUCTURE ONLY (never linked into libhiphase_gpu.so).
//
// CPU model of the round-2 graph-WFA stage, used to pin its DESIGN against the oracle without a GPU:
//   * the device graph builder w2_build / w2_map_alleles (hiphase_amd/csrc/hp_wfa2_dev.h) compiled for the host;
//   * the compact wavefront formulation of hp_wfa2_kernel.hip, diagonal by diagonal instead of lane by lane:
//     per-round arenas over the hull of each live node, pull from the previous round (d+1: offset+1, d: offset+1,
//     d-1: offset), a same-round injection list, and - instead of the reference's max_wavefronts map
//     (wfa_graph.rs:360,464-470) - the set of CAPPED diagonals:
//         a wave on (node, d) is stale  <=>  (node, d) once held a wave with offset == cap(node, d) and this wave has
//         offset < cap(node, d),  cap = min(node length, read length - d).
//     Why that is the same test: a diagonal whose wave is interior with read left gets offset+1 on itself next round,
//     so it stays strictly ahead of its own record until it is pruned by min_progression (after which every later,
//     shorter wave on it is pruned too) or reaches its cap (node end or read end), where its record stays forever.
// Same capacity limits as the kernel's LDS layout (W2Cfg): exceeding them returns W2_ST_NEED_BIG.
#include <algorithm>
#include <cstring>
#include <set>
#include <vector>

#include "../../hiphase_amd/csrc/hp_wfa2_dev.h"
#include "../../include/hiphase_gpu.h"

using namespace hp;

uint64_t g_x[64];
int g_opt = 1;   // bit 0: commit by KEY below the smallest target a finished wave hands on (round 5, as the kernel does); 0: by node (round 4)   // step / lane / insert counters of the third-generation model (scripts/w3_explore.cpp prints them)

namespace {

uint64_t g_reason[8];   // which limit sent a job to NEED_BIG (diagnostics)
uint64_t g_peak[4];     // peak live nodes / slots / injections / rounds

struct Built {
    std::vector<W2Variant> vars;
    std::vector<uint8_t> pool;
    W2Job job{};
    std::vector<W2Node> nodes;
    std::vector<uint16_t> edges;
    std::vector<uint32_t> tags;
    W2Info info{};
};

void build_from_job(const hp_wfa_job* j, Built& b) {
    auto add = [&](const hp_wfa_variant& v) {
        W2Variant w{};
        w.position = v.position; w.ref_len = v.ref_len; w.flags = v.flags;
        if (v.flags & 2u) { w.a0_off = (uint32_t)b.pool.size(); w.a0_len = v.allele0_len; b.pool.insert(b.pool.end(), v.allele0, v.allele0 + v.allele0_len); }
        w.a1_off = (uint32_t)b.pool.size(); w.a1_len = v.allele1_len; b.pool.insert(b.pool.end(), v.allele1, v.allele1 + v.allele1_len);
        b.vars.push_back(w);
    };
    for (uint32_t i = 0; i < j->n_hets; ++i) add(j->hets[i]);
    for (uint32_t i = 0; i < j->n_homs; ++i) add(j->homs[i]);
    W2Job& J = b.job;
    J.ref_start = (int64_t)j->ref_start;
    J.ref_len = (uint32_t)(j->ref_end - j->ref_start);
    J.read_len = j->read_len;
    J.het_first = 0; J.n_hets = j->n_hets; J.hom_first = j->n_hets; J.n_homs = j->n_homs;
    const uint32_t V = j->n_hets + j->n_homs;
    J.node_cap = 5 * V + 2; J.edge_cap = 2 * J.node_cap; J.tag_cap = 2 * j->n_hets + 2;
    b.nodes.resize(J.node_cap); b.edges.resize(J.edge_cap); b.tags.resize(J.tag_cap);
    w2_build(J, b.vars.data(), b.nodes.data(), b.edges.data(), b.tags.data(), &b.info);
}

template <int W> struct Slot { uint32_t ek = 0; uint32_t set[W]; };

// One edit-distance round = work items (node, interval of diagonals), nodes in index order. The items of node n are
// the merged union of (a) the hulls of n's entries of the previous round grown by one diagonal a side and (b) the
// hulls of the diagonals on which a parent finished THIS round, shifted by the parent's length. Results go to the
// round's arena, one slot per diagonal of the item; an item's slots are published as one entry per cluster of
// non-empty diagonals (clusters are separated by at least three empty diagonals... see `commit`).
template <int W>
int model_wfa(const Built& b, const uint8_t* ref, const uint8_t* read, uint64_t prune, uint64_t max_ed, uint64_t* score, uint32_t* out_set) {
    using C = W2Cfg<W>;
    const uint32_t nn = b.info.n_nodes;
    if (nn > (uint32_t)C::MAXN || b.info.n_edges > (uint32_t)C::MAXE) { g_reason[0]++; return W2_ST_NEED_BIG; }
    const uint32_t other_len = b.job.read_len, last = nn - 1;
    struct Live { uint32_t node; int32_t lo; uint32_t off; int32_t vlo, vhi, flo, fhi; };   // live hull / finished hull (may be empty)
    std::vector<Live> live[2];
    std::vector<Slot<W>> arena[2];
    arena[0].resize(C::SLOTS); arena[1].resize(C::SLOTS);
    struct Pair { uint32_t child, li; };
    std::vector<Pair> pairs;
    std::vector<uint8_t> pend(nn, 0);
    std::set<std::pair<uint32_t, int32_t>> capped;
    for (int w = 0; w < W; ++w) out_set[w] = 0;
    pend[0] = 1;   // start wave: node 0, diagonal 0, offset 0, set {0} (wfa_graph.rs:366-378) = a virtual injection in round 0
    uint64_t farthest = 0, min_prog = 0;
    for (uint32_t ed = 0;; ++ed) {
        const uint32_t c = ed & 1u, p = c ^ 1u;
        live[c].clear();
        pairs.clear();
        uint32_t top = 0;
        size_t pp = 0, n_live_entries = 0, n_fin_entries = 0;
        bool final_found = false;
        uint64_t round_far = 0;
        for (;;) {
            while (pp < live[p].size() && live[p][pp].vlo > live[p][pp].vhi) ++pp;   // entries that only held finished waves
            uint32_t a = pp < live[p].size() ? live[p][pp].node : 0xFFFFu, bq = 0xFFFFu;
            for (uint32_t n = 0; n < nn; ++n) if (pend[n]) { bq = n; break; }
            const uint32_t n = std::min(a, bq);
            if (n == 0xFFFFu) break;
            pend[n] = 0;
            const W2Node nd = b.nodes[n];
            const uint32_t len = nd.len_ref & ~W2_IS_REF;
            const uint8_t* nseq = (nd.len_ref & W2_IS_REF) ? ref + nd.seq_off : b.pool.data() + nd.seq_off;
            const uint32_t n_child = nd.child & 0xFFFFu;
            uint32_t scan = nd.child >> 16;   // first overflow entry (children 2 and later)
            // ---- sources ----
            const size_t p_first = pp;
            std::vector<std::pair<int32_t, int32_t>> src;
            while (pp < live[p].size() && live[p][pp].node == n) {
                if (live[p][pp].vlo <= live[p][pp].vhi) src.push_back({live[p][pp].vlo - 1, live[p][pp].vhi + 1});
                ++pp;
            }
            const size_t p_last = pp;
            for (auto& pr : pairs) if (pr.child == n) {
                const Live& P = live[c][pr.li];
                const int32_t pl = (int32_t)(b.nodes[P.node].len_ref & ~W2_IS_REF);
                src.push_back({P.flo + pl, P.fhi + pl});
            }
            if (ed == 0 && n == 0) src.push_back({0, 0});
            {   // the kernel keeps two previous entries of a node in registers and C::MAXPAR parent entries per node
                size_t nprev = 0, npar = 0;
                for (size_t k = p_first; k < p_last; ++k) nprev += live[p][k].vlo <= live[p][k].vhi;
                for (auto& pr : pairs) npar += pr.child == n;
                if (nprev > 2 || npar > (size_t)C::MAXPAR) { g_reason[4]++; return W2_ST_NEED_BIG; }
            }
            // one item over the whole span unless the sources are far apart; then merge overlapping / touching intervals
            std::sort(src.begin(), src.end());
            std::vector<std::pair<int32_t, int32_t>> items;
            {
                int32_t slo = INT32_MAX, shi = INT32_MIN;
                for (auto& iv : src) { slo = std::min(slo, iv.first); shi = std::max(shi, iv.second); }
                if (shi - slo < 2 * 8) items.push_back({slo, shi});
                else
                    for (auto& iv : src) {
                        if (!items.empty() && iv.first <= items.back().second + 1) items.back().second = std::max(items.back().second, iv.second);
                        else items.push_back(iv);
                    }
            }
            bool any_finished_node = false;
            for (auto& it : items) {
                const int32_t lo = it.first, hi = it.second;
                const uint32_t cnt = (uint32_t)(hi - lo + 1);
                if (top + cnt > (uint32_t)C::SLOTS) { g_reason[2]++; return W2_ST_NEED_BIG; }
                const uint32_t coff = top;
                top += cnt;
                for (int32_t d = lo; d <= hi; ++d) {
                    if (d <= -W2_DIAG_LIM || d >= W2_DIAG_LIM) { g_reason[3]++; return W2_ST_NEED_BIG; }
                    int64_t oA = -1, oB = -1, oC = -1;
                    const uint32_t *qA = nullptr, *qB = nullptr, *qC = nullptr;
                    auto at = [&](int32_t dd) -> const Slot<W>* {
                        for (size_t k = p_first; k < p_last; ++k) {
                            const Live& P = live[p][k];
                            if (dd >= P.vlo && dd <= P.vhi) return &arena[p][P.off + (uint32_t)(dd - P.lo)];
                        }
                        return nullptr;
                    };
                    if (auto s = at(d + 1)) { if (s->ek & 1u) { oA = (int64_t)(s->ek >> 3) + 1; qA = s->set; } }
                    if (auto s = at(d)) { if ((s->ek & 7u) == W2_KIND_INTERIOR_READ) { oB = (int64_t)(s->ek >> 3) + 1; qB = s->set; } }
                    if (auto s = at(d - 1)) { const uint32_t k = s->ek & 7u; if (k == W2_KIND_INTERIOR_READ || k == W2_KIND_END_LAST) { oC = (int64_t)(s->ek >> 3); qC = s->set; } }
                    uint32_t qD[W]; bool hinj = false;
                    for (int w = 0; w < W; ++w) qD[w] = 0;
                    for (auto& pr : pairs) if (pr.child == n) {
                        const Live& P = live[c][pr.li];
                        const int32_t dd = d - (int32_t)(b.nodes[P.node].len_ref & ~W2_IS_REF);
                        if (dd >= P.flo && dd <= P.fhi) {
                            const Slot<W>& s = arena[c][P.off + (uint32_t)(dd - P.lo)];
                            if ((s.ek & 7u) == W2_KIND_FINISHED) { hinj = true; for (int w = 0; w < W; ++w) qD[w] |= s.set[w]; }
                        }
                    }
                    if (ed == 0 && n == 0 && d == 0) hinj = true;
                    if (hinj) qD[n >> 5] |= 1u << (n & 31u);   // the successor's set = best + the successor (wfa_graph.rs:535-541)
                    const bool has = oA >= 0 || oB >= 0 || oC >= 0 || hinj;
                    Slot<W>& out = arena[c][coff + (uint32_t)(d - lo)];
                    out.ek = 0;
                    for (int w = 0; w < W; ++w) out.set[w] = 0;
                    if (!has) continue;
                    int64_t omax = std::max(std::max(oA, oB), std::max(oC, hinj ? (int64_t)0 : (int64_t)-1));
                    auto extend = [&](int64_t o) -> int64_t {   // wfa_graph.rs:454-459
                        int64_t pos = (int64_t)d + o;
                        while (o < (int64_t)len && pos >= 0 && pos < (int64_t)other_len && nseq[o] == read[pos]) { ++o; ++pos; }
                        return o;
                    };
                    const int64_t E = extend(omax);
                    auto ties = [&](int64_t o) -> bool {   // a candidate behind the furthest ties iff it extends to the same offset
                        if (o < 0) return false;
                        if (o == omax) return true;
                        return extend(o) == E;
                    };
                    const bool tA = ties(oA), tB = ties(oB), tC = ties(oC), tD = hinj && ties(0);
                    const int64_t pos_end = (int64_t)d + E;
                    const int64_t cap = std::min<int64_t>((int64_t)len, (int64_t)other_len - (int64_t)d);
                    const bool is_capped = capped.count({n, d}) != 0;
                    uint32_t best[W];
                    for (int w = 0; w < W; ++w) best[w] = (tA ? qA[w] : 0u) | (tB ? qB[w] : 0u) | (tC ? qC[w] : 0u) | (tD ? qD[w] : 0u);
                    // finals are collected from ALL waves of the last node, skipped diagonals included (wfa_graph.rs:576-588)
                    if (n == last && E == (int64_t)len && pos_end == (int64_t)other_len) {
                        final_found = true;
                        for (int w = 0; w < W; ++w) out_set[w] |= best[w];
                    }
                    const bool skip = (is_capped && E < cap) || (pos_end < (int64_t)min_prog);   // wfa_graph.rs:465
                    uint32_t kind = W2_KIND_NONE;
                    if (!skip) {
                        if ((uint64_t)pos_end > round_far) round_far = (uint64_t)pos_end;
                        if (E == cap) capped.insert({n, d});
                        if (E == (int64_t)len) {
                            if (n == last) { if (pos_end < (int64_t)other_len) kind = W2_KIND_END_LAST; }
                            else kind = W2_KIND_FINISHED;
                        } else kind = (pos_end < (int64_t)other_len) ? W2_KIND_INTERIOR_READ : W2_KIND_INTERIOR;
                    }
                    out.ek = ((uint32_t)E << 3) | kind;
                    for (int w = 0; w < W; ++w) out.set[w] = best[w];
                }
                // commit: one entry per cluster of non-empty diagonals; two non-empty diagonals share a cluster unless
                // at least two empty ones lie between them (then their grown hulls cannot even touch next round)
                int32_t clo = 0, chi = INT32_MIN;
                auto emit = [&]() -> bool {
                    if (chi == INT32_MIN) return true;
                    Live L{n, clo, coff + (uint32_t)(clo - lo), INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN};
                    for (int32_t d = clo; d <= chi; ++d) {
                        const uint32_t k = arena[c][coff + (uint32_t)(d - lo)].ek & 7u;
                        if (k == W2_KIND_FINISHED) { L.flo = std::min(L.flo, d); L.fhi = std::max(L.fhi, d); }
                        else if (k != W2_KIND_NONE) { L.vlo = std::min(L.vlo, d); L.vhi = std::max(L.vhi, d); }
                    }
                    if (L.vlo <= L.vhi) { if (++n_live_entries > (size_t)C::MAXL) { g_reason[1]++; return false; } }
                    else if (++n_fin_entries > (size_t)C::MAXF) { g_reason[1]++; return false; }
                    if (L.flo <= L.fhi) {
                        any_finished_node = true;
                        uint32_t sc = scan;
                        for (uint32_t j = 0; j < n_child; ++j) {
                            const uint32_t cid = j == 0 ? (nd.c01 & 0xFFFFu) : (j == 1 ? (nd.c01 >> 16) : w2_next_child(b.edges.data(), n, sc));
                            pairs.push_back(Pair{cid, (uint32_t)live[c].size()});
                            pend[cid] = 1;
                            size_t waiting = 0;
                            for (uint32_t q = 0; q < nn; ++q) waiting += pend[q];
                            if (waiting > (size_t)C::MAXQ) { g_reason[4]++; return false; }
                        }
                    }
                    live[c].push_back(L);
                    return true;
                };
                for (int32_t d = lo; d <= hi; ++d) {
                    if ((arena[c][coff + (uint32_t)(d - lo)].ek & 7u) == W2_KIND_NONE) continue;
                    if (chi != INT32_MIN && (d - chi >= 3 || d - clo >= C::MAXW)) { if (!emit()) return W2_ST_NEED_BIG; chi = INT32_MIN; }
                    if (chi == INT32_MIN) clo = d;
                    chi = d;
                }
                if (!emit()) return W2_ST_NEED_BIG;
            }
            (void)any_finished_node;
            g_peak[0] = std::max<uint64_t>(g_peak[0], live[c].size()); g_peak[1] = std::max<uint64_t>(g_peak[1], top); g_peak[2] = std::max<uint64_t>(g_peak[2], pairs.size());
        }
        if (final_found) { *score = ed; return W2_ST_OK; }
        if (round_far > farthest) farthest = round_far;
        if (farthest > prune) min_prog = farthest - prune;
        if ((uint64_t)ed + 1 > max_ed) { *score = max_ed; return W2_ST_MAX_ED; }
        bool any_live = false;
        for (auto& L : live[c]) any_live |= L.vlo <= L.vhi;
        if (!any_live) return W2_ST_INTERNAL;
    }
}

// ---- third generation (hp_wfa3_kernel.hip): flat sorted slot lists, tiles of G targets over any nodes ------------------------------
// Mirrors the kernel tile by tile: target list built from the previous round's live list, a tile = the next G targets, the commit
// rule (only slots of nodes below the first child of a node that finished in the tile), finished waves inserted into / merged with
// the sorted target list, live slots appended in key order, the finished waves' sets from the top of the round's set arena.
template <int W, int G>
int model_wfa3(const Built& b, const uint8_t* ref, const uint8_t* read, uint64_t prune, uint64_t max_ed, uint64_t* score, uint32_t* out_set) {
    using C = W3Cfg<W>;
    const uint32_t nn = b.info.n_nodes;
    if (nn > (uint32_t)C::MAXN || b.info.n_edges > (uint32_t)C::MAXE || b.job.read_len >= (uint32_t)W2_DIAG_LIM) { g_reason[0]++; return W2_ST_NEED_BIG; }
    const uint32_t other_len = b.job.read_len, last = nn - 1;
    struct E2 { uint32_t x, y; };
    std::vector<E2> A[2];
    A[0].assign(C::SLOTS, E2{0, 0}); A[1].assign(C::SLOTS, E2{0, 0});
    std::vector<uint32_t> sets[2];
    sets[0].assign((size_t)C::SLOTS * W, 0); sets[1].assign((size_t)C::SLOTS * W, 0);
    std::set<std::pair<uint32_t, int32_t>> capped;
    for (int w = 0; w < W; ++w) out_set[w] = 0;
    uint64_t farthest = 0, min_prog = 0;
    uint32_t nl_prev = 0;
    for (uint32_t ed = 0;; ++ed) {
        const uint32_t c = ed & 1u, p = c ^ 1u;
        uint32_t ip = 0, np = 0, nl = 0, nf = 0;
        // ---- the round's targets ----
        if (ed == 0) { A[c][0] = E2{w3_key(0, 0), w3_aux(W3_NONE, W3_NONE, W3_NONE) | W3_START}; np = 1; }
        else {
            uint32_t prevkey = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < nl_prev; ++i) {
                const uint32_t key = A[p][i].x;
                const bool same = prevkey != 0xFFFFFFFFu && w3_key_node(prevkey) == w3_key_node(key);
                const int32_t gap = same ? w3_key_diag(key) - w3_key_diag(prevkey) : 1 << 30;
                const int cnt = gap == 1 ? 1 : (gap == 2 ? 2 : 3);
                const uint32_t n = w3_key_node(key);
                const int32_t d = w3_key_diag(key);
                for (int t = 2 - cnt; t < 2; ++t) {   // targets d - 1 (cnt 3), d (cnt >= 2), d + 1
                    const int32_t td = d - 1 + t + (cnt == 3 ? 0 : 0);
                    (void)td;
                }
                for (int j = 0; j < cnt; ++j) {
                    const int32_t td = d + 1 - (cnt - 1) + j;   // cnt 3: d-1, d, d+1; cnt 2: d, d+1; cnt 1: d+1
                    if (td <= -W2_DIAG_LIM || td >= W2_DIAG_LIM) { g_reason[3]++; return W2_ST_NEED_BIG; }
                    if (np >= (uint32_t)C::SLOTS) { g_reason[2]++; return W2_ST_NEED_BIG; }
                    A[c][np++] = E2{w3_key(n, td), w3_aux(i, W3_NONE, W3_NONE)};
                }
                prevkey = key;
            }
        }
        bool final_found = false;
        uint64_t round_far = 0;
        uint32_t steps = 0;
        g_x[0]++; g_x[13] += np;
        while (ip < np) {
            if (++steps > 100000) return W2_ST_INTERNAL;
            const uint32_t tn = std::min<uint32_t>((uint32_t)G, np - ip);
            g_x[1]++; g_x[2] += tn; if (steps == 1) g_x[12]++;
            struct Res { uint32_t n; int32_t d; bool has; uint32_t E, kind; bool is_final, ins; uint32_t best[W]; uint64_t pos_end; uint32_t len; int64_t omax; bool inj; };
            std::vector<Res> R(tn);
            uint32_t X = 0xFFFFFFFFu;
            for (uint32_t l = 0; l < tn; ++l) {
                const E2 tgt = A[c][ip + l];
                Res& r = R[l];
                const uint32_t n = w3_key_node(tgt.x);
                const int32_t d = w3_key_diag(tgt.x);
                r.n = n; r.d = d;
                const W2Node nd = b.nodes[n];
                const uint32_t len = nd.len_ref & ~W2_IS_REF;
                r.len = len;
                const uint8_t* nseq = (nd.len_ref & W2_IS_REF) ? ref + nd.seq_off : b.pool.data() + nd.seq_off;
                int64_t oA = -1, oB = -1, oC = -1;
                const uint32_t *qA = nullptr, *qB = nullptr, *qC = nullptr;
                const uint32_t back = tgt.y & 0x3FFu, src0 = (tgt.y >> 10) & 0x3FFu, src1 = (tgt.y >> 20) & 0x3FFu;
                if (back != W3_NONE)
                    for (uint32_t k = back; k < back + 3 && k < nl_prev; ++k) {
                        const E2 e = A[p][k];
                        if (w3_key_node(e.x) != n) continue;
                        const int32_t dd = w3_key_diag(e.x);
                        const uint32_t kd = e.y & 7u;
                        const uint32_t* q = &sets[p][(size_t)k * W];
                        if (dd == d + 1) { if (kd & 1u) { oA = (int64_t)(e.y >> 3) + 1; qA = q; } }
                        else if (dd == d) { if (kd == W2_KIND_INTERIOR_READ) { oB = (int64_t)(e.y >> 3) + 1; qB = q; } }
                        else if (dd == d - 1) { if (kd == W2_KIND_INTERIOR_READ || kd == W2_KIND_END_LAST) { oC = (int64_t)(e.y >> 3); qC = q; } }
                    }
                uint32_t qD[W];
                for (int w = 0; w < W; ++w) qD[w] = 0;
                bool hinj = (tgt.y & W3_START) != 0;
                if (src0 != W3_NONE) { hinj = true; for (int w = 0; w < W; ++w) qD[w] |= sets[c][(size_t)src0 * W + w]; }
                if (src1 != W3_NONE) { hinj = true; for (int w = 0; w < W; ++w) qD[w] |= sets[c][(size_t)src1 * W + w]; }
                if (hinj) qD[n >> 5] |= 1u << (n & 31u);
                r.has = oA >= 0 || oB >= 0 || oC >= 0 || hinj;
                r.kind = W2_KIND_NONE; r.E = 0; r.is_final = false; r.ins = false; r.pos_end = 0;
                for (int w = 0; w < W; ++w) r.best[w] = 0;
                if (!r.has) continue;
                const int64_t omax = std::max(std::max(oA, oB), std::max(oC, hinj ? (int64_t)0 : (int64_t)-1));
                auto extend = [&](int64_t o) -> int64_t {
                    int64_t pos = (int64_t)d + o;
                    while (o < (int64_t)len && pos >= 0 && pos < (int64_t)other_len && nseq[o] == read[pos]) { ++o; ++pos; }
                    return o;
                };
                const int64_t E = extend(omax);
                r.omax = omax; r.inj = hinj;
                auto ties = [&](int64_t o) -> bool { if (o < 0) return false; if (o == omax) return true; return extend(o) == E; };
                const bool tA = ties(oA), tB = ties(oB), tC = ties(oC), tD = hinj && ties(0);
                const int64_t pos_end = (int64_t)d + E;
                const int64_t cap = std::min<int64_t>((int64_t)len, (int64_t)other_len - (int64_t)d);
                const bool is_capped = capped.count({n, d}) != 0;
                for (int w = 0; w < W; ++w) r.best[w] = (tA ? qA[w] : 0u) | (tB ? qB[w] : 0u) | (tC ? qC[w] : 0u) | (tD ? qD[w] : 0u);
                r.is_final = n == last && E == (int64_t)len && pos_end == (int64_t)other_len;
                const bool skip = (is_capped && E < cap) || (pos_end < (int64_t)min_prog);
                if (!skip) {
                    r.ins = E == cap && !is_capped;
                    if (E == (int64_t)len) {
                        if (n == last) { if (pos_end < (int64_t)other_len) r.kind = W2_KIND_END_LAST; }
                        else r.kind = W2_KIND_FINISHED;
                    } else r.kind = (pos_end < (int64_t)other_len) ? W2_KIND_INTERIOR_READ : W2_KIND_INTERIOR;
                    r.pos_end = (uint64_t)pos_end;
                }
                r.E = (uint32_t)E;
                if (r.kind == W2_KIND_FINISHED) X = std::min(X, (g_opt & 1) ? w3_key(nd.c01 & 0xFFFFu, d + (int32_t)len) : ((nd.c01 & 0xFFFFu) << 19));   // its first child (ids ascend with creation)
            }
            // ---- commit: the slots of nodes below X (a prefix of the tile: targets are sorted) ----
            uint32_t ncommit = 0;
            while (ncommit < tn && w3_key(R[ncommit].n, R[ncommit].d) < X) ++ncommit;
            if (ncommit == 0) return W2_ST_INTERNAL;
            const uint32_t ip_next = ip + ncommit;
            g_x[4] += ncommit; g_x[14] += tn - ncommit; if (X != 0xFFFFFFFFu) g_x[11]++;
            for (uint32_t l = 0; l < tn; ++l) g_x[3] += R[l].has ? 1 : 0;
            for (uint32_t l = 0; l < ncommit; ++l) {
                const Res& r = R[l];
                if (!r.has) continue;
                if (r.is_final) { final_found = true; for (int w = 0; w < W; ++w) out_set[w] |= r.best[w]; }
                if (r.kind == W2_KIND_NONE) continue;
                if (r.pos_end > round_far) round_far = r.pos_end;
                if (r.ins) capped.insert({r.n, r.d});
                if (r.kind == W2_KIND_FINISHED) {
                    if (nl + nf + 1 > (uint32_t)C::SLOTS) { g_reason[2]++; return W2_ST_NEED_BIG; }
                    const uint32_t si = (uint32_t)C::SLOTS - 1u - nf;
                    ++nf;
                    g_x[5]++; if (r.omax == (int64_t)r.len) g_x[16]++; else if (r.E - r.omax <= 2) g_x[17]++; if (r.len <= 2) g_x[18]++; if (r.inj && r.omax == 0) g_x[19]++;
                    for (int w = 0; w < W; ++w) sets[c][(size_t)si * W + w] = r.best[w];
                    const W2Node nd = b.nodes[r.n];
                    const uint32_t n_child = nd.child & 0xFFFFu;
                    uint32_t scan = nd.child >> 16;
                    for (uint32_t j = 0; j < n_child; ++j) {
                        const uint32_t cid = j == 0 ? (nd.c01 & 0xFFFFu) : (j == 1 ? (nd.c01 >> 16) : w2_next_child(b.edges.data(), r.n, scan));
                        const int32_t td = r.d + (int32_t)r.len;
                        if (td <= -W2_DIAG_LIM || td >= W2_DIAG_LIM) { g_reason[3]++; return W2_ST_NEED_BIG; }
                        const uint32_t key = w3_key(cid, td);
                        uint32_t pos = ip_next;
                        while (pos < np && A[c][pos].x < key) ++pos;
                        g_x[6]++;
                        if (pos == np) g_x[7]++; else if (A[c][pos].x == key) g_x[8]++; else if (np - ip_next <= (uint32_t)G) g_x[9]++; else g_x[10]++;
                        if (r.omax == (int64_t)r.len) { if (pos == np) g_x[20]++; else if (A[c][pos].x == key) g_x[21]++; else g_x[22]++; }
                        if (pos != np && pos - ip_next >= (uint32_t)G) g_x[23]++;   // neither appended nor within the first G targets still to come
                        if (pos != np && pos - ip_next < (uint32_t)G && A[c][pos].x != key && ip_next - nl < 1) g_x[24]++;
                        if (pos < np && A[c][pos].x == key) {
                            uint32_t& y = A[c][pos].y;
                            if (((y >> 10) & 0x3FFu) == W3_NONE) y = (y & ~(0x3FFu << 10)) | (si << 10);
                            else if (((y >> 20) & 0x3FFu) == W3_NONE) y = (y & ~(0x3FFu << 20)) | (si << 20);
                            else {   // a third wave onto one target (rare): its set and the second one's are merged into a fresh arena entry
                                g_x[15]++;
                                if (nl + nf + 1 > (uint32_t)C::SLOTS) { g_reason[2]++; return W2_ST_NEED_BIG; }
                                const uint32_t sm = (uint32_t)C::SLOTS - 1u - nf;
                                ++nf;
                                const uint32_t s1 = (y >> 20) & 0x3FFu;
                                for (int w = 0; w < W; ++w) sets[c][(size_t)sm * W + w] = sets[c][(size_t)s1 * W + w] | sets[c][(size_t)si * W + w];
                                y = (y & ~(0x3FFu << 20)) | (sm << 20);
                            }
                        } else {
                            if (np >= (uint32_t)C::SLOTS) { g_reason[2]++; return W2_ST_NEED_BIG; }
                            for (uint32_t k = np; k > pos; --k) A[c][k] = A[c][k - 1];
                            A[c][pos] = E2{key, w3_aux(W3_NONE, si, W3_NONE)};
                            ++np;
                        }
                    }
                } else {
                    if (nl + nf + 1 > (uint32_t)C::SLOTS || nl >= ip_next) { g_reason[2]++; return W2_ST_NEED_BIG; }
                    for (int w = 0; w < W; ++w) sets[c][(size_t)nl * W + w] = r.best[w];
                    A[c][nl] = E2{w3_key(r.n, r.d), (r.E << 3) | r.kind};
                    ++nl;
                }
            }
            ip = ip_next;
            g_peak[1] = std::max<uint64_t>(g_peak[1], np);
        }
        g_peak[3] += steps;
        if (final_found) { *score = ed; return W2_ST_OK; }
        if (round_far > farthest) farthest = round_far;
        if (farthest > prune) min_prog = farthest - prune;
        if ((uint64_t)ed + 1 > max_ed) { *score = max_ed; return W2_ST_MAX_ED; }
        if (nl == 0) return W2_ST_INTERNAL;
        nl_prev = nl;
    }
}

}  // namespace

extern "C" {

const uint64_t* w2m_reasons() { return g_reason; }
const uint64_t* w2m_peaks() { return g_peak; }

// Same contract as the oracle's hpo_wfa_assign; *path: 0 compact model, 1 = NEED_BIG (limits exceeded, nothing
// written), 2 = builder asked for the host path. W = set words of the class the job falls in (2, 4, 8).
int w2m_wfa_assign(const hp_wfa_job* job, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* alleles, int* path) {
    Built b;
    build_from_job(job, b);
    *path = 0;
    if (b.info.status == W2B_NEED_HOST) { *path = 2; return 0; }
    if (b.info.status != W2B_OK) return HP_ERR_INVARIANT;
    const uint8_t* ref = job->reference + (job->ref_start - job->ref_base);
    uint64_t score = 0;
    uint32_t set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int st;
    if (b.info.n_nodes <= 64) st = model_wfa<2>(b, ref, job->read, prune_distance, max_ed, &score, set);
    else if (b.info.n_nodes <= 128) st = model_wfa<4>(b, ref, job->read, prune_distance, max_ed, &score, set);
    else if (b.info.n_nodes <= 256) st = model_wfa<8>(b, ref, job->read, prune_distance, max_ed, &score, set);
    else { st = W2_ST_NEED_BIG; g_reason[5]++; }
    if (st == W2_ST_NEED_BIG) { *path = 1; return 0; }
    if (st != W2_ST_OK && st != W2_ST_MAX_ED) return HP_ERR_INVARIANT;
    out->status = st == W2_ST_OK ? HP_OK : HP_WFA_MAX_ED;
    out->n_nodes = b.info.n_nodes;
    out->score = score;
    w2_map_alleles(b.tags.data(), b.info.n_tags, set, st == W2_ST_OK, alleles, job->n_hets);
    return 0;
}

// the third-generation formulation (flat sorted slot lists), same contract
int w3m_wfa_assign(const hp_wfa_job* job, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* alleles, int* path) {
    Built b;
    build_from_job(job, b);
    *path = 0;
    if (b.info.status == W2B_NEED_HOST) { *path = 2; return 0; }
    if (b.info.status != W2B_OK) return HP_ERR_INVARIANT;
    const uint8_t* ref = job->reference + (job->ref_start - job->ref_base);
    uint64_t score = 0;
    uint32_t set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int st;
    if (b.info.n_nodes <= 64) st = model_wfa3<2, 8>(b, ref, job->read, prune_distance, max_ed, &score, set);
    else if (b.info.n_nodes <= 128) st = model_wfa3<4, 8>(b, ref, job->read, prune_distance, max_ed, &score, set);
    else if (b.info.n_nodes <= 256) st = model_wfa3<8, 16>(b, ref, job->read, prune_distance, max_ed, &score, set);
    else { st = W2_ST_NEED_BIG; g_reason[5]++; }
    if (st == W2_ST_NEED_BIG) { *path = 1; return 0; }
    if (st != W2_ST_OK && st != W2_ST_MAX_ED) return HP_ERR_INVARIANT;
    out->status = st == W2_ST_OK ? HP_OK : HP_WFA_MAX_ED;
    out->n_nodes = b.info.n_nodes;
    out->score = score;
    w2_map_alleles(b.tags.data(), b.info.n_tags, set, st == W2_ST_OK, alleles, job->n_hets);
    return 0;
}

}
