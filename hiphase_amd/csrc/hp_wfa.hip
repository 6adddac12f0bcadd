// hp_wfa.hip — host side of the graph-WFA allele assignment: builds each read's variant graph
// (reference src/wfa_graph.rs:119-284 `from_reference_variants_with_hom`, restated over sequence SPANS
// instead of copies), lays out the per-node diagonal bands, launches hp_wfa_kernel and maps the
// traversed nodes to per-het AlleleTypes (reference src/read_parsing.rs:790-800).
//
// Boundary: hp_wfa_assign_batch replaces the two calls at reference src/read_parsing.rs:769-780.
#include "hp_wfa_kernel.hip"
#include "hp_combine.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <numeric>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace hp { extern thread_local const uint32_t* g_wfa_min_ed_hint; }   // hp_wfa2_host.h: set while a block set's leftovers take their dense-band pass
namespace hp {
namespace {

struct HostNode {          // POD: parent / child lists live in the job's flat arrays (few allocations per read)
    uint32_t seq_off, seq_len;
    uint32_t par_off, n_par;      // into HostJob::par (sorted, wfa_graph.rs:36)
    uint32_t child_off, n_child;  // into HostJob::child (creation order, wfa_graph.rs:321-323); filled by finish_graph
    int64_t emin, emax;           // shortest / longest path length from the root to this node's first base
};

// graph under construction: a per-thread scratch whose vectors keep their capacity from one read to the next
struct GraphBuild {
    std::vector<HostNode> nodes;
    std::vector<uint32_t> par, child;
    // sequence offsets are "virtual": [ref slice][alt allele bytes][read][pad]. On the device the reference slice lives
    // in the shared merged ranges and [alt][read][pad] are the job's private bytes (pack_jobs); only the (small)
    // allele bytes are copied here, reference and read go straight from the caller into the upload staging buffer
    const uint8_t* ref_ptr = nullptr;
    uint32_t ref_len = 0;
    std::vector<uint8_t> alt;
    const uint8_t* read_ptr = nullptr;
    uint32_t read_off = 0, read_len = 0, seq_bytes = 0;
    // node_to_alleles (wfa_graph.rs:19): (node, het index, allele)
    std::vector<std::array<uint32_t, 3>> tags;
    void clear() { nodes.clear(); par.clear(); child.clear(); alt.clear(); tags.clear(); ref_ptr = read_ptr = nullptr; ref_len = read_off = read_len = seq_bytes = 0; }
};
// The finished graphs of all reads a worker thread built live back to back in that thread's arena (five
// allocations per thread instead of five per read; freed in one go), a HostJob is a set of views into it.
template <class T> struct View {
    const T* p = nullptr;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const T* data() const { return p; }
    const T* begin() const { return p; }
    const T* end() const { return p + n; }
    const T& operator[](size_t i) const { return p[i]; }
};
struct GraphArena {
    std::vector<HostNode> nodes;
    std::vector<uint32_t> par, child;
    std::vector<uint8_t> alt;
    std::vector<std::array<uint32_t, 3>> tags;
};
struct HostJob {
    View<HostNode> nodes;
    View<uint32_t> par, child;
    View<uint8_t> alt;
    View<std::array<uint32_t, 3>> tags;
    const uint8_t* ref_ptr = nullptr;
    const uint8_t* read_ptr = nullptr;
    uint32_t ref_len = 0, read_off = 0, read_len = 0, seq_bytes = 0;
    size_t off[5] = {0, 0, 0, 0, 0};   // arena offsets until the views are bound (the arena may still grow)
    void stash(const GraphBuild& g, GraphArena& a) {
        off[0] = a.nodes.size(); off[1] = a.par.size(); off[2] = a.child.size(); off[3] = a.alt.size(); off[4] = a.tags.size();
        a.nodes.insert(a.nodes.end(), g.nodes.begin(), g.nodes.end());
        a.par.insert(a.par.end(), g.par.begin(), g.par.end());
        a.child.insert(a.child.end(), g.child.begin(), g.child.end());
        a.alt.insert(a.alt.end(), g.alt.begin(), g.alt.end());
        a.tags.insert(a.tags.end(), g.tags.begin(), g.tags.end());
        nodes.n = g.nodes.size(); par.n = g.par.size(); child.n = g.child.size(); alt.n = g.alt.size(); tags.n = g.tags.size();
        ref_ptr = g.ref_ptr; read_ptr = g.read_ptr; ref_len = g.ref_len; read_off = g.read_off; read_len = g.read_len; seq_bytes = g.seq_bytes;
    }
    void bind(const GraphArena& a) {
        nodes.p = a.nodes.data() + off[0]; par.p = a.par.data() + off[1]; child.p = a.child.data() + off[2];
        alt.p = a.alt.data() + off[3]; tags.p = a.tags.data() + off[4];
    }
};

// WFAGraph::add_node (wfa_graph.rs:298-331)
int add_node(GraphBuild& g, uint32_t seq_off, uint32_t seq_len, const std::vector<uint32_t>& parents) {
    const uint32_t idx = (uint32_t)g.nodes.size();
    if (idx == 0) { if (!parents.empty()) return -1; }
    else {
        if (parents.empty()) return -1;
        for (uint32_t p : parents) if (idx <= p) return -1;
    }
    HostNode n{};
    n.seq_off = seq_off;
    n.seq_len = seq_len;
    n.par_off = (uint32_t)g.par.size();
    n.n_par = (uint32_t)parents.size();
    g.par.insert(g.par.end(), parents.begin(), parents.end());
    std::sort(g.par.begin() + n.par_off, g.par.end());
    g.nodes.push_back(n);
    return (int)idx;
}

// children lists (edges[p].push(child) in creation order) + path-length ranges, once all nodes exist
void finish_graph(GraphBuild& g) {
    const size_t nn = g.nodes.size();
    std::vector<uint32_t> cnt(nn + 1, 0);
    for (size_t n = 0; n < nn; ++n)
        for (uint32_t k = 0; k < g.nodes[n].n_par; ++k) cnt[g.par[g.nodes[n].par_off + k] + 1]++;
    for (size_t n = 0; n < nn; ++n) { cnt[n + 1] += cnt[n]; g.nodes[n].child_off = cnt[n]; g.nodes[n].n_child = 0; }
    g.child.assign(g.par.size(), 0);
    for (size_t n = 0; n < nn; ++n)   // ascending child index == creation order
        for (uint32_t k = 0; k < g.nodes[n].n_par; ++k) {
            HostNode& p = g.nodes[g.par[g.nodes[n].par_off + k]];
            g.child[p.child_off + p.n_child++] = (uint32_t)n;
        }
    for (size_t n = 1; n < nn; ++n) {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (uint32_t k = 0; k < g.nodes[n].n_par; ++k) {
            const HostNode& q = g.nodes[g.par[g.nodes[n].par_off + k]];
            lo = std::min(lo, q.emin + (int64_t)q.seq_len);
            hi = std::max(hi, q.emax + (int64_t)q.seq_len);
        }
        g.nodes[n].emin = lo;
        g.nodes[n].emax = hi;
    }
}

// from_reference_variants_with_hom (wfa_graph.rs:119-284). Reference nodes are spans of the copied
// reference slice; allele nodes are spans of the appended allele bytes.
int build_graph(const hp_wfa_job* job, GraphBuild& g) {
    if (job->ref_end < job->ref_start || job->ref_start < job->ref_base) { set_error("bad reference window"); return HP_ERR_ARG; }
    const size_t ref_len = (size_t)(job->ref_end - job->ref_start);
    g.ref_ptr = job->reference + (job->ref_start - job->ref_base);
    g.ref_len = (uint32_t)ref_len;
    auto ref_span = [&](uint64_t a) { return (uint32_t)(a - job->ref_start); };
    const uint64_t ref_start = job->ref_start, ref_end = job->ref_end;
    uint64_t previous_end = ref_start;
    std::vector<uint32_t> reference_reconnect;
    std::vector<std::pair<uint32_t, uint32_t>> reference_alleles;  // (het index, 0)
    // (reconnect position, alt node), smallest first; a handful of entries at most, so a sorted vector (no node
    // allocations) stands in for the reference's priority queue; ties in any order
    struct ReconnectQueue {
        std::vector<std::pair<uint64_t, uint32_t>> v;
        bool empty() const { return v.empty(); }
        const std::pair<uint64_t, uint32_t>* begin() const { return v.data(); }
        void insert(std::pair<uint64_t, uint32_t> x) { v.insert(std::upper_bound(v.begin(), v.end(), x), x); }
        void erase(const std::pair<uint64_t, uint32_t>*) { v.erase(v.begin()); }
    } reconnect_queue;

    struct VarRef { const hp_wfa_variant* v; int64_t index; };
    std::vector<VarRef> all;
    all.reserve((size_t)job->n_hets + job->n_homs);
    for (uint32_t i = 0; i < job->n_hets; ++i) all.push_back({&job->hets[i], (int64_t)i});
    for (uint32_t i = 0; i < job->n_homs; ++i) all.push_back({&job->homs[i], -1});
    {   // stable by position, hets before homs on ties (wfa_graph.rs:137-144); both lists normally arrive sorted, then a
        // merge does it without stable_sort's temporary buffer
        auto by_pos = [](const VarRef& a, const VarRef& b) { return a.v->position < b.v->position; };
        const auto mid = all.begin() + job->n_hets;
        if (std::is_sorted(all.begin(), mid, by_pos) && std::is_sorted(mid, all.end(), by_pos)) {
            std::vector<VarRef> merged(all.size());
            std::merge(all.begin(), mid, mid, all.end(), merged.begin(), by_pos);
            all.swap(merged);
        } else std::stable_sort(all.begin(), all.end(), by_pos);
    }
    {   // capacities up front: at most 3 nodes per variant + 2, one or two parents each
        size_t alt_bytes = 0;
        for (auto& vr : all) alt_bytes += (size_t)vr.v->allele1_len + ((vr.v->flags & 2u) ? vr.v->allele0_len : 0u);
        g.nodes.reserve(3 * all.size() + 2);
        g.par.reserve(6 * all.size() + 4);
        g.tags.reserve(2 * (size_t)job->n_hets + 2);
        g.alt.reserve(alt_bytes);
        reference_reconnect.reserve(8);
        reconnect_queue.v.reserve(8);
    }

    auto flush_ref_alleles = [&](int node) {
        for (auto& ra : reference_alleles) g.tags.push_back({(uint32_t)node, ra.first, ra.second});
        reference_alleles.clear();
    };
    auto drain_one = [&]() -> bool {  // wfa_graph.rs:168-189 and :256-272
        auto it = reconnect_queue.begin();
        const uint64_t alt_reconnect = it->first;
        const uint32_t alt_index = it->second;
        reconnect_queue.erase(it);
        if (!(alt_reconnect > previous_end)) return false;  // assert!(alt_reconnect > previous_end)
        int ri = add_node(g, ref_span(previous_end), (uint32_t)(alt_reconnect - previous_end), reference_reconnect);
        if (ri < 0) return false;
        flush_ref_alleles(ri);
        previous_end = alt_reconnect;
        reference_reconnect = {(uint32_t)ri, alt_index};
        while (!reconnect_queue.empty() && reconnect_queue.begin()->first == alt_reconnect) {
            reference_reconnect.push_back(reconnect_queue.begin()->second);
            reconnect_queue.erase(reconnect_queue.begin());
        }
        return true;
    };
    auto add_allele = [&](const uint8_t* bytes, uint32_t len) -> int {
        const uint32_t off = g.ref_len + (uint32_t)g.alt.size();
        g.alt.insert(g.alt.end(), bytes, bytes + len);
        return add_node(g, off, len, reference_reconnect);
    };

    for (auto& vr : all) {
        const hp_wfa_variant* v = vr.v;
        if (v->flags & 1u) continue;                                     // is_ignored (wfa_graph.rs:147-150)
        if (v->position < (int64_t)ref_start) continue;                  // :155-159
        const uint64_t pos = (uint64_t)v->position;
        if (pos + v->ref_len > ref_end) continue;                        // :160-164
        while (!reconnect_queue.empty() && reconnect_queue.begin()->first <= pos)
            if (!drain_one()) { set_error("graph construction assert (alt_reconnect > previous_end)"); return HP_ERR_INVARIANT; }
        if (previous_end < pos || g.nodes.empty()) {                     // :196-209
            int ri = add_node(g, ref_span(previous_end), (uint32_t)(pos - previous_end), reference_reconnect);
            if (ri < 0) { set_error("graph construction: add_node failed"); return HP_ERR_INVARIANT; }
            flush_ref_alleles(ri);
            reference_reconnect = {(uint32_t)ri};
            previous_end = pos;
        } else if (previous_end != pos) {
            set_error("graph construction assert (previous_end == variant_pos)");
            return HP_ERR_INVARIANT;
        }
        if (v->flags & 2u) {                                             // allele0 is itself an ALT (:217-231)
            int ai = add_allele(v->allele0, v->allele0_len);
            if (ai < 0) { set_error("graph construction: add_node failed"); return HP_ERR_INVARIANT; }
            if (vr.index >= 0) g.tags.push_back({(uint32_t)ai, (uint32_t)vr.index, 0u});
            reconnect_queue.insert({pos + v->ref_len, (uint32_t)ai});
        } else if (vr.index >= 0) {
            reference_alleles.push_back({(uint32_t)vr.index, 0u});       // tags the NEXT reference node (:233-237)
        }
        int ai = add_allele(v->allele1, v->allele1_len);                  // :240-251
        if (ai < 0) { set_error("graph construction: add_node failed"); return HP_ERR_INVARIANT; }
        if (vr.index >= 0) g.tags.push_back({(uint32_t)ai, (uint32_t)vr.index, 1u});
        reconnect_queue.insert({pos + v->ref_len, (uint32_t)ai});
    }
    while (!reconnect_queue.empty())
        if (!drain_one()) { set_error("graph construction assert (alt_reconnect > previous_end)"); return HP_ERR_INVARIANT; }
    if (!(previous_end <= ref_end)) { set_error("graph construction assert (previous_end <= ref_end)"); return HP_ERR_INVARIANT; }
    if (add_node(g, ref_span(previous_end), (uint32_t)(ref_end - previous_end), reference_reconnect) < 0) {
        set_error("graph construction: add_node failed"); return HP_ERR_INVARIANT;
    }
    if (!reference_alleles.empty()) { set_error("graph construction assert (dangling reference alleles)"); return HP_ERR_INVARIANT; }

    g.read_off = g.ref_len + (uint32_t)g.alt.size();
    g.read_len = job->read_len;
    g.read_ptr = job->read;
    g.seq_bytes = (g.read_off + g.read_len + 16 + 15) & ~15u;  // 8-byte compares may read past the last base
    finish_graph(g);
    // the allele mapping walks the traversed nodes in ascending order (read_parsing.rs:790-800)
    // (tags are pushed when the node they name is created, so they normally are in node order already)
    auto by_node = [](const std::array<uint32_t, 3>& x, const std::array<uint32_t, 3>& y) { return x[0] < y[0]; };
    if (!std::is_sorted(g.tags.begin(), g.tags.end(), by_node)) std::stable_sort(g.tags.begin(), g.tags.end(), by_node);
    return HP_OK;
}


// Per-(thread, device) scratch that survives across calls: the kernel leaves its band scratch zeroed, so the
// (large) allocation + memset is paid once per worker thread, not once per block.
struct WfaContext {
    int device = -1;
    DevBuf scratch;
    bool dirty = true;     // needs a memset before use
};
thread_local WfaContext g_ctx;

double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// Reused upload staging for the sequence bytes: pinned host memory (DMA straight from it), grow-only, never
// value-initialised (a std::vector would memset ~35 KB per read on every call).
struct StageBuf {
    uint8_t* p = nullptr;
    size_t cap = 0, len = 0;
    ~StageBuf() { if (p) (void)hipHostFree(p); }
    int resize(size_t n) {
        if (n > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            // (pinning and unpinning host memory waits for the whole device - inside a block stream that is the other stages' persistent
            // kernels: a leftover pass whose tables outgrew its thread's staging by a few per cent stood still for 50-80 ms.
            // Hence the floor and the headroom: the passes of a stream's late results never grow it after their first.)
            // (the 64 MB floor only on a thread that runs a block set's leftovers - g_wfa_min_ed_hint: the late / early pass; every other
            // thread that comes here - a small set's helper per pipeline slot, a generic caller - keeps 8 MB: three such buffers per thread,
            // and a depth-7 stream on eight pipelines pinned 2.7 GB of host memory for nothing)
            const size_t want = std::max<size_t>(2 * n + 4096, (size_t)(g_wfa_min_ed_hint ? 64 : 8) << 20);   // (round 5: 64 MB, twice the need - a HiFi-shaped set leaves five hundred reads to this pass where the uniform workload leaves seventy-five, and how many varies from set to set: at 8 MB + half, a stream's tables grew three or four times, 100-370 ms of standing still each, some of them inside a timed region)
            g_device_syncing_allocs.fetch_add(1);
            if (hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault) != hipSuccess) {
                p = nullptr;
                set_error("hipHostMalloc(%zu) failed", want);
                return HP_ERR_OOM;
            }
            cap = want;
        }
        len = n;
        return HP_OK;
    }
    uint8_t* data() { return p; }
    size_t size() const { return len; }
};
template <class T> struct StageVec {   // typed view of a StageBuf (node / edge tables: tens of MB per call)
    StageBuf b;
    int resize(size_t n) { return b.resize(n * sizeof(T)); }
    T* data() { return reinterpret_cast<T*>(b.p); }
    size_t size() const { return b.len / sizeof(T); }
};
thread_local StageBuf g_seq_stage;
thread_local StageVec<WfaNode> g_node_stage;
thread_local StageVec<WfaEdge> g_edge_stage;

struct WfaPack {
    std::vector<WfaJobDesc> jobs;
    StageVec<WfaNode>& nodes = g_node_stage;
    StageVec<WfaEdge>& edges = g_edge_stage;
    StageBuf& seq = g_seq_stage;
    uint64_t out_set_words = 0;
    uint64_t max_scratch = 0;
    uint32_t max_nodes = 0, max_edges = 0;
};

unsigned wfa_host_threads(size_t n) {
    const char* tenv = std::getenv("HP_WFA_HOST_THREADS");
    unsigned nt = tenv ? (unsigned)std::atoi(tenv) : host_threads(8u);
    return (unsigned)std::min<size_t>(std::max(1u, nt), std::max<size_t>(1, n / 64));
}

// lays the jobs `ids` out for edit-distance capacity `band`: offsets first (serial prefix sums over sizes that the
// graphs already know), then the node/edge tables and the sequence bytes are filled by host threads
int pack_jobs(const std::vector<HostJob>& hj, const std::vector<uint32_t>& ids, uint32_t band, WfaPack& pk, bool big) {
    const size_t n = ids.size();
    pk.jobs.resize(n);
    // Reference slices: the reads of a block look at overlapping windows of ONE caller buffer (the chromosome), so
    // the union of the host address ranges is uploaded once and every job points into it (16 readable bytes follow
    // each range: the kernel's 16-byte compares may run past a node's last base).
    struct Range { const uint8_t* lo; const uint8_t* hi; uint64_t dev_off; };
    std::vector<Range> ranges;
    {
        std::vector<std::pair<const uint8_t*, uint32_t>> iv;
        iv.reserve(n);
        for (size_t i = 0; i < n; ++i) if (hj[ids[i]].ref_len) iv.push_back({hj[ids[i]].ref_ptr, hj[ids[i]].ref_len});
        std::sort(iv.begin(), iv.end());
        for (auto& v : iv) {
            if (!ranges.empty() && v.first <= ranges.back().hi) ranges.back().hi = std::max(ranges.back().hi, v.first + v.second);
            else ranges.push_back({v.first, v.first + v.second, 0});
        }
    }
    uint64_t node_off = 0, edge_off = 0, seq_off = 0;
    for (auto& r : ranges) { r.dev_off = seq_off; seq_off += ((uint64_t)(r.hi - r.lo) + 16 + 15) & ~15ull; }
    for (size_t i = 0; i < n; ++i) {
        const HostJob& g = hj[ids[i]];
        WfaJobDesc& jd = pk.jobs[i];
        jd = WfaJobDesc{};
        jd.node_off = node_off;
        jd.edge_off = edge_off;
        jd.seq_off = seq_off;
        if (g.ref_len) {
            auto it = std::upper_bound(ranges.begin(), ranges.end(), g.ref_ptr, [](const uint8_t* p, const Range& r) { return p < r.lo; });
            --it;
            jd.ref_off = it->dev_off + (uint64_t)(g.ref_ptr - it->lo);
        }
        jd.n_nodes = (uint32_t)g.nodes.size();
        jd.n_edges = (uint32_t)g.child.size();
        pk.max_edges = std::max(pk.max_edges, jd.n_edges);
        jd.set_words = (jd.n_nodes + 31) / 32;
        jd.read_off = g.read_off - g.ref_len;
        jd.read_len = g.read_len;
        jd.band = band;
        jd.out_set_off = pk.out_set_words;
        pk.out_set_words += jd.set_words;
        if (!big && jd.n_nodes > WFA_MAX_NODES) { set_error("internal: graph of %u nodes on the LDS path", jd.n_nodes); return HP_ERR_INVARIANT; }
        node_off += g.nodes.size();
        edge_off += g.child.size();
        seq_off += g.seq_bytes - g.ref_len;
        pk.max_nodes = std::max(pk.max_nodes, jd.n_nodes);
    }
    if (int rc = pk.nodes.resize(node_off)) return rc;
    if (int rc = pk.edges.resize(edge_off)) return rc;
    if (int rc = pk.seq.resize(seq_off)) return rc;
    const unsigned nt = wfa_host_threads(n);
    std::vector<int> rcs(nt, HP_OK);
    std::vector<std::string> errs(nt);
    std::vector<uint64_t> max_scratch(nt, 0);
    auto work = [&](unsigned t) {
        char msg[160];
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
            const HostJob& g = hj[ids[i]];
            WfaJobDesc& jd = pk.jobs[i];
            WfaNode* nodes = pk.nodes.data() + jd.node_off;
            WfaEdge* edges = pk.edges.data() + jd.edge_off;
            uint64_t entry_off = 0;
            uint32_t e_cur = 0;
            for (uint32_t k = 0; k < jd.n_nodes; ++k) {
                const HostNode& hn = g.nodes[k];
                WfaNode dn{};
                const bool is_ref = hn.seq_off < g.ref_len;
                dn.seq_off = is_ref ? hn.seq_off : hn.seq_off - g.ref_len;   // reference span | job-private bytes
                dn.seq_len = hn.seq_len;
                dn.child_off = e_cur;
                dn.n_children = (uint16_t)hn.n_child;
                const uint32_t np = k == 0 ? 1u : hn.n_par;
                if (np > WFA_MAX_PARENTS || hn.n_child > 65535) { snprintf(msg, sizeof msg, "graph node with %u parents (> %u supported)", np, WFA_MAX_PARENTS); errs[t] = msg; rcs[t] = HP_ERR_UNSUPPORTED; return; }
                dn.n_parents = (uint16_t)(np | (is_ref ? WFA_NODE_IS_REF : 0u));
                const int64_t width = (hn.emax - hn.emin) + 2 * (int64_t)band + 3;
                if (width > 65535) { snprintf(msg, sizeof msg, "diagonal band of %lld exceeds 65535", (long long)width); errs[t] = msg; rcs[t] = HP_ERR_UNSUPPORTED; return; }
                dn.dbase = (int32_t)(hn.emin - (int64_t)band - 1);
                dn.width = (uint32_t)width;
                dn.entry_stride = 5 + 2 * jd.set_words + np * jd.set_words;
                dn.entry_off = (uint32_t)entry_off;
                entry_off += (uint64_t)dn.width * dn.entry_stride;
                if (entry_off > 0xFFFFFFF0ull) { errs[t] = "WFA scratch of one read exceeds 16 GiB"; rcs[t] = HP_ERR_UNSUPPORTED; return; }
                for (uint32_t ci = 0; ci < hn.n_child; ++ci) {
                    const uint32_t c = g.child[hn.child_off + ci];
                    const uint32_t* cp = g.par.data() + g.nodes[c].par_off;
                    const uint32_t ord = (uint32_t)(std::lower_bound(cp, cp + g.nodes[c].n_par, k) - cp);
                    edges[e_cur++] = WfaEdge{c, ord};
                }
                nodes[k] = dn;
            }
            jd.scratch_dwords = (uint32_t)entry_off;
            max_scratch[t] = std::max<uint64_t>(max_scratch[t], entry_off);
            uint8_t* dst = pk.seq.data() + jd.seq_off;   // [alt allele bytes][read][pad]
            if (!g.alt.empty()) std::memcpy(dst, g.alt.data(), g.alt.size());
            if (g.read_len) std::memcpy(dst + jd.read_off, g.read_ptr, g.read_len);
            std::memset(dst + jd.read_off + g.read_len, 0, g.seq_bytes - g.read_off - g.read_len);
        }
        // this thread's share of the merged reference ranges, in 256 KiB pieces
        constexpr uint64_t PIECE = 256u << 10;
        uint64_t piece = 0;
        for (const Range& r : ranges) {
            const uint64_t len = (uint64_t)(r.hi - r.lo);
            for (uint64_t o = 0; o < len; o += PIECE, ++piece)
                if (piece % nt == t) std::memcpy(pk.seq.data() + r.dev_off + o, r.lo + o, (size_t)std::min(PIECE, len - o));
            if (piece % nt == t) std::memset(pk.seq.data() + r.dev_off + len, 0, (size_t)((((len + 16 + 15) & ~15ull)) - len));
        }
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    for (unsigned t = 0; t < nt; ++t) {
        if (rcs[t] != HP_OK) { set_error("%s", errs[t].c_str()); return rcs[t]; }
        pk.max_scratch = std::max(pk.max_scratch, max_scratch[t]);
    }
    return HP_OK;
}

// (every copy and launch of a pass goes to the calling thread's own stream and the pass waits for THAT stream only: the pass may
// run beside another stage's persistent kernels - a block stream - and must not wait for them)
template <class T> int up(DevBuf& buf, StageVec<T>& v, hipStream_t st) {
    int rc = buf.alloc(v.size() * sizeof(T));
    if (rc != HP_OK) return rc;
    return v.size() ? dev_put(buf.p, v.data(), v.size() * sizeof(T), st) : HP_OK;   // (hp_common.h: small transfers stay off the runtime's copy paths)
}
template <class T> int up(DevBuf& buf, const std::vector<T>& v, hipStream_t st) {
    int rc = buf.alloc(v.size() * sizeof(T));
    if (rc != HP_OK) return rc;
    return v.empty() ? HP_OK : dev_put(buf.p, v.data(), v.size() * sizeof(T), st);
}

// one launch over `ids` with capacity `band`; fills status/score/sets for those jobs
int run_pass(const std::vector<HostJob>& hj, const std::vector<uint32_t>& ids, uint32_t band, uint64_t prune, uint64_t max_ed,
             int n_cu, std::vector<int32_t>& status, std::vector<uint64_t>& score, std::vector<uint32_t>& sets, const std::vector<uint64_t>& set_off, bool big) {
    WfaPack pk;
    {
        size_t tot = 0, nn = 0;
        for (uint32_t id : ids) { tot += hj[id].seq_bytes; nn += hj[id].nodes.size(); }
        pk.jobs.reserve(ids.size());
    }
    const double t_pk0 = now_ms();
    int rc = pack_jobs(hj, ids, band, pk, big);
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] wfa pack_jobs %.2f ms\n", now_ms() - t_pk0); fflush(stderr); }
    if (rc != HP_OK) return rc;
    const size_t n = ids.size();
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return pk.jobs[a].read_len > pk.jobs[b].read_len; });
    const bool verbose = std::getenv("HP_DEBUG") != nullptr;
    const double t_pack = now_ms();
    int cur_dev = 0;
    HP_HIP_CHECK(hipGetDevice(&cur_dev));
    hipStream_t stm = thread_stream(cur_dev);
    if (!stm) { set_error("stream creation failed"); return HP_ERR_HIP; }
    struct StreamDrain { hipStream_t s; ~StreamDrain() { dev_io_abort(s); } } drain{stm};   // (waits for the stream; host vectors below may be read by async copies)
    DevBuf d_jobs, d_order, d_nodes, d_edges, d_seq, d_sets, d_score, d_status;
    if ((rc = up(d_jobs, pk.jobs, stm)) || (rc = up(d_order, order, stm)) || (rc = up(d_nodes, pk.nodes, stm)) || (rc = up(d_edges, pk.edges, stm)) ||
        (rc = d_seq.alloc(pk.seq.size())))
        return rc;
    if (pk.seq.size() && (rc = dev_put(d_seq.p, pk.seq.data(), pk.seq.size(), stm)) != HP_OK) return rc;
    if ((rc = d_sets.alloc(pk.out_set_words * 4 + 16)) || (rc = d_score.alloc(n * 8)) || (rc = d_status.alloc(n * 4))) return rc;
    std::vector<int32_t> st0(n, WFA_ST_PENDING);
    if ((rc = dev_put(d_status.p, st0.data(), n * 4, stm)) != HP_OK) return rc;
    const uint32_t lds_nodes_off = (uint32_t)(((size_t)pk.max_nodes * WFA_NODE_STATE_BYTES + 15) & ~(size_t)15);
    const uint32_t lds_edges_off = lds_nodes_off + pk.max_nodes * 32;
    const size_t lds = big ? 0 : (size_t)lds_edges_off + (size_t)pk.max_edges * sizeof(WfaEdge);   // big graphs: state in HBM, tables read in place
    // resident single-wave workgroups per CU: the kernel is a chain of dependent memory round trips per read, so
    // throughput comes from resident reads; 113 VGPRs allow 4 per SIMD = 16 per CU (measured, 4096 x 17 kb reads:
    // 8 -> 4.6 ms, 12 -> 4.3 ms, 16 -> 2.8 ms)
    const char* pcenv = std::getenv("HP_WFA_PER_CU");
    uint32_t per_cu = (uint32_t)std::min<size_t>(pcenv ? std::atoi(pcenv) : 16, (160 * 1024) / std::max<size_t>((lds + 1279) / 1280 * 1280, 1280));
    if (per_cu == 0) per_cu = 1;
    size_t free_b = 0, total_b = 0;
    HP_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const uint64_t stride = (pk.max_scratch + 63) & ~63ull;
    const size_t per_slot = (size_t)stride * 4;
    uint32_t slots = (uint32_t)std::min<size_t>({n, (size_t)n_cu * per_cu, std::max<size_t>(1, (free_b / 2) / std::max<size_t>(per_slot, 1))});
    if (slots == 0) slots = 1;
    if (g_ctx.device != cur_dev) { g_ctx.scratch.release(); g_ctx.device = cur_dev; g_ctx.dirty = true; }
    if (g_ctx.scratch.bytes < (size_t)slots * per_slot) {
        // The band scratch does not grow for ONE pass's sake when it is a block set's leftovers that ask (g_wfa_min_ed_hint: the late pass
        // of a stream): a pass's need is jobs x the widest band of any of them - 350 reads at 2-4 % noise x 10-35 MB on a HiFi-shaped
        // set, different from set to set - and a hipMalloc inside a stream waits for the resident launch sets (round 5, HiFi-shaped
        // bench: 8 of 28 late passes took 140-260 ms instead of 17-25, 1.3-1.6 M hets/s instead of 2 M+). The pass is served with the
        // slots the scratch HOLDS - its workgroups take the jobs in turns, the kernel was written that way - as long as that is a
        // reasonable number; the first such pass of a thread asks for HP_WFA_SCRATCH_FLOOR_MB (4 096) at once.
        const size_t have = g_ctx.scratch.bytes / std::max<size_t>(per_slot, 1);
        static const size_t floor_b = [] { const char* e = std::getenv("HP_WFA_SCRATCH_FLOOR_MB"); return (size_t)(e ? std::max(0, std::atoi(e)) : 4096) << 20; }();
        if (g_wfa_min_ed_hint && have >= std::min<size_t>(slots, 48)) slots = (uint32_t)std::min<size_t>(slots, have);
        else {
            size_t want = (size_t)slots * per_slot;
            if (g_wfa_min_ed_hint && want < floor_b && floor_b <= free_b / 4) want = floor_b;
            if ((rc = g_ctx.scratch.alloc(want)) != HP_OK) return rc;
            g_ctx.dirty = true;
        }
    }
    if (g_ctx.dirty) {
        HP_HIP_CHECK(hipMemsetAsync(g_ctx.scratch.p, 0, g_ctx.scratch.bytes, stm));
        g_ctx.dirty = false;
    }
    const double t_up = now_ms();
    WfaBatchDev B{};
    B.jobs = d_jobs.as<WfaJobDesc>(); B.order = d_order.as<uint32_t>(); B.n_items = (uint32_t)n;
    B.nodes = d_nodes.as<WfaNode>(); B.edges = d_edges.as<WfaEdge>(); B.seq = d_seq.as<uint8_t>();
    B.out_sets = d_sets.as<uint32_t>(); B.out_score = d_score.as<uint64_t>(); B.status = d_status.as<int32_t>();
    B.scratch = g_ctx.scratch.as<uint32_t>(); B.scratch_stride = stride; B.prune_distance = prune; B.max_ed = max_ed;
    B.lds_nodes_off = lds_nodes_off;
    B.lds_edges_off = lds_edges_off;
    DevBuf d_big;
    if (big) {
        B.big_stride = ((uint64_t)pk.max_nodes * WFA_NODE_STATE_BYTES + 63) & ~63ull;
        if ((rc = d_big.alloc((size_t)slots * B.big_stride)) != HP_OK) return rc;
        B.big_state = d_big.as<unsigned char>();
    }
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] wfa launch jobs=%zu band=%u slots=%u lds=%zu scratch/slot=%zu B\n", n, band, slots, lds, per_slot); fflush(stderr); }
    hipEvent_t e0, e1;
    HP_HIP_CHECK(hipEventCreate(&e0));
    HP_HIP_CHECK(hipEventCreate(&e1));
    HP_HIP_CHECK(hipEventRecord(e0, stm));
    if (big) hipLaunchKernelGGL(hp_wfa_big_kernel, dim3(slots), dim3(64), 0, stm, B);
    else hipLaunchKernelGGL(hp_wfa_kernel, dim3(slots), dim3(64), lds, stm, B);
    HP_HIP_CHECK(hipGetLastError());
    HP_HIP_CHECK(hipEventRecord(e1, stm));
    if (hipStreamSynchronize(stm) != hipSuccess) { g_ctx.dirty = true; set_error("WFA kernel failed"); return HP_ERR_HIP; }
    float kms = 0.f;
    HP_HIP_CHECK(hipEventElapsedTime(&kms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    g_last_kernel_ms += kms;
    if (verbose) { fprintf(stderr, "[hp] wfa pack+upload %.2f ms, alloc/memset %.2f ms, kernel %.3f ms\n", t_up - t_pack, now_ms() - t_up - kms, kms); fflush(stderr); }
    const double t_dl = now_ms();
    std::vector<int32_t> st(n);
    std::vector<uint64_t> sc(n);
    std::vector<uint32_t> all_sets(pk.out_set_words + 4);
    if ((rc = dev_get(st.data(), d_status.p, n * 4, stm)) || (rc = dev_get(sc.data(), d_score.p, n * 8, stm)) || (rc = dev_get(all_sets.data(), d_sets.p, pk.out_set_words * 4, stm))) return rc;
    if (dev_io_sync(stm) != HP_OK) { set_error("WFA result download failed"); return HP_ERR_HIP; }
    for (size_t i = 0; i < n; ++i) {
        status[ids[i]] = st[i];
        score[ids[i]] = sc[i];
        std::memcpy(sets.data() + set_off[ids[i]], all_sets.data() + pk.jobs[i].out_set_off, (size_t)pk.jobs[i].set_words * 4);
    }
    if (verbose) { fprintf(stderr, "[hp] wfa download+scatter %.2f ms (pass so far %.2f ms)\n", now_ms() - t_dl, now_ms() - t_pk0); fflush(stderr); }
    return HP_OK;
}

}  // namespace
}  // namespace hp

namespace hp {
int wfa_assign_batch_v2(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out,
                        uint8_t* const* alleles, int device_id);   // hp_wfa2.hip
int wfa_assign_batch_v1(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out,
                        uint8_t* const* alleles, int device_id);
extern thread_local const uint32_t* g_wfa_min_ed_hint;   // hp_wfa2_host.h
}
using namespace hp;

// Two kernels, one result: a batch large enough to fill the chip goes to the compact several-reads-per-wavefront kernel
// (hp_wfa2.hip: ~5x fewer instructions per read, throughput-bound); a small batch is latency-bound and goes to the
// one-read-per-wavefront dense-band kernel below (a lone wavefront steps ~3x faster through its read than a group that
// shares its wavefront with seven others). HP_WFA2_MIN_JOBS moves the switch (0: always the compact kernel,
// a huge value: never); the compact path hands whatever outgrows its state back to this one.
static int wfa_assign_dispatch(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed,
                               hp_wfa_result* out, uint8_t* const* alleles, int device_id) {
    const char* mj = std::getenv("HP_WFA2_MIN_JOBS");
    const size_t min_jobs = mj ? (size_t)std::strtoull(mj, nullptr, 10) : 4608;
    return n < min_jobs ? wfa_assign_batch_v1(jobs, n, prune_distance, max_ed, out, alleles, device_id)
                        : wfa_assign_batch_v2(jobs, n, prune_distance, max_ed, out, alleles, device_id);
}

// Calls that are in flight together (one per block from HiPhase's thread pool) are merged into one batch: hp_combine.h.
namespace {
struct WfaReq {
    const hp_wfa_job* jobs; size_t n; uint64_t prune, max_ed; hp_wfa_result* out; uint8_t* const* alleles; int device;
    int rc = HP_OK; std::string err; bool done = false;
};
void run_wfa_batch(std::vector<WfaReq*>& batch);
// never destroyed: its service thread may outlive every static destructor
hp::Combiner<WfaReq>& g_wfa_combiner() { static auto* c = new hp::Combiner<WfaReq>(run_wfa_batch); return *c; }
void run_wfa_batch(std::vector<WfaReq*>& batch) {
    std::vector<char> taken(batch.size(), 0);
    for (size_t i = 0; i < batch.size(); ++i) {
        if (taken[i]) continue;
        std::vector<WfaReq*> grp;
        for (size_t j = i; j < batch.size(); ++j)
            if (!taken[j] && batch[j]->device == batch[i]->device && batch[j]->prune == batch[i]->prune && batch[j]->max_ed == batch[i]->max_ed &&
                (batch[j]->alleles != nullptr) == (batch[i]->alleles != nullptr)) { taken[j] = 1; grp.push_back(batch[j]); }
        if (grp.size() == 1) {
            WfaReq* r = grp[0];
            r->rc = wfa_assign_dispatch(r->jobs, r->n, r->prune, r->max_ed, r->out, r->alleles, r->device);
            if (r->rc != HP_OK) r->err = hp_last_error();
            continue;
        }
        size_t tot = 0;
        for (WfaReq* r : grp) tot += r->n;
        std::vector<hp_wfa_job> jobs;
        std::vector<hp_wfa_result> out(tot);
        std::vector<uint8_t*> al;
        jobs.reserve(tot); al.reserve(tot);
        for (WfaReq* r : grp) {
            jobs.insert(jobs.end(), r->jobs, r->jobs + r->n);
            for (size_t k = 0; k < r->n; ++k) al.push_back(r->alleles ? r->alleles[k] : nullptr);
        }
        const int rc = wfa_assign_dispatch(jobs.data(), tot, grp[0]->prune, grp[0]->max_ed, out.data(), grp[0]->alleles ? al.data() : nullptr, grp[0]->device);
        if (rc == HP_OK) {
            size_t o = 0;
            for (WfaReq* r : grp) { std::copy(out.begin() + o, out.begin() + o + r->n, r->out); o += r->n; r->rc = HP_OK; }
            continue;
        }
        for (WfaReq* r : grp) {   // a malformed job fails the merged call: every caller gets the status of its own jobs
            r->rc = wfa_assign_dispatch(r->jobs, r->n, r->prune, r->max_ed, r->out, r->alleles, r->device);
            if (r->rc != HP_OK) r->err = hp_last_error();
        }
    }
}
}  // namespace

// Two kernels, one result: a batch large enough to fill the chip goes to the compact several-reads-per-wavefront kernel
// (hp_wfa2.hip: ~5x fewer instructions per read, throughput-bound); a small batch is latency-bound and goes to the
// one-read-per-wavefront dense-band kernel below (a lone wavefront steps ~3x faster through its read than a group that
// shares its wavefront with seven others). HP_WFA2_MIN_JOBS moves the switch (0: always the compact kernel,
// a huge value: never); the compact path hands whatever outgrows its state back to this one.
extern "C" int hp_wfa_assign_batch(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed,
                                   hp_wfa_result* out, uint8_t* const* alleles, int device_id) {
    const double t0 = now_ms();
    int rc;
    if (n == 0 || !jobs || !out || !hp::Combiner<WfaReq>::enabled()) rc = wfa_assign_dispatch(jobs, n, prune_distance, max_ed, out, alleles, device_id);
    else {
        WfaReq r{jobs, n, prune_distance, max_ed, out, alleles, device_id < 0 ? hp_default_device() : device_id};
        g_wfa_combiner().submit(&r);
        rc = r.rc;
        if (rc != HP_OK) set_error("%s", r.err.c_str());
    }
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] hp_wfa_assign_batch total %.2f ms\n", now_ms() - t0); fflush(stderr); }
    return rc;
}
// edit_distance_with_pruning (wfa_graph.rs:350-650) for finished host graphs on the dense-band kernels: status, score and
// the traversed-node bitsets ((n_nodes + 31) / 32 words per job at set_off[job]).
static int run_graphs(const std::vector<HostJob>& hj, uint64_t prune_distance, uint64_t max_ed, int device_id,
                      std::vector<int32_t>& status, std::vector<uint64_t>& score, std::vector<uint32_t>& sets, std::vector<uint64_t>& set_off,
                      const uint32_t* min_ed_hint = nullptr) {
    const size_t n = hj.size();
    // host-side work is done; from here on a GPU is mandatory (no CPU fallback)
    if (device_id < 0) device_id = hp_default_device();
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed - no usable GPU; there is no CPU fallback", device_id); return HP_ERR_HIP; }
    const int n_cu = device_cu_count(device_id);
    status.assign(n, WFA_ST_PENDING);
    score.assign(n, 0);
    // traversed-node bitsets of all jobs in one flat array ((n_nodes + 31) / 32 words per job)
    set_off.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) set_off[i + 1] = set_off[i] + (hj[i].nodes.size() + 31) / 32;
    sets.assign(set_off[n], 0);
    // graphs within the LDS budget and graphs beyond it (state in HBM) run as separate launches; each starts with a
    // narrow band (most reads finish within a few dozen edits) and re-runs what needs more at full width
    const char* benv = std::getenv("HP_WFA_BAND");
    for (int big = 0; big < 2; ++big) {
        std::vector<uint32_t> ids;
        for (size_t i = 0; i < n; ++i) if ((hj[i].nodes.size() > WFA_MAX_NODES) == (big == 1)) ids.push_back((uint32_t)i);
        if (ids.empty()) continue;
        uint32_t band = (uint32_t)std::min<uint64_t>(max_ed, benv ? (uint64_t)std::atoi(benv) : 96);
        std::vector<uint32_t> wide;   // known to need more than the narrow band (g_wfa_min_ed_hint): they join the second pass
        // The leftovers of a block set's compact kernels (a few dozen reads with hints) take ONE pass at full width: their pass is what
        // the set's rows wait for, two launches are twice one launch's latency (8 ms each beside a resident launch set), and the
        // narrow attempt mostly fails for them - they are here because their alignment is not an ordinary one.
        static const size_t one_pass_max = [] { const char* e = std::getenv("HP_WFA_ONE_PASS_MAX"); return e ? (size_t)std::max(0, std::atoi(e)) : (size_t)512; }();
        if (min_ed_hint && band < max_ed && ids.size() <= one_pass_max && max_ed <= 2000) band = (uint32_t)max_ed;
        if (min_ed_hint && band < max_ed) {
            std::vector<uint32_t> narrow;
            for (uint32_t id : ids) (min_ed_hint[id] >= band ? wide : narrow).push_back(id);
            ids.swap(narrow);
        }
        for (;;) {
            {   // jobs outside what a launch at this band can hold get the soft per-job verdict instead of failing the call
                std::vector<uint32_t> fit;
                for (uint32_t id : ids) {
                    const HostJob& g = hj[id];
                    const uint64_t set_words = (g.nodes.size() + 31) / 32;
                    uint64_t entry = 0;
                    bool ok = true;
                    for (size_t k = 0; k < g.nodes.size() && ok; ++k) {
                        const uint32_t np = k == 0 ? 1u : g.nodes[k].n_par;
                        const int64_t width = (g.nodes[k].emax - g.nodes[k].emin) + 2 * (int64_t)band + 3;
                        if (np > WFA_MAX_PARENTS || g.nodes[k].n_child > 65535 || width > 65535) ok = false;
                        entry += (uint64_t)width * (5 + 2 * set_words + np * set_words);
                        if (entry > 0xFFFFFFF0ull) ok = false;
                    }
                    if (ok) fit.push_back(id); else status[id] = WFA_ST_UNSUPPORTED;
                }
                ids.swap(fit);
            }
            if (!ids.empty()) {
                int rc = run_pass(hj, ids, band, prune_distance, max_ed, n_cu, status, score, sets, set_off, big == 1);
                if (rc != HP_OK) return rc;
            }
            std::vector<uint32_t> again;
            for (uint32_t id : ids) if (status[id] == WFA_ST_NEED_BAND) again.push_back(id);
            again.insert(again.end(), wide.begin(), wide.end());
            wide.clear();
            if (again.empty()) break;
            if (band >= max_ed) { set_error("WFA band overflow at full width (internal)"); return HP_ERR_INVARIANT; }
            band = (uint32_t)std::min<uint64_t>(max_ed, (uint64_t)band * 6);
            ids.swap(again);
        }
    }
    for (size_t i = 0; i < n; ++i)
        if (status[i] != WFA_ST_OK && status[i] != WFA_ST_MAX_ED && status[i] != WFA_ST_UNSUPPORTED) { set_error("job %zu: device status %d", i, status[i]); return HP_ERR_INVARIANT; }
    return HP_OK;
}

thread_local const uint32_t* hp::g_wfa_min_ed_hint = nullptr;

int hp::wfa_assign_batch_v1(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed,
                            hp_wfa_result* out, uint8_t* const* alleles, int device_id) {
    if (n == 0) return HP_OK;
    if (!jobs || !out) { set_error("null argument"); return HP_ERR_ARG; }
    if (n > 0x7FFFFFFFull) { set_error("too many jobs"); return HP_ERR_ARG; }
    if (max_ed > 60000) {   // (a band that wide does not fit the kernels' 16-bit diagonal indices: every job of the call, softly)
        for (size_t i = 0; i < n; ++i) { out[i] = hp_wfa_result{HP_WFA_UNSUPPORTED, 0, 0}; if (alleles && alleles[i]) for (uint32_t k = 0; k < jobs[i].n_hets; ++k) alleles[i][k] = HP_ALLELE_NOOVERLAP; }
        return HP_OK;
    }
    std::vector<GraphArena> arenas;
    std::vector<HostJob> hj(n);
    g_last_kernel_ms = 0.0;   // (the caller adds the compact path's time back when this runs its leftovers)
    const double t_build = now_ms();
    {
        // graph construction is independent per read: spread it over host threads (HP_WFA_HOST_THREADS, default
        // min(8, cores)); the caller's own thread pool (main.rs:332) composes with this
        const unsigned nt = wfa_host_threads(n);
        std::atomic<int> first_rc{HP_OK};
        std::vector<std::string> errs(nt);
        arenas.resize(nt);
        auto work = [&](unsigned t) {
            GraphBuild g;
            GraphArena& arena = arenas[t];
            for (size_t i = t; i < n && first_rc.load(std::memory_order_relaxed) == HP_OK; i += nt) {
                int rc = HP_OK;
                g.clear();
                if (!jobs[i].reference || (!jobs[i].read && jobs[i].read_len)) { set_error("job %zu: null sequence", i); rc = HP_ERR_ARG; }
                else rc = build_graph(&jobs[i], g);
                if (rc != HP_OK) {
                    int exp = HP_OK;
                    if (first_rc.compare_exchange_strong(exp, rc)) errs[t] = hp_last_error();
                    return;
                }
                hj[i].stash(g, arena);
            }
            for (size_t i = t; i < n; i += nt) hj[i].bind(arena);
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
        }
        if (first_rc.load() != HP_OK) {
            for (auto& e : errs) if (!e.empty()) { set_error("%s", e.c_str()); break; }
            return first_rc.load();
        }
    }
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] wfa graph build %.2f ms for %zu jobs\n", now_ms() - t_build, n); fflush(stderr); }
    std::vector<int32_t> status;
    std::vector<uint64_t> score, set_off;
    std::vector<uint32_t> sets;
    {
        const int rc = run_graphs(hj, prune_distance, max_ed, device_id, status, score, sets, set_off, g_wfa_min_ed_hint);
        if (rc != HP_OK) return rc;
    }
    const double t_map = now_ms();
    auto map_jobs = [&](unsigned t, unsigned nt) {
        for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
            out[i].status = status[i] == WFA_ST_OK ? HP_OK : status[i] == WFA_ST_UNSUPPORTED ? HP_WFA_UNSUPPORTED : HP_WFA_MAX_ED;
            out[i].n_nodes = (uint32_t)hj[i].nodes.size();
            out[i].score = score[i];
            if (alleles && alleles[i]) {
                uint8_t* a = alleles[i];
                for (uint32_t k = 0; k < jobs[i].n_hets; ++k) a[k] = HP_ALLELE_NOOVERLAP;
                if (status[i] == WFA_ST_OK) {
                    // read_parsing.rs:790-800: traversed nodes ascending; first assignment wins, a different one -> Ambiguous
                    const uint32_t* set = sets.data() + set_off[i];
                    for (auto& t3 : hj[i].tags) {   // sorted by node in build_graph
                        if (!((set[t3[0] >> 5] >> (t3[0] & 31)) & 1u)) continue;
                        if (a[t3[1]] == HP_ALLELE_NOOVERLAP) a[t3[1]] = (uint8_t)t3[2];
                        else if (a[t3[1]] != (uint8_t)t3[2]) a[t3[1]] = HP_ALLELE_AMBIGUOUS;
                    }
                }
            }
        }
    };
    {
        const unsigned nt = wfa_host_threads(n);
        if (nt == 1) map_jobs(0, 1);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(map_jobs, t, nt);
            for (auto& x : th) x.join();
        }
    }
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] wfa allele mapping %.2f ms (function body so far %.2f ms)\n", now_ms() - t_map, now_ms() - t_build); fflush(stderr); }
    return HP_OK;
}


// WFAGraph::new + add_node (wfa_graph.rs:100-117, :298-331) + edit_distance_with_pruning (:350) for caller-built graphs:
// the reference's API below from_reference_variants_with_hom, which its own tests drive with hand-built topologies
// (nested / overlapping / triple splits, :677-839) that no variant set produces. Dense-band kernels.
extern "C" int hp_wfa_align_graphs(const hp_graph_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed, hp_graph_result* out,
                                   uint32_t* const* traversed, int device_id) {
    if (n == 0) return HP_OK;
    if (!jobs || !out) { set_error("null argument"); return HP_ERR_ARG; }
    if (n > 0x7FFFFFFFull) { set_error("too many jobs"); return HP_ERR_ARG; }
    if (max_ed > 60000) {
        for (size_t i = 0; i < n; ++i) { out[i] = hp_graph_result{HP_WFA_UNSUPPORTED, 0, 0}; if (traversed && traversed[i]) for (uint32_t w = 0; w < (jobs[i].n_nodes + 31) / 32; ++w) traversed[i][w] = 0; }
        return HP_OK;
    }
    GraphArena arena;
    std::vector<HostJob> hj(n);
    GraphBuild g;
    std::vector<uint32_t> parents;
    for (size_t i = 0; i < n; ++i) {
        const hp_graph_job& J = jobs[i];
        if (!J.nodes || J.n_nodes == 0 || (!J.read && J.read_len)) { set_error("job %zu: null graph or read", i); return HP_ERR_ARG; }
        g.clear();
        for (uint32_t k = 0; k < J.n_nodes; ++k) {
            const hp_graph_node& nd = J.nodes[k];
            if ((!nd.seq && nd.seq_len) || (!nd.parents && nd.n_parents)) { set_error("job %zu node %u: null array", i, k); return HP_ERR_ARG; }
            parents.assign(nd.parents, nd.parents + nd.n_parents);
            const uint32_t off = (uint32_t)g.alt.size();
            g.alt.insert(g.alt.end(), nd.seq, nd.seq + nd.seq_len);
            // the reference asserts: the first node has no parents, every other node has some, all of them earlier (:305-312)
            if (add_node(g, off, nd.seq_len, parents) < 0) { set_error("job %zu node %u: assert of WFAGraph::add_node (wfa_graph.rs:305-312)", i, k); return HP_ERR_INVARIANT; }
        }
        g.ref_ptr = nullptr; g.ref_len = 0;
        g.read_off = (uint32_t)g.alt.size();
        g.read_len = J.read_len;
        g.read_ptr = J.read;
        g.seq_bytes = (g.read_off + g.read_len + 16 + 15) & ~15u;
        finish_graph(g);
        hj[i].stash(g, arena);
    }
    for (size_t i = 0; i < n; ++i) hj[i].bind(arena);
    std::vector<int32_t> status;
    std::vector<uint64_t> score, set_off;
    std::vector<uint32_t> sets;
    const int rc = run_graphs(hj, prune_distance, max_ed, device_id, status, score, sets, set_off);
    if (rc != HP_OK) return rc;
    for (size_t i = 0; i < n; ++i) {
        out[i].status = status[i] == WFA_ST_OK ? HP_OK : status[i] == WFA_ST_UNSUPPORTED ? HP_WFA_UNSUPPORTED : HP_WFA_MAX_ED;
        out[i].score = score[i];
        const uint32_t words = (jobs[i].n_nodes + 31) / 32;
        uint32_t cnt = 0;
        for (uint32_t w = 0; w < words; ++w) {
            const uint32_t v = status[i] == WFA_ST_OK ? sets[set_off[i] + w] : 0u;
            cnt += (uint32_t)__builtin_popcount(v);
            if (traversed && traversed[i]) traversed[i][w] = v;
        }
        out[i].n_traversed = cnt;
    }
    return HP_OK;
}
