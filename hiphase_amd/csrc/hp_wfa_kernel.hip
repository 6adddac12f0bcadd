// hp_wfa_kernel.hip — graph-WFA (unit costs, end-to-end) on gfx950: one wavefront per read.
//
// Replaces, bit-identically, reference src/wfa_graph.rs:350-650 `edit_distance_with_pruning`:
//   * lanes = diagonals of the node being processed; nodes are visited in index (topological) order inside an
//     edit-distance round because a wave that reaches the end of a node is handed to its successors in the SAME
//     round (wfa_graph.rs:527-553);
//   * the hash maps of the reference become a dense per-node diagonal band (hp_wfa_dev.h); the next round
//     PULLS its candidates (d+1: offset+1, d: offset+1, d-1: offset) instead of receiving pushes;
//   * candidates on one diagonal are extended by 8-byte compares (+ a wave-cooperative 512-byte/step mode for
//     long exact runs); ties after extension union their traversed-node bitsets (wfa_graph.rs:476-510);
//   * the strict `<` stale test against maxfront, the min_progression pruning floor that only moves between
//     rounds, and the "finals are collected from ALL waves of the last node" rule are kept as they are
//     (wfa_graph.rs:465, :638-640, :576-588).
// Integer/byte work; bound by memory latency, not MFMA.
#include "hp_common.h"
#include "hp_wfa_dev.h"

namespace hp {

#define WDEV __device__ __forceinline__
#define WFA_PROF 0
#if WFA_PROF
#define PT(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); pc[i] += t_ - tl; tl = t_; } while (0)
#define PC(i, v) do { pn[i] += (v); } while (0)
#else
#define PT(i)
#define PC(i, v)
#endif
__device__ uint32_t g_p2_iters, g_p2_lanes;

extern __shared__ __attribute__((aligned(16))) unsigned char wfa_smem[];
// LDS node state: 4 hulls of u16 pairs (lo, hi), empty = (0xFFFF, 0)
struct NodeState {
    uint32_t hull[2];   // results written in even / odd rounds: lo << 16 | hi
    uint32_t stamp[2];  // round in which hull[parity] was written (stale hulls must not be pulled from)
    uint32_t inj;       // pending same-round injections
    uint32_t ever;      // everything ever touched (cleared at job end)
};
constexpr uint32_t HULL_EMPTY = 0xFFFF0000u;
WDEV uint32_t hull_lo(uint32_t h) { return h >> 16; }
WDEV uint32_t hull_hi(uint32_t h) { return h & 0xFFFFu; }
WDEV bool hull_empty(uint32_t h) { return hull_lo(h) > hull_hi(h); }
WDEV uint32_t hull_make(uint32_t lo, uint32_t hi) { return (lo << 16) | hi; }
WDEV uint32_t hull_union(uint32_t a, uint32_t b) {
    if (hull_empty(a)) return b;
    if (hull_empty(b)) return a;
    return hull_make(min(hull_lo(a), hull_lo(b)), max(hull_hi(a), hull_hi(b)));
}
WDEV bool hull_has(uint32_t h, int32_t di) { return di >= (int32_t)hull_lo(h) && di <= (int32_t)hull_hi(h); }

WDEV uint32_t wlane() { return __lane_id(); }
WDEV uint32_t wb32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
WDEV uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m));
    return wb32(v);
}
WDEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m));
    return wb32(v);
}
WDEV uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v |= (uint32_t)__shfl_xor((int)v, m);
    return wb32(v);
}

WDEV uint64_t ld8(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Length of the common prefix of a[0..maxlen) and b[0..maxlen) for every lane (maxlen == 0 for idle lanes).
// Phase 1: every lane compares its first 16 bytes in ONE memory round trip (four independent 8-byte loads) - a
// diagonal that is not the alignment's own mismatches within a base or two, so this settles almost every lane.
// Phase 2: lanes that are still matching are served one after the other by the whole wave, 512 bytes per step
// (coalesced 8-byte loads + ballot). Buffers are padded so that reading 16 bytes at any in-range position is legal.
struct Pre16 { uint64_t a0, a1, b0, b1; };
// issues the four 8-byte loads of a phase-1 compare without using them: several of these are put in flight
// together (the extension and its tie checks), so a (round, node) step pays ONE sequence round trip
WDEV Pre16 pre16(const uint8_t* a, const uint8_t* b, bool on) {
    Pre16 p{0, 0, 0, 0};
    if (on) { p.a0 = ld8(a); p.a1 = ld8(a + 8); p.b0 = ld8(b); p.b1 = ld8(b + 8); }
    return p;
}
WDEV uint32_t pre16_len(const Pre16& p) {   // common prefix of the two 16-byte windows
    const uint64_t x0 = p.a0 ^ p.b0, x1 = p.a1 ^ p.b1;
    return x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : (x1 ? 8u + ((uint32_t)__builtin_ctzll(x1) >> 3) : 16u);
}
WDEV uint32_t match_rest(const uint8_t* a, const uint8_t* b, uint32_t maxlen, uint32_t n, bool done);
WDEV uint32_t match_run(const uint8_t* a, const uint8_t* b, uint32_t maxlen) {
    uint32_t n = 0;
    bool done = (maxlen == 0);
    if (__any(!done)) {
        if (!done) {
            uint32_t m = pre16_len(pre16(a, b, true));
            if (m > maxlen) m = maxlen;
            n = m;
            if (m < 16 || n >= maxlen) done = true;
        }
    }
    return match_rest(a, b, maxlen, n, done);
}
// phase 2: lanes with `done == false` have matched their first n (= 16) bytes and have more to compare
WDEV uint32_t match_rest(const uint8_t* a, const uint8_t* b, uint32_t maxlen, uint32_t n, bool done) {
    uint64_t pending = __ballot(!done);
    while (pending) {
        const int L = __builtin_ctzll(pending);
        pending &= pending - 1;
        // broadcast lane L's cursor
        const uint64_t pa = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)(a + n) >> 32), L) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)(a + n), L);
        const uint64_t pb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)(b + n) >> 32), L) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)(b + n), L);
        const uint32_t rem = (uint32_t)__builtin_amdgcn_readlane((int)(maxlen - n), L);
        const uint8_t* ga = reinterpret_cast<const uint8_t*>(pa);
        const uint8_t* gb = reinterpret_cast<const uint8_t*>(pb);
        uint32_t got = 0;  // uniform
        for (uint32_t base = 0; base < rem; base += 512) {
            const uint32_t off = base + wlane() * 8;
            uint32_t m = 8;
            if (off < rem) {
                const uint64_t x = ld8(ga + off) ^ ld8(gb + off);
                m = x ? ((uint32_t)__builtin_ctzll(x) >> 3) : 8u;
                if (m > rem - off) m = rem - off;
            } else {
                m = 0;  // beyond the end: acts as a stop
            }
            const uint64_t stop = __ballot(m < 8);
            if (stop) {
                const int S = __builtin_ctzll(stop);
                got = base + (uint32_t)S * 8 + (uint32_t)__builtin_amdgcn_readlane((int)m, S);
                break;
            }
            got = base + 512;
        }
        if (got > rem) got = rem;
        if ((int)wlane() == L) n += got;
    }
    return n;
}

struct WfaNodeU {  // uniform copy of a WfaNode
    uint32_t seq_off, seq_len, child_off, n_children, n_parents, width, entry_off, entry_stride;
    int32_t dbase;
    bool is_ref;
};
// The job's node table is copied into LDS once (solve_job): a node lookup is then a broadcast LDS read instead of a
// dependent HBM round trip at the head of every (round, node) step.
WDEV WfaNodeU load_node(const WfaNode* p) {
    const uint4* s = reinterpret_cast<const uint4*>(p);
    const uint4 a = s[0], b = s[1];   // every lane reads the same address
    WfaNodeU u;
    u.seq_off = wb32(a.x);
    u.seq_len = wb32(a.y);
    u.child_off = wb32(a.z);
    const uint32_t cp = wb32(a.w);
    u.n_children = cp & 0xFFFFu;
    u.n_parents = (cp >> 16) & (WFA_NODE_IS_REF - 1);
    u.is_ref = (cp >> 16) & WFA_NODE_IS_REF;
    u.dbase = (int32_t)wb32(b.x);
    u.width = wb32(b.y);
    u.entry_off = wb32(b.z);
    u.entry_stride = wb32(b.w);
    return u;
}

// BIG: graphs beyond the LDS budget (WFA_MAX_NODES) keep their per-node state in HBM and read the node / edge tables
// where the host put them; everything else is the same code.
template <bool BIG> WDEV void solve_job(const WfaBatchDev& B, uint32_t job, uint32_t slot) {
    const uint32_t lane = wlane();
    uint64_t pc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; uint32_t pn[8] = {0,0,0,0,0,0,0,0};
    uint64_t tl = __builtin_amdgcn_s_memtime();
    const uint64_t tstart = tl;
    const WfaJobDesc jd = B.jobs[job];
    const uint32_t n_nodes = jd.n_nodes, W = jd.set_words;
    const WfaNode* gnodes = B.nodes + jd.node_off;
    const WfaEdge* gedges = B.edges + jd.edge_off;
    const WfaEdge* edges = BIG ? gedges : reinterpret_cast<const WfaEdge*>(wfa_smem + (size_t)B.lds_edges_off);
    const uint8_t* seq = B.seq + jd.seq_off;
    const uint8_t* refseq = B.seq + jd.ref_off;
    const uint8_t* read = seq + jd.read_off;
    const uint32_t other_len = jd.read_len;
    uint32_t* scr = B.scratch + (size_t)slot * B.scratch_stride;
    NodeState* ns = BIG ? reinterpret_cast<NodeState*>(B.big_state + (size_t)slot * B.big_stride) : reinterpret_cast<NodeState*>(wfa_smem);
    const WfaNode* nodes = BIG ? gnodes : reinterpret_cast<const WfaNode*>(wfa_smem + (size_t)B.lds_nodes_off);
    uint32_t* out_set = B.out_sets + jd.out_set_off;

    for (uint32_t i = lane; i < n_nodes; i += 64) ns[i] = NodeState{{HULL_EMPTY, HULL_EMPTY}, {0xFFFFFFFFu, 0xFFFFFFFFu}, HULL_EMPTY, HULL_EMPTY};
    if (!BIG) {
        uint4* ln = reinterpret_cast<uint4*>(wfa_smem + (size_t)B.lds_nodes_off);
        WfaEdge* le = reinterpret_cast<WfaEdge*>(wfa_smem + (size_t)B.lds_edges_off);
        for (uint32_t i = lane; i < n_nodes * 2; i += 64) ln[i] = reinterpret_cast<const uint4*>(gnodes)[i];   // 32-byte node entries as 16-byte halves
        for (uint32_t i = lane; i < jd.n_edges; i += 64) le[i] = gedges[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (lane == 0) for (uint32_t w = 0; w < W; ++w) out_set[w] = 0;

    int32_t status = WFA_ST_PENDING;
    uint64_t score = 0;
    // start wave: node 0, diagonal 0, offset 0, set {0}  (wfa_graph.rs:366-378) = an injection from a virtual parent
    {
        const WfaNodeU n0 = load_node(nodes);
        const int32_t di0 = 0 - n0.dbase;
        if (di0 < 0 || di0 >= (int32_t)n0.width) status = WFA_ST_INTERNAL;
        else if (lane == 0) {
            uint32_t* e = scr + n0.entry_off + (size_t)di0 * n0.entry_stride + 5 + 2 * W;
            e[0] = 1u;
            ns[0].inj = hull_make((uint32_t)di0, (uint32_t)di0);
            ns[0].ever = hull_make((uint32_t)di0, (uint32_t)di0);
        }
    }
    uint32_t amin = 0, amax = 0;  // range of nodes that may have work
    uint64_t farthest = 0, min_prog = 0;
    const uint32_t last = n_nodes - 1;

    PT(0);
    for (uint32_t ed = 0; status == WFA_ST_PENDING; ++ed) {
        PC(0, 1);
        const uint32_t c = ed & 1u, p = c ^ 1u;
        uint32_t lane_far = 0;
        bool final_found = false;
        uint32_t new_amin = 0xFFFFFFFFu, new_amax = 0;
        bool band_overflow = false;
        for (uint32_t n = amin; n <= amax && n < n_nodes; ++n) {
            // results / injections written by other lanes of THIS wavefront (the only one of the workgroup): the
            // memory pipeline executes a wave's accesses in order, so only the compiler must not move them
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t ph = (ed > 0 && ns[n].stamp[p] == ed - 1) ? ns[n].hull[p] : HULL_EMPTY;
            const uint32_t ih = ns[n].inj;
            if (hull_empty(ph) && hull_empty(ih)) { PC(1, 1); PT(1); continue; }
            const WfaNodeU nd = load_node(nodes + n);
            // this round's hull: previous results grown by one diagonal each side, plus the injection targets
            uint32_t lo, hi;
            {
                uint32_t g = HULL_EMPTY;
                if (!hull_empty(ph)) {
                    const uint32_t glo = hull_lo(ph) > 0 ? hull_lo(ph) - 1 : 0;
                    uint32_t ghi = hull_hi(ph) + 1;
                    if (ghi >= nd.width) ghi = nd.width - 1;
                    g = hull_make(glo, ghi);
                }
                g = hull_union(g, ih);
                lo = hull_lo(g);
                hi = hull_hi(g);
            }
            const uint8_t* nseq = (nd.is_ref ? refseq : seq) + nd.seq_off;
            const uint32_t len = nd.seq_len;
            const uint32_t ES = nd.entry_stride;
            uint32_t* ebase = scr + nd.entry_off;
            bool any_valid = false;
            uint32_t vlo = 0xFFFFu, vhi = 0;   // diagonals that produced a wave the next round can pull from
            bool any_final_here = false;

            PT(2);
            for (uint32_t base = lo; base <= hi; base += 64) {
                PC(2, 1);
                const int32_t di = (int32_t)(base + lane);
                const bool act = (uint32_t)di <= hi;
                const int32_t d = nd.dbase + di;  // other_start
                uint32_t* e = ebase + (size_t)(act ? di : (int32_t)lo) * ES;
                // ---- gather candidates -------------------------------------------------------------------
                // A: from d+1 (offset+1)  B: from d (offset+1)  C: from d-1 (offset)  D_k: injections (offset 0)
                int32_t oA = -1, oB = -1, oC = -1;
                uint32_t mf = 0;   // max_wavefronts of this diagonal, fetched with the candidates (same round trip)
                // the first four set words of each candidate travel with its (offset, kind): graphs of up to 128 nodes
                // (W <= 4) never need a second round trip for the union of the tied sets
                uint32_t qA[4] = {0u, 0u, 0u, 0u}, qB[4] = {0u, 0u, 0u, 0u}, qC[4] = {0u, 0u, 0u, 0u};
                if (act) {
                    mf = e[0];
                    if (hull_has(ph, di + 1)) {
                        const uint32_t* r = e + ES + 1 + p * (2 + W);
                        const uint32_t r0 = r[0], r1 = r[1];
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) if (j < W) qA[j] = r[2 + j];
                        if (r1 & 1u) oA = (int32_t)r0 + 1;
                    }
                    if (hull_has(ph, di)) {
                        const uint32_t* r = e + 1 + p * (2 + W);
                        const uint32_t r0 = r[0], r1 = r[1];
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) if (j < W) qB[j] = r[2 + j];
                        if (r1 == WFA_KIND_INTERIOR_READ) oB = (int32_t)r0 + 1;
                    }
                    if (hull_has(ph, di - 1)) {
                        const uint32_t* r = e - ES + 1 + p * (2 + W);
                        const uint32_t r0 = r[0], r1 = r[1];
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) if (j < W) qC[j] = r[2 + j];
                        if (r1 == WFA_KIND_INTERIOR_READ || r1 == WFA_KIND_END_LAST) oC = (int32_t)r0;
                    }
                }
                uint64_t inj_mask = 0;  // which parents injected on this diagonal (n_parents <= 64 checked by host)
                if (act && hull_has(ih, di)) {
                    for (uint32_t k = 0; k < nd.n_parents; ++k) {
                        const uint32_t* s = e + 5 + 2 * W + k * W;
                        uint32_t nz = 0;
                        for (uint32_t w = 0; w < W; ++w) nz |= s[w];
                        if (nz) inj_mask |= 1ull << k;
                    }
                }
                const bool has = act && (oA >= 0 || oB >= 0 || oC >= 0 || inj_mask != 0);
                int32_t omax = max(max(oA, oB), max(oC, inj_mask ? 0 : -1));
                if (!has) omax = 0;
                // ---- extend the furthest candidate; the others tie iff they match up to its start ----------
                const int64_t pos0 = (int64_t)d + omax;  // >= 0 for real candidates
                uint32_t room = 0;
                if (has) {
                    const uint32_t rn = len - (uint32_t)omax;
                    const uint32_t rr = (pos0 >= 0 && (uint64_t)pos0 < other_len) ? (uint32_t)(other_len - (uint64_t)pos0) : 0u;
                    room = min(rn, rr);
                }
                PT(3);
                bool tA = has && oA == omax, tB = has && oB == omax, tC = has && oC == omax;
                bool tD = has && inj_mask != 0 && omax == 0;
                // candidates behind the furthest one tie with it iff they match the read up to its start
                const bool nA = has && oA >= 0 && oA < omax, nB = has && oB >= 0 && oB < omax, nC = has && oC >= 0 && oC < omax;
                const bool nD = has && inj_mask != 0 && omax > 0;
                // all sequence loads of this step go out together: the extension's first 16 bytes and the first 16
                // bytes of every tie check (their gaps are almost always a base or two, so this settles them)
                const uint8_t* ra = read + (has ? pos0 : 0);
                const Pre16 pm = pre16(nseq + omax, ra, room > 0);
                const Pre16 pA = pre16(nseq + (nA ? oA : 0), read + (nA ? (int64_t)d + oA : 0), nA);
                const Pre16 pB = pre16(nseq + (nB ? oB : 0), read + (nB ? (int64_t)d + oB : 0), nB);
                const Pre16 pC = pre16(nseq + (nC ? oC : 0), read + (nC ? (int64_t)d + oC : 0), nC);
                const Pre16 pD = pre16(nseq, read + (nD ? (int64_t)d : 0), nD);
                uint32_t E;
                {
                    uint32_t n0 = 0;
                    bool done = (room == 0);
                    if (!done) {
                        uint32_t m = pre16_len(pm);
                        if (m > room) m = room;
                        n0 = m;
                        if (m < 16 || n0 >= room) done = true;
                    }
                    E = (uint32_t)omax + match_rest(nseq + omax, ra, room, n0, done);   // wave-collective
                }
                PT(4);
                PC(3, __popcll(__ballot(has)));
                PC(4, __popcll(__ballot(nA)) + __popcll(__ballot(nB)) + __popcll(__ballot(nC)) + __popcll(__ballot(nD)));
                {
                    // a tie check whose gap exceeds 16 bytes and whose first 16 match falls back to the full compare
                    // (match_run is a wave-collective: every lane must call it, no short-circuit around the call)
                    auto tie = [&](const Pre16& pp, bool nX, int32_t oX) -> bool {
                        const uint32_t g = nX ? (uint32_t)(omax - oX) : 0u;
                        bool res = false, pend = false;
                        if (nX) {
                            const uint32_t m = pre16_len(pp);
                            if (g <= 16) res = (m >= g);
                            else pend = (m == 16);
                        }
                        if (__any(pend)) {
                            const uint32_t mr = match_run(nseq + (pend ? oX : 0), read + (pend ? (int64_t)d + oX : 0), pend ? g : 0u);
                            res = res || (pend && mr == g);
                        }
                        return res;
                    };
                    const bool xA = tie(pA, nA, oA), xB = tie(pB, nB, oB), xC = tie(pC, nC, oC), xD = tie(pD, nD, 0);   // no `||`: every lane must take part
                    tA = tA || xA;
                    tB = tB || xB;
                    tC = tC || xC;
                    tD = tD || xD;
                }
                PT(5);
                // ---- decide (wfa_graph.rs:463-474) -----------------------------------------------------------
                const uint64_t pos_end = has ? (uint64_t)((int64_t)d + (int64_t)E) : 0;
                const bool is_final = has && n == last && E == len && pos_end == other_len;
                uint32_t kind = WFA_KIND_NONE;
                bool inject = false;
                if (has) {
                    const bool skip = (E < mf) || (pos_end < min_prog);
                    if (!skip) {
                        e[0] = E;
                        if (pos_end > lane_far) lane_far = (uint32_t)pos_end;
                        if (E == len) {
                            if (n == last) { if (pos_end < other_len) kind = WFA_KIND_END_LAST; }
                            else inject = true;
                        } else {
                            kind = (pos_end < other_len) ? WFA_KIND_INTERIOR_READ : WFA_KIND_INTERIOR;
                        }
                    }
                }
                // ---- write this round's result and the union of the tied sets ---------------------------------
                uint32_t keep[4] = {0u, 0u, 0u, 0u};   // first four words of this diagonal's new set (reused by the hand-off below)
                if (act) {
                    uint32_t* r = e + 1 + c * (2 + W);
                    r[0] = E;
                    r[1] = kind;
                    // four set words at a time: all loads first, then the stores (a store between two loads would
                    // serialise them - the compiler cannot tell that the parities do not alias)
                    const uint32_t* sA = e + ES + 1 + p * (2 + W) + 2;
                    const uint32_t* sB = e + 1 + p * (2 + W) + 2;
                    const uint32_t* sC = e - ES + 1 + p * (2 + W) + 2;
                    for (uint32_t w0 = 0; w0 < W; w0 += 4) {
                        uint32_t bs[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) {
                            const uint32_t w = w0 + j;
                            if (w < W) {
                                if (w0 == 0) {   // fetched with the candidates
                                    if (tA) bs[j] |= qA[j];
                                    if (tB) bs[j] |= qB[j];
                                    if (tC) bs[j] |= qC[j];
                                } else {
                                    if (tA) bs[j] |= sA[w];
                                    if (tB) bs[j] |= sB[w];
                                    if (tC) bs[j] |= sC[w];
                                }
                                if (tD) for (uint32_t k = 0; k < nd.n_parents; ++k) bs[j] |= (e + 5 + 2 * W + k * W)[w];
                            }
                        }
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) {
                            if (w0 + j < W) r[2 + w0 + j] = bs[j];
                            if (w0 == 0) keep[j] = bs[j];
                        }
                    }
                    if (inj_mask) {  // consume the injections (leave the slots zeroed)
                        for (uint32_t k = 0; k < nd.n_parents; ++k) {
                            uint32_t* s = e + 5 + 2 * W + k * W;
                            if ((inj_mask >> k) & 1ull) for (uint32_t w = 0; w < W; ++w) s[w] = 0;
                        }
                    }
                }
                PT(6);
                {
                    const uint64_t vm = __ballot(kind != WFA_KIND_NONE);
                    if (vm) {
                        any_valid = true;
                        vlo = min(vlo, base + (uint32_t)__builtin_ctzll(vm));
                        vhi = max(vhi, base + 63u - (uint32_t)__builtin_clzll(vm));
                    }
                }
                // ---- hand waves that finished this node to its successors, same round (wfa_graph.rs:527-553) ----
                if (__any(inject)) {
                    for (uint32_t j = 0; j < nd.n_children; ++j) {
                        uint32_t cid = 0, ord = 0;
                        { const WfaEdge ed2 = edges[nd.child_off + j]; cid = ed2.child; ord = ed2.ordinal; }   // LDS broadcast
                        cid = wb32(cid);
                        ord = wb32(ord);
                        const WfaNodeU ch = load_node(nodes + cid);
                        const int64_t tdi = (int64_t)d + (int64_t)E - (int64_t)ch.dbase;
                        const bool ok = inject && tdi >= 0 && tdi < (int64_t)ch.width;
                        if (__any(inject && !ok)) band_overflow = true;
                        if (ok) {
                            uint32_t* s = scr + ch.entry_off + (size_t)tdi * ch.entry_stride + 5 + 2 * W + ord * W;
                            const uint32_t* r = e + 1 + c * (2 + W);
#pragma unroll
                            for (uint32_t w = 0; w < 4; ++w)   // still in registers from the union above
                                if (w < W) s[w] = keep[w] | (((cid >> 5) == w) ? 1u << (cid & 31u) : 0u);
                            for (uint32_t w = 4; w < W; ++w) {
                                uint32_t v = r[2 + w];
                                if ((cid >> 5) == w) v |= 1u << (cid & 31u);
                                s[w] = v;
                            }
                        }
                        const uint32_t tlo = wave_min_u32(ok ? (uint32_t)tdi : 0xFFFFu);
                        const uint32_t thi = wave_max_u32(ok ? (uint32_t)tdi : 0u);
                        if (tlo <= thi && tlo != 0xFFFFu) {
                            if (lane == 0) {
                                ns[cid].inj = hull_union(ns[cid].inj, hull_make(tlo, thi));
                                ns[cid].ever = hull_union(ns[cid].ever, hull_make(tlo, thi));
                            }
                            if (cid > amax) amax = cid;
                            if (cid > new_amax) new_amax = cid;
                            if (cid < new_amin) new_amin = cid;
                        }
                    }
                }
                PT(7);
                // ---- finals (wfa_graph.rs:576-629): every wave of the last node that consumed node and read ----
                if (__any(is_final)) {
                    any_final_here = true;
                    for (uint32_t w = 0; w < W; ++w) {
                        const uint32_t v = is_final ? (e + 1 + c * (2 + W))[2 + w] : 0u;
                        const uint32_t o = wave_or_u32(v);
                        if (lane == 0) out_set[w] |= o;
                    }
                }
            }
            if (lane == 0) {
                // the next round pulls only from diagonals that hold a live wave: pruned and stale waves (most of a
                // hull that has grown by one diagonal a round) drop out here instead of costing steps
                ns[n].hull[c] = any_valid ? hull_make(vlo, vhi) : HULL_EMPTY;
                ns[n].stamp[c] = ed;
                ns[n].inj = HULL_EMPTY;
                ns[n].ever = hull_union(ns[n].ever, hull_make(lo, hi));
            }
            if (any_valid) {
                if (n < new_amin) new_amin = n;
                if (n > new_amax) new_amax = n;
            }
            if (any_final_here) final_found = true;
        }
        if (band_overflow) { status = WFA_ST_NEED_BAND; break; }
        if (final_found) { status = WFA_ST_OK; score = ed; break; }
        // end of round (wfa_graph.rs:633-648)
        const uint32_t far = wave_max_u32(lane_far);
        if (far > farthest) farthest = far;
        if (farthest > B.prune_distance) min_prog = farthest - B.prune_distance;
        if ((uint64_t)ed + 1 > B.max_ed) { status = WFA_ST_MAX_ED; score = B.max_ed; break; }
        if (ed + 1 > jd.band) { status = WFA_ST_NEED_BAND; break; }
        if (new_amin == 0xFFFFFFFFu) { status = WFA_ST_INTERNAL; break; }  // no live wave left: cannot happen
        amin = new_amin;
        amax = new_amax;
    }

    PT(8);
    // ---- leave the scratch zeroed for the next job of this slot -----------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (uint32_t n = 0; n < n_nodes; ++n) {
        const uint32_t ev = ns[n].ever;
        if (hull_empty(ev)) continue;
        const WfaNodeU nd = load_node(nodes + n);
        uint32_t* b0 = scr + nd.entry_off + (size_t)hull_lo(ev) * nd.entry_stride;
        const uint32_t cnt = (hull_hi(ev) - hull_lo(ev) + 1) * nd.entry_stride;
        for (uint32_t i = lane; i < cnt; i += 64) b0[i] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    PT(9);
    if (lane == 0) {
        B.status[job] = status;
        B.out_score[job] = score;
    }
#if WFA_PROF
    if (lane == 0 && (job % 1024) == 5)
        printf("job %u nodes %u score %llu total %llu | init %llu skip %llu nodehead %llu gather %llu match %llu tie %llu decide+write %llu inject %llu tail %llu cleanup %llu | rounds %u skipped %u steps %u has-lanes %u tie-lanes %u\n",
               job, n_nodes, (unsigned long long)score, (unsigned long long)(tl - tstart), (unsigned long long)pc[0], (unsigned long long)pc[1], (unsigned long long)pc[2], (unsigned long long)pc[3], (unsigned long long)pc[4], (unsigned long long)pc[5],
               (unsigned long long)pc[6], (unsigned long long)pc[7], (unsigned long long)pc[8], (unsigned long long)pc[9], pn[0], pn[1], pn[2], pn[3], pn[4]);
#endif
}

template <bool BIG> WDEV void wfa_kernel_body(const WfaBatchDev& B) {
    const uint32_t slot = blockIdx.x, G = gridDim.x;
    for (uint32_t round = 0;; ++round) {
        const uint32_t base = round * G;
        if (base >= B.n_items) break;
        const uint32_t i = base + ((round & 1u) ? (G - 1u - slot) : slot);
        if (i < B.n_items) solve_job<BIG>(B, B.order[i], slot);
    }
}
__global__ void __launch_bounds__(64) hp_wfa_kernel(WfaBatchDev B) { wfa_kernel_body<false>(B); }
__global__ void __launch_bounds__(64) hp_wfa_big_kernel(WfaBatchDev B) { wfa_kernel_body<true>(B); }

}  // namespace hp
