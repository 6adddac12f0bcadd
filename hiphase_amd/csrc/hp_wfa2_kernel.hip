// hp_wfa2_kernel.hip — graph-WFA (unit costs, end-to-end) on gfx950, second generation: SEVERAL READS PER WAVEFRONT.
//
// Replaces, bit-identically, reference src/wfa_graph.rs:350-650 `edit_distance_with_pruning` (and, in
// hp_wfa2_build_kernel / hp_wfa2_map_kernel, the graph construction of :119-284 and the node -> allele mapping of
// src/read_parsing.rs:790-800, so that a read never leaves the device between its bases and its allele row).
//
// Why a second kernel: hp_wfa_kernel.hip gives a read a whole wavefront and a dense (node, diagonal) band in HBM;
// on HiFi reads 2-3 of its 64 lanes hold a live wave per step and every step is two HBM round trips. Here
//   * a wavefront is split into 64/G groups of G lanes, one read per group, all groups stepping in lockstep through
//     their own (node, interval-of-diagonals) work items: the vector instructions of a step serve 64/G reads;
//   * a read's wavefront state is COMPACT and lives in LDS (W2Cfg): per round one arena slot per diagonal of each
//     live cluster of diagonals, two rounds deep. The next round pulls (d+1: offset+1, d: offset+1, d-1: offset) from
//     the previous round's slots; waves that finish a node are picked up by its children in the same round through
//     (child, finished entry) pairs (wfa_graph.rs:527-553) - the only global-memory traffic of a step is the sequence
//     bytes themselves and one probe of the capped-diagonal set;
//   * the reference's max_wavefronts map (:360, :464-470) is replaced by the set of CAPPED diagonals, a small tagged
//     hash set per group in HBM: a wave on (node, d) is stale <=> (node, d) once reached cap = min(node length,
//     read length - d) and this wave stops short of it. (A diagonal whose wave is interior with read left gets
//     offset + 1 on itself next round, so it stays ahead of its own record until it is pruned by min_progression -
//     after which any later, shorter wave on it is pruned too - or reaches its cap.) tests/cpp/wfa2_model.cpp pins
//     this formulation against the oracle on the CPU.
// A read that outgrows the compact state (W2_ST_NEED_BIG) is re-run by the dense-band kernel; nothing is approximated.
// Integer/byte work, no MFMA; bound by dependent latency per step, hence the lockstep packing.
#include "hp_common.h"
#include "hp_wfa2_dev.h"

namespace hp {

#define W2DEV __device__ __forceinline__

extern __shared__ __attribute__((aligned(16))) unsigned char w2_smem[];

W2DEV uint32_t w2_lane() { return __lane_id(); }
W2DEV uint64_t w2_ld8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

struct W2Pre16 { uint64_t a0, a1, b0, b1; };
W2DEV W2Pre16 w2_pre16(const uint8_t* a, const uint8_t* b, bool on) {
    W2Pre16 p{0, 0, 0, 0};
    if (on) { p.a0 = w2_ld8(a); p.a1 = w2_ld8(a + 8); p.b0 = w2_ld8(b); p.b1 = w2_ld8(b + 8); }
    return p;
}
W2DEV uint32_t w2_pre16_len(const W2Pre16& p) {
    const uint64_t x0 = p.a0 ^ p.b0, x1 = p.a1 ^ p.b1;
    return x0 ? ((uint32_t)__builtin_ctzll(x0) >> 3) : (x1 ? 8u + ((uint32_t)__builtin_ctzll(x1) >> 3) : 16u);
}

// ---- group collectives (G consecutive lanes; every lane of the group must be active) -------------------------------
template <int G> W2DEV uint64_t w2_gballot(bool pred, uint32_t gbase) {
    const uint64_t b = __ballot(pred);
    if (G == 64) return b;
    return (b >> gbase) & ((1ull << (G & 63)) - 1ull);
}
template <int G> W2DEV int32_t w2_gmax(int32_t v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
template <int G> W2DEV int32_t w2_gmin(int32_t v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
template <int G> W2DEV uint32_t w2_gor(uint32_t v) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) v |= (uint32_t)__shfl_xor((int)v, m);
    return v;
}
W2DEV uint64_t w2_shfl64(uint64_t v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// common prefix of two 32-byte windows (four 8-byte words each)
W2DEV uint32_t w2_cmp32(const uint8_t* a, const uint8_t* b) {
    const uint64_t x0 = w2_ld8(a) ^ w2_ld8(b), x1 = w2_ld8(a + 8) ^ w2_ld8(b + 8);
    const uint64_t x2 = w2_ld8(a + 16) ^ w2_ld8(b + 16), x3 = w2_ld8(a + 24) ^ w2_ld8(b + 24);
    if (x0) return (uint32_t)__builtin_ctzll(x0) >> 3;
    if (x1) return 8u + ((uint32_t)__builtin_ctzll(x1) >> 3);
    if (x2) return 16u + ((uint32_t)__builtin_ctzll(x2) >> 3);
    if (x3) return 24u + ((uint32_t)__builtin_ctzll(x3) >> 3);
    return 32u;
}

// Lanes with done == false have matched their first n bytes of a / b and may match up to maxlen: the group serves
// them one after the other, G x 32 bytes per step. `on`: this group takes part (group-uniform).
template <int G> W2DEV uint32_t w2_match_rest(const uint8_t* a, const uint8_t* b, uint32_t maxlen, uint32_t n, bool done,
                                              bool on, uint32_t gbase, uint32_t gl) {
    bool pending = on && !done;
    while (__any(pending)) {
        const uint64_t gb = w2_gballot<G>(pending, gbase);
        const bool active = gb != 0;
        const int L = active ? __builtin_ctzll(gb) : 0;
        const int src = (int)gbase + L;
        const uint64_t pa = w2_shfl64((uint64_t)(a + n), src), pb = w2_shfl64((uint64_t)(b + n), src);
        const uint32_t rem = (uint32_t)__shfl((int)(maxlen - n), src);
        const uint32_t off = gl * 32u;
        uint32_t m = 32u;
        if (active) {
            if (off < rem) {
                m = w2_cmp32(reinterpret_cast<const uint8_t*>(pa) + off, reinterpret_cast<const uint8_t*>(pb) + off);
                if (m > rem - off) m = rem - off;
            } else m = 0u;   // beyond the end: acts as a stop
        }
        const uint64_t stop = w2_gballot<G>(active && m < 32u, gbase);
        uint32_t got = (uint32_t)G * 32u;
        if (stop) {
            const int S = __builtin_ctzll(stop);
            got = (uint32_t)S * 32u + (uint32_t)__shfl((int)m, (int)gbase + S);
        }
        if (got > rem) got = rem;
        if (active && (int)gl == L) {
            n += got;
            if (stop || n >= maxlen) pending = false;
        }
    }
    return n;
}

template <int W> W2DEV void w2_ldset(const uint32_t* p, uint32_t (&s)[W]) {
#pragma unroll
    for (int w = 0; w < W; ++w) s[w] = p[w];
}

// =====================================================================================================================
template <int G, int W>
__global__ void __launch_bounds__(64) hp_wfa2_kernel(W2Batch B) {
    using C = W2Cfg<W>;
    static_assert(G >= 8 && G <= 64 && (G & (G - 1)) == 0, "group size");
    static_assert(C::MAXS <= G, "one lane per source interval");
    constexpr uint32_t NG = 64 / G;
    const uint32_t lane = w2_lane(), gid = lane / G, gl = lane % G, gbase = gid * G;
    unsigned char* R = w2_smem + (size_t)gid * C::BYTES;
    W2Node* desc = reinterpret_cast<W2Node*>(R + C::O_DESC);
    uint16_t* edg = reinterpret_cast<uint16_t*>(R + C::O_EDGE);
    uint4* live = reinterpret_cast<uint4*>(R + C::O_LIVE);          // [parity * MAXL + i]: node | off << 16, lo, vlo|vhi<<16, flo|fhi<<16 (relative to lo)
    uint32_t* ek = reinterpret_cast<uint32_t*>(R + C::O_EK);        // [parity * SLOTS + s]
    uint32_t* sets = reinterpret_cast<uint32_t*>(R + C::O_SET);     // [(parity * SLOTS + s) * W]
    uint32_t* pairs = reinterpret_cast<uint32_t*>(R + C::O_PAIR);   // child | entry << 16
    uint32_t* pend = reinterpret_cast<uint32_t*>(R + C::O_MISC);
    uint32_t* outset = pend + W;
    // small per-node scratch, overlaid on the tail of the pairs area is NOT safe; keep separate words after outset
    int2* srcs = reinterpret_cast<int2*>(R + C::O_SRC);             // [MAXS] source / item intervals
    uint32_t* parli = reinterpret_cast<uint32_t*>(R + C::O_SRC + 8 * C::MAXS);   // [MAXS] parents' entries

    const uint32_t TG = gridDim.x * NG, slot = blockIdx.x * NG + gid;
    uint64_t* htab = B.htab + ((size_t)slot << B.hcap_log2);
    const uint32_t hmask = (1u << B.hcap_log2) - 1u;
    const uint32_t prune32 = B.prune_distance > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)B.prune_distance;
    const uint32_t maxed32 = B.max_ed > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (uint32_t)B.max_ed;

    // ---- group-uniform state (every lane of a group holds the same value) ------------------------------------------
    enum : uint32_t { S_JOB = 0, S_NODE = 1, S_ITEM = 2, S_TILE = 3, S_DONE = 4 };
    uint32_t state = S_JOB, jround = 0, job = 0;
    uint32_t n_nodes = 0, last = 0, other_len = 0, tag = 0;
    const uint8_t* refp = nullptr; const uint8_t* readp = nullptr;
    const uint8_t* altp = B.seq + B.alt_off;
    uint32_t ed = 0, c = 0, p = 1, lcnt_prev = 0, lcnt_cur = 0, top = 0, pcnt = 0, pp = 0;
    uint32_t farthest = 0, min_prog = 0;
    bool final_found = false;
    int32_t status = W2_ST_PENDING;
    uint32_t score = 0;
    // current node / item
    uint32_t n = 0, len = 0, child_off = 0, n_child = 0, p_first = 0, p_last = 0, npar = 0, n_items = 0, item = 0;
    const uint8_t* nseq = nullptr;
    int32_t lo = 0, hi = 0, base = 0;
    uint32_t coff = 0;
    int32_t clo = 0, chi = INT32_MIN, cvlo = INT32_MAX, cvhi = INT32_MIN, cflo = INT32_MAX, cfhi = INT32_MIN;   // cluster being formed
    uint32_t lane_far = 0;   // per lane

    // publishes the cluster [clo, chi] of the current item as an entry of this round (group-uniform)
    auto emit_cluster = [&]() {
        if (chi == INT32_MIN) return;
        if (lcnt_cur >= (uint32_t)C::MAXL) { status = W2_ST_NEED_BIG; return; }
        const bool fin = cflo <= cfhi;
        if (fin && pcnt + n_child > (uint32_t)C::MAXP) { status = W2_ST_NEED_BIG; return; }
        if (gl == 0) {
            uint4 h;
            h.x = n | ((coff + (uint32_t)(clo - lo)) << 16);
            h.y = (uint32_t)clo;
            h.z = (cvlo <= cvhi) ? ((uint32_t)(cvlo - clo) | ((uint32_t)(cvhi - clo) << 16)) : 0x0000FFFFu;
            h.w = fin ? ((uint32_t)(cflo - clo) | ((uint32_t)(cfhi - clo) << 16)) : 0x0000FFFFu;
            live[c * C::MAXL + lcnt_cur] = h;
            if (fin)
                for (uint32_t j = 0; j < n_child; ++j) {
                    const uint32_t cid = edg[child_off + j];
                    pairs[pcnt + j] = cid | (lcnt_cur << 16);
                    pend[cid >> 5] |= 1u << (cid & 31u);
                }
        }
        if (fin) pcnt += n_child;
        lcnt_cur++;
        chi = INT32_MIN; cvlo = INT32_MAX; cvhi = INT32_MIN; cflo = INT32_MAX; cfhi = INT32_MIN;
    };

    uint32_t steps = 0;      // tiles of the current job (a watchdog: no input needs anywhere near W2_MAX_STEPS)
    for (;;) {
        // ============================ 1. control: advance every group to its next tile ===============================
        uint32_t spins = 0;
        while (state != S_TILE && state != S_DONE) {
            if (++spins > (1u << 20)) { state = S_DONE; break; }   // cannot happen; never hang the device
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (state == S_JOB) {
                // results of the job that just ended
                if (status != W2_ST_PENDING) {
                    if (gl == 0) { B.status[job] = status; B.out_score[job] = score; }
                    if (gl < (uint32_t)W) B.out_sets[(size_t)job * W2_SET_STRIDE + gl] = outset[gl];
                    status = W2_ST_PENDING;
                }
                const uint32_t k = jround * TG + ((jround & 1u) ? (TG - 1u - slot) : slot);
                jround++;
                if (k >= B.n_items) { state = S_DONE; break; }
                job = B.order[k];
                const W2Job jd = B.jobs[job];
                const W2Info ji = B.info[job];
                n_nodes = ji.n_nodes; last = n_nodes - 1u; other_len = jd.read_len;
                refp = B.seq + jd.ref_off; readp = B.seq + jd.read_off;
                tag = B.tag_base + job + 1u;
                if (n_nodes == 0 || n_nodes > (uint32_t)C::MAXN || ji.n_edges > (uint32_t)C::MAXE || other_len >= (uint32_t)W2_DIAG_LIM) {
                    status = W2_ST_NEED_BIG; score = 0;
                    if (gl < (uint32_t)W) outset[gl] = 0u;
                    continue;   // stays in S_JOB: the next pass writes this status and fetches the next job
                }
                {
                    const uint32_t* gd = reinterpret_cast<const uint32_t*>(B.nodes + jd.node_off);
                    uint32_t* ld = reinterpret_cast<uint32_t*>(desc);
                    for (uint32_t i = gl; i < n_nodes * 3u; i += G) ld[i] = gd[i];
                    const uint16_t* ge = B.edges + jd.edge_off;
                    for (uint32_t i = gl; i < ji.n_edges; i += G) edg[i] = ge[i];
                    if (gl < (uint32_t)W) { pend[gl] = gl == 0 ? 1u : 0u; outset[gl] = 0u; }   // start wave: node 0 pending in round 0
                }
                ed = 0; c = 0; p = 1; lcnt_prev = 0; lcnt_cur = 0; top = 0; pcnt = 0; pp = 0; steps = 0;
                farthest = 0; min_prog = 0; final_found = false; lane_far = 0; score = 0;
                state = S_NODE;
                continue;
            }
            if (state == S_NODE) {
                // next node of this round: the smaller of the previous round's next entry and the first pending child
                uint32_t a = 0xFFFFu;
                while (pp < lcnt_prev) {
                    const uint4 h = live[p * C::MAXL + pp];
                    if ((h.z & 0xFFFFu) <= (h.z >> 16)) { a = h.x & 0xFFFFu; break; }
                    ++pp;   // an entry that only held finished waves
                }
                uint32_t bq = 0xFFFFu;
#pragma unroll
                for (int w = W - 1; w >= 0; --w) { const uint32_t v = pend[w]; if (v) bq = (uint32_t)w * 32u + (uint32_t)__builtin_ctz(v); }
                n = min(a, bq);
                if (n == 0xFFFFu) {
                    // ---- end of round (wfa_graph.rs:633-648) ----
                    const uint32_t far = (uint32_t)w2_gmax<G>((int32_t)lane_far);
                    lane_far = 0;
                    if (final_found) { status = W2_ST_OK; score = ed; state = S_JOB; continue; }
                    if (far > farthest) farthest = far;
                    if (farthest > prune32) min_prog = farthest - prune32;
                    if (ed + 1u > maxed32) { status = W2_ST_MAX_ED; score = maxed32; state = S_JOB; continue; }
                    bool any_live = false;
                    for (uint32_t i = 0; i < lcnt_cur; ++i) { const uint4 h = live[c * C::MAXL + i]; any_live = any_live || ((h.z & 0xFFFFu) <= (h.z >> 16)); }
                    if (!any_live) { status = W2_ST_INTERNAL; state = S_JOB; continue; }
                    ++ed; p = c; c ^= 1u; lcnt_prev = lcnt_cur; lcnt_cur = 0; top = 0; pcnt = 0; pp = 0;
                    continue;
                }
                if (bq == n && gl == 0) pend[n >> 5] &= ~(1u << (n & 31u));
                {
                    const W2Node nd = desc[n];
                    len = nd.len_ref & ~W2_IS_REF;
                    nseq = ((nd.len_ref & W2_IS_REF) ? refp : altp) + nd.seq_off;
                    child_off = nd.child & 0xFFFFu; n_child = nd.child >> 16;
                }
                // ---- source intervals: previous entries of n (grown by one diagonal a side), finished parents, start ----
                uint32_t ns = 0;
                bool over = false;
                p_first = pp;
                while (pp < lcnt_prev) {
                    const uint4 h = live[p * C::MAXL + pp];
                    if ((h.x & 0xFFFFu) != n) break;
                    if ((h.z & 0xFFFFu) <= (h.z >> 16)) {
                        if (ns < (uint32_t)C::MAXS) { if (gl == 0) srcs[ns] = make_int2((int32_t)h.y + (int32_t)(h.z & 0xFFFFu) - 1, (int32_t)h.y + (int32_t)(h.z >> 16) + 1); }
                        else over = true;
                        ++ns;
                    }
                    ++pp;
                }
                p_last = pp;
                npar = 0;
                for (uint32_t i = 0; i < pcnt; ++i) {
                    const uint32_t pr = pairs[i];
                    if ((pr & 0xFFFFu) != n) continue;
                    const uint4 h = live[c * C::MAXL + (pr >> 16)];
                    const int32_t pl = (int32_t)(desc[h.x & 0xFFFFu].len_ref & ~W2_IS_REF);
                    if (ns < (uint32_t)C::MAXS) {
                        if (gl == 0) { srcs[ns] = make_int2((int32_t)h.y + (int32_t)(h.w & 0xFFFFu) + pl, (int32_t)h.y + (int32_t)(h.w >> 16) + pl); parli[npar] = pr >> 16; }
                    } else over = true;
                    ++ns; ++npar;
                }
                if (ed == 0 && n == 0) {
                    if (ns < (uint32_t)C::MAXS) { if (gl == 0) srcs[ns] = make_int2(0, 0); } else over = true;
                    ++ns;
                }
                if (over) { status = W2_ST_NEED_BIG; state = S_JOB; continue; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // ---- merge overlapping / touching intervals: lane i owns source i ----
                n_items = ns;
                if (ns > 1) {
                    int2 iv = gl < ns ? srcs[gl] : make_int2(INT32_MAX, INT32_MIN);
                    for (uint32_t it = 1; it < ns; ++it)
                        for (uint32_t j = 0; j < ns; ++j) {
                            const int32_t lj = __shfl(iv.x, (int)(gbase + j)), hj = __shfl(iv.y, (int)(gbase + j));
                            if (gl < ns && lj <= iv.y + 1 && iv.x <= hj + 1) { iv.x = min(iv.x, lj); iv.y = max(iv.y, hj); }
                        }
                    bool leader = gl < ns;
                    for (uint32_t j = 0; j + 1 < ns; ++j) {
                        const int32_t lj = __shfl(iv.x, (int)(gbase + j)), hj = __shfl(iv.y, (int)(gbase + j));
                        if (gl > j && gl < ns && lj == iv.x && hj == iv.y) leader = false;
                    }
                    const uint64_t lm = w2_gballot<G>(leader, gbase);
                    n_items = (uint32_t)__popcll(lm);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (leader) srcs[__popcll(lm & ((1ull << gl) - 1ull))] = iv;
                }
                item = 0;
                state = S_ITEM;
                continue;
            }
            // state == S_ITEM
            if (item >= n_items) { state = S_NODE; continue; }
            {
                const int2 iv = srcs[item];
                ++item;
                lo = iv.x; hi = iv.y;
                const uint32_t cnt = (uint32_t)(hi - lo + 1);
                if (top + cnt > (uint32_t)C::SLOTS || lo <= -W2_DIAG_LIM || hi >= W2_DIAG_LIM) { status = W2_ST_NEED_BIG; state = S_JOB; continue; }
                coff = top; top += cnt; base = lo;
                chi = INT32_MIN; cvlo = INT32_MAX; cvhi = INT32_MIN; cflo = INT32_MAX; cfhi = INT32_MIN;
                state = S_TILE;
            }
        }
        if (!__any(state == S_TILE)) break;   // every group is done
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // ============================ 2. one tile: G diagonals of the current item ===================================
        const bool run = state == S_TILE;
        const int32_t d = base + (int32_t)gl;
        const bool act = run && d <= hi;
        // ---- candidates from the previous round: A from d+1 (offset+1), B from d (offset+1), C from d-1 (offset) ----
        int32_t oA = -1, oB = -1, oC = -1;
        uint32_t qA[W], qB[W], qC[W], qD[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { qA[w] = 0; qB[w] = 0; qC[w] = 0; qD[w] = 0; }
        if (run) {
            for (uint32_t k = p_first; k < p_last; ++k) {
                const uint4 h = live[p * C::MAXL + k];
                const int32_t plo = (int32_t)h.y, pvlo = plo + (int32_t)(h.z & 0xFFFFu), pvhi = plo + (int32_t)(h.z >> 16);
                const uint32_t poff = p * C::SLOTS + (h.x >> 16);
                if (act && d + 1 >= pvlo && d + 1 <= pvhi) {
                    const uint32_t s = poff + (uint32_t)(d + 1 - plo), e = ek[s];
                    if (e & 1u) { oA = (int32_t)(e >> 3) + 1; w2_ldset<W>(sets + (size_t)s * W, qA); }
                }
                if (act && d >= pvlo && d <= pvhi) {
                    const uint32_t s = poff + (uint32_t)(d - plo), e = ek[s];
                    if ((e & 7u) == W2_KIND_INTERIOR_READ) { oB = (int32_t)(e >> 3) + 1; w2_ldset<W>(sets + (size_t)s * W, qB); }
                }
                if (act && d - 1 >= pvlo && d - 1 <= pvhi) {
                    const uint32_t s = poff + (uint32_t)(d - 1 - plo), e = ek[s], kk = e & 7u;
                    if (kk == W2_KIND_INTERIOR_READ || kk == W2_KIND_END_LAST) { oC = (int32_t)(e >> 3); w2_ldset<W>(sets + (size_t)s * W, qC); }
                }
            }
        }
        // ---- waves that finished a parent THIS round (offset 0; wfa_graph.rs:527-553) ----
        bool hinj = false;
        if (run) {
            for (uint32_t i = 0; i < npar; ++i) {
                const uint4 h = live[c * C::MAXL + parli[i]];
                const int32_t pl = (int32_t)(desc[h.x & 0xFFFFu].len_ref & ~W2_IS_REF);
                const int32_t dd = d - pl - (int32_t)h.y;   // relative to the parent entry's first diagonal
                if (act && dd >= (int32_t)(h.w & 0xFFFFu) && dd <= (int32_t)(h.w >> 16)) {
                    const uint32_t s = c * C::SLOTS + (h.x >> 16) + (uint32_t)dd;
                    if ((ek[s] & 7u) == W2_KIND_FINISHED) {
                        hinj = true;
                        uint32_t t[W];
                        w2_ldset<W>(sets + (size_t)s * W, t);
#pragma unroll
                        for (int w = 0; w < W; ++w) qD[w] |= t[w];
                    }
                }
            }
            if (act && ed == 0 && n == 0 && d == 0) hinj = true;   // the start wave (wfa_graph.rs:366-378)
        }
        if (hinj) {
#pragma unroll
            for (int w = 0; w < W; ++w) if ((n >> 5) == (uint32_t)w) qD[w] |= 1u << (n & 31u);   // best + the successor (:535-541)
        }
        const bool has = act && (oA >= 0 || oB >= 0 || oC >= 0 || hinj);
        int32_t omax = max(max(oA, oB), max(oC, hinj ? 0 : -1));
        if (!has) omax = 0;
        // ---- extend the furthest candidate; the others tie iff they match the read up to its start -----------------
        const int32_t pos0 = d + omax;   // >= 0 for real candidates
        uint32_t room = 0;
        if (has) {
            const uint32_t rn = len - (uint32_t)omax;
            const uint32_t rr = (pos0 >= 0 && (uint32_t)pos0 < other_len) ? other_len - (uint32_t)pos0 : 0u;
            room = min(rn, rr);
        }
        bool tA = has && oA == omax, tB = has && oB == omax, tC = has && oC == omax;
        bool tD = has && hinj && omax == 0;
        const bool nA = has && oA >= 0 && oA < omax, nB = has && oB >= 0 && oB < omax, nC = has && oC >= 0 && oC < omax;
        const bool nD = has && hinj && omax > 0;
        // every global load of the step goes out together: the extension's first 16 bytes, the first 16 bytes of every
        // tie check, and the probe of the capped-diagonal set
        const uint8_t* ra = readp + (has ? pos0 : 0);
        const uint8_t* na = nseq + (has ? omax : 0);
        const uint64_t key = ((uint64_t)tag << 32) | ((uint64_t)(n & 0x3FFu) << 18) | (uint64_t)((uint32_t)d & 0x3FFFFu);
        uint32_t hpos = (n * 0x9E3779B1u + (uint32_t)d) & hmask;
        uint64_t he = 0;
        if (has) he = htab[hpos];
        const W2Pre16 pm = w2_pre16(na, ra, room > 0);
        const W2Pre16 pA = w2_pre16(nseq + (nA ? oA : 0), readp + (nA ? d + oA : 0), nA);
        const W2Pre16 pB = w2_pre16(nseq + (nB ? oB : 0), readp + (nB ? d + oB : 0), nB);
        const W2Pre16 pC = w2_pre16(nseq + (nC ? oC : 0), readp + (nC ? d + oC : 0), nC);
        const W2Pre16 pD = w2_pre16(nseq, readp + (nD ? d : 0), nD);
        uint32_t E;
        {
            uint32_t n0 = 0;
            bool done = (room == 0);
            if (!done) {
                uint32_t m = w2_pre16_len(pm);
                if (m > room) m = room;
                n0 = m;
                if (m < 16 || n0 >= room) done = true;
            }
            E = (uint32_t)omax + w2_match_rest<G>(na, ra, room, n0, done, run, gbase, gl);
        }
        {
            auto tie = [&](const W2Pre16& pp16, bool nX, int32_t oX) -> bool {
                const uint32_t g = nX ? (uint32_t)(omax - oX) : 0u;
                bool res = false, pend16 = false;
                if (nX) {
                    const uint32_t m = w2_pre16_len(pp16);
                    if (g <= 16) res = (m >= g);
                    else pend16 = (m == 16);
                }
                if (__any(pend16)) {   // wave-uniform: every lane takes part
                    const uint32_t mr = w2_match_rest<G>(nseq + (pend16 ? oX : 0), readp + (pend16 ? d + oX : 0), pend16 ? g : 0u, pend16 ? 16u : 0u, !pend16, run, gbase, gl);
                    res = res || (pend16 && mr == g);
                }
                return res;
            };
            const bool xA = tie(pA, nA, oA), xB = tie(pB, nB, oB), xC = tie(pC, nC, oC), xD = tie(pD, nD, 0);   // no `||`: collectives inside
            tA = tA || xA; tB = tB || xB; tC = tC || xC; tD = tD || xD;
        }
        // ---- capped-diagonal set: is (n, d) recorded? (linear probing past other keys of this job; rare) -------------
        bool capped = false, hfull = false;
        if (has) {
            uint32_t probes = 0;
            while (he != key && (uint32_t)(he >> 32) == tag) {
                if (++probes > 24u) { hfull = true; break; }
                hpos = (hpos + 1u) & hmask;
                he = htab[hpos];
            }
            capped = he == key;
        }
        // ---- decide (wfa_graph.rs:463-474) ---------------------------------------------------------------------------
        const int32_t pos_end = has ? d + (int32_t)E : 0;
        const int32_t cap = min((int32_t)len, (int32_t)other_len - d);
        const bool is_final = has && n == last && E == len && (uint32_t)pos_end == other_len;
        uint32_t kind = W2_KIND_NONE;
        bool ins = false;
        if (has) {
            const bool skip = (capped && (int32_t)E < cap) || ((uint32_t)pos_end < min_prog);
            if (!skip) {
                if ((uint32_t)pos_end > lane_far) lane_far = (uint32_t)pos_end;
                ins = (int32_t)E == cap && !capped;
                if (E == len) {
                    if (n == last) { if ((uint32_t)pos_end < other_len) kind = W2_KIND_END_LAST; }
                    else kind = W2_KIND_FINISHED;
                } else kind = ((uint32_t)pos_end < other_len) ? W2_KIND_INTERIOR_READ : W2_KIND_INTERIOR;
            }
        }
        // ---- record newly capped diagonals. A lane whose probe ended on its own home slot stores there; displaced
        // lanes (rare) go one at a time and probe again, so that two of them never take the same empty slot -----------
        {
            const uint32_t home = (n * 0x9E3779B1u + (uint32_t)d) & hmask;
            const bool direct = ins && hpos == home && !hfull;
            if (direct) htab[hpos] = key;
            bool later = ins && !direct && !hfull;
            while (__any(later)) {
                const uint64_t lm = __ballot(later);
                const int L = __builtin_ctzll(lm);
                if ((int)lane == L) {
                    uint32_t hp = home, probes = 0;
                    uint64_t e = htab[hp];
                    while (e != key && (uint32_t)(e >> 32) == tag) {
                        if (++probes > 24u) { hfull = true; break; }
                        hp = (hp + 1u) & hmask;
                        e = htab[hp];
                    }
                    if (!hfull) htab[hp] = key;
                    later = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        if (__any(hfull)) { if (w2_gballot<G>(hfull, gbase)) status = W2_ST_NEED_BIG; }
        // ---- write this round's slot: offset | kind and the union of the tied sets -----------------------------------
        uint32_t best[W];
#pragma unroll
        for (int w = 0; w < W; ++w) best[w] = (tA ? qA[w] : 0u) | (tB ? qB[w] : 0u) | (tC ? qC[w] : 0u) | (tD ? qD[w] : 0u);
        if (act) {
            const uint32_t s = c * C::SLOTS + coff + (uint32_t)(d - lo);
            ek[s] = has ? ((E << 3) | kind) : 0u;
            if (kind != W2_KIND_NONE) {
#pragma unroll
                for (int w = 0; w < W; ++w) sets[(size_t)s * W + w] = best[w];
            }
        }
        // ---- finals (wfa_graph.rs:576-629): every wave of the last node that consumed node and read -------------------
        if (__any(is_final)) {
            const bool gf = w2_gballot<G>(is_final, gbase) != 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t o = w2_gor<G>(is_final ? best[w] : 0u);
                if (gf && gl == 0) outset[w] |= o;
            }
            if (gf) final_found = true;
        }
        // ---- clusters of non-empty diagonals become this round's entries ----------------------------------------------
        if (run) {
            const uint64_t lm = w2_gballot<G>(kind == W2_KIND_INTERIOR || kind == W2_KIND_INTERIOR_READ || kind == W2_KIND_END_LAST, gbase);
            const uint64_t fm = w2_gballot<G>(kind == W2_KIND_FINISHED, gbase);
            uint64_t any = lm | fm;
            while (any) {
                const int bpos = __builtin_ctzll(any);
                any &= any - 1;
                const int32_t dd = base + bpos;
                if (chi != INT32_MIN && dd - chi >= 3) emit_cluster();
                if (chi == INT32_MIN) clo = dd;
                chi = dd;
                if ((fm >> bpos) & 1ull) { cflo = min(cflo, dd); cfhi = max(cfhi, dd); }
                else { cvlo = min(cvlo, dd); cvhi = max(cvhi, dd); }
            }
            base += (int32_t)G;
            if (base > hi) { emit_cluster(); state = S_ITEM; }
            if (++steps > W2_MAX_STEPS) status = W2_ST_INTERNAL;
            if (status != W2_ST_PENDING) state = S_JOB;
        }
    }
}

// ---- graph construction: one thread per job (wfa_graph.rs:119-284 via w2_build) ------------------------------------
struct W2BuildArgs {
    const W2Job* jobs;
    uint32_t n_jobs;
    const W2Variant* vars;
    W2Node* nodes;
    uint16_t* edges;
    uint32_t* tags;
    uint16_t* par;      // [sum edge_cap] scratch
    uint32_t* poff;     // [sum node_cap + n_jobs] scratch (job j: at node_off + j)
    uint32_t* cnt;      // [sum node_cap] scratch
    W2Info* info;
};
__global__ void __launch_bounds__(64) hp_wfa2_build_kernel(W2BuildArgs A) {
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    if (j >= A.n_jobs) return;
    const W2Job J = A.jobs[j];
    w2_build(J, A.vars, A.nodes + J.node_off, A.edges + J.edge_off, A.tags + J.tag_off, A.par + J.edge_off,
             A.poff + J.node_off + j, A.cnt + J.node_off, A.info + j);
}

// ---- traversed nodes -> per-het AlleleType (read_parsing.rs:790-800): one thread per job ---------------------------
struct W2MapArgs {
    const W2Job* jobs;
    const W2Info* info;
    uint32_t n_jobs;
    const uint32_t* tags;
    const uint32_t* out_sets;
    const int32_t* status;
    uint8_t* alleles;
};
__global__ void __launch_bounds__(64) hp_wfa2_map_kernel(W2MapArgs A) {
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    if (j >= A.n_jobs) return;
    const W2Job J = A.jobs[j];
    w2_map_alleles(A.tags + J.tag_off, A.info[j].n_tags, A.out_sets + (size_t)j * W2_SET_STRIDE, A.status[j] == W2_ST_OK,
                   A.alleles + J.allele_off, J.n_hets);
}

}  // namespace hp
