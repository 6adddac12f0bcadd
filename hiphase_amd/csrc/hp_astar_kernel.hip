// hp_astar_kernel.hip — the A* MEC phasing solver for gfx950 (MI355X), one wavefront per phase block.
//
// Replaces, bit-identically, reference src/astar_phaser.rs:
//   new_extended_node (:69-119)  -> score_children(): lanes = rows covering the new variant; each lane scores
//                                   its row against both haplotype windows with bit-sliced AND/popcount
//                                   (v_bcnt_u32_b32) over 32-variant plane words, then wave reductions.
//   PriorityQueue (:316,460)     -> 64-way sharded binary heap (one private heap per lane, wave arg-min over
//                                   the 64 tops); the priority is a total order so any heap gives the same pops.
//   astar_subsolver (:311-405), calculate_astar_heuristic (:246-292), astar_solver (:426-633)
//                                -> subsolve()/solve_block(), same statement order, same `<`/`<=`.
// Integer-only (u32 per-row scores, u64 costs); no MFMA (there is no dense contraction on this path).
// Node state is O(1): a chain of 32-variant haplotype windows (see hp_astar_dev.h), not O(len) copies.
//
// Memory-ordering rule used throughout: every global/LDS location is written and later read by the SAME
// lane (lane 0 for node records / H / tracker, the owning lane for its private heap), then broadcast with
// v_readfirstlane / v_readlane — so no cross-lane visibility fences are needed inside the wave.
#include "hp_astar_dev.h"
#include "hp_common.h"

namespace hp {

#define DEVINL __device__ __forceinline__

DEVINL bool key_less(const Key& a, const Key& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
DEVINL Key key_inf() { return Key{~0ull, ~0ull}; }
DEVINL bool key_is_inf(const Key& k) { return (k.hi & k.lo) == ~0ull; }

DEVINL uint32_t bcast32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DEVINL uint64_t bcast64(uint64_t v) {
    uint32_t lo = bcast32((uint32_t)v), hi = bcast32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
DEVINL uint32_t lane_id() { return __lane_id(); }

DEVINL uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
DEVINL uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
DEVINL Key wave_min_key(Key k) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Key o;
        o.hi = __shfl_xor(k.hi, m);
        o.lo = __shfl_xor(k.lo, m);
        if (key_less(o, k)) k = o;
    }
    return k;
}

// ---- node record I/O: lane 0 writes, lane 0 reads, broadcast ------------------------------------
DEVINL void store_rec(NodeRec* dst, const NodeRec& r) {
    if (lane_id() == 0) {
        uint4* d = reinterpret_cast<uint4*>(dst);
        d[0] = make_uint4((uint32_t)r.frozen, (uint32_t)(r.frozen >> 32), r.depth, r.hets);
        d[1] = make_uint4(r.anc1, r.anc2, r.w0.h1, r.w0.h2);
        d[2] = make_uint4(r.w0.nv, r.w1.h1, r.w1.h2, r.w1.nv);
    }
}
DEVINL NodeRec load_rec(const NodeRec* src) {
    uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a;
    if (lane_id() == 0) {
        const uint4* s = reinterpret_cast<const uint4*>(src);
        a = s[0];
        b = s[1];
        c = s[2];
    }
    NodeRec r;
    r.frozen = ((uint64_t)bcast32(a.y) << 32) | bcast32(a.x);
    r.depth = bcast32(a.z);
    r.hets = bcast32(a.w);
    r.anc1 = bcast32(b.x);
    r.anc2 = bcast32(b.y);
    r.w0.h1 = bcast32(b.z);
    r.w0.h2 = bcast32(b.w);
    r.w0.nv = bcast32(c.x);
    r.w1.h1 = bcast32(c.y);
    r.w1.h2 = bcast32(c.z);
    r.w1.nv = bcast32(c.w);
    return r;
}

// ---- 64-way sharded heap: lane l owns elements base[j*64 + l] --------------------------------------
struct Heap {
    Key* base;          // uniform
    uint32_t jcap;      // uniform: per-lane capacity
    uint32_t cnt;       // per lane
    Key top;            // uniform cache of the global minimum (inf when empty)
    uint32_t top_lane;  // uniform
    uint32_t ovf;       // per lane
};

DEVINL void heap_reset(Heap& h) {
    h.cnt = 0;
    h.top = key_inf();
    h.top_lane = 0;
}
DEVINL void lane_sift_up(Key* hp, uint32_t j, Key k) {
    while (j > 0) {
        uint32_t pj = (j - 1) >> 1;
        Key pk = hp[(size_t)pj * 64];
        if (key_less(k, pk)) {
            hp[(size_t)j * 64] = pk;
            j = pj;
        } else
            break;
    }
    hp[(size_t)j * 64] = k;
}
DEVINL void lane_sift_down(Key* hp, uint32_t i, uint32_t n, Key k) {
    for (;;) {
        uint32_t c = 2 * i + 1;
        if (c >= n) break;
        Key ck = hp[(size_t)c * 64];
        if (c + 1 < n) {
            Key c2 = hp[(size_t)(c + 1) * 64];
            if (key_less(c2, ck)) {
                ck = c2;
                c = c + 1;
            }
        }
        if (key_less(ck, k)) {
            hp[(size_t)i * 64] = ck;
            i = c;
        } else
            break;
    }
    hp[(size_t)i * 64] = k;
}
DEVINL void heap_recompute_top(Heap& h) {
    Key mine = key_inf();
    if (h.cnt > 0) mine = h.base[lane_id()];
    Key m = wave_min_key(mine);
    h.top.hi = bcast64(m.hi);
    h.top.lo = bcast64(m.lo);
    uint64_t who = __ballot(h.cnt > 0 && mine.hi == m.hi && mine.lo == m.lo);
    h.top_lane = who ? (uint32_t)__builtin_ctzll(who) : 0;
}
// uniform key; the lane (node_index % 64) inserts it into its private heap
DEVINL void heap_push(Heap& h, Key k) {
    uint32_t tgt = (uint32_t)(k.lo >> 24) & 63u;
    if (lane_id() == tgt) {
        if (h.cnt >= h.jcap) {
            h.ovf = 1;
        } else {
            lane_sift_up(h.base + lane_id(), h.cnt, k);
            h.cnt += 1;
        }
    }
    if (key_less(k, h.top)) {
        h.top = k;
        h.top_lane = tgt;
    }
}
// removes the global minimum (h.top); caller copied it first
DEVINL void heap_pop(Heap& h) {
    if (lane_id() == h.top_lane) {
        h.cnt -= 1;
        if (h.cnt > 0) {
            Key last = h.base[(size_t)h.cnt * 64 + lane_id()];
            lane_sift_down(h.base + lane_id(), 0, h.cnt, last);
        }
    }
    heap_recompute_top(h);
}
DEVINL bool heap_empty(const Heap& h) { return key_is_inf(h.top); }

// ---- search node held in (uniform) registers -------------------------------------------------------
struct Cur {
    uint64_t frozen, total, idx;
    uint32_t depth, hets, anc1, anc2;
    Win w0, w1;
};

DEVINL Key make_key(uint64_t total, uint32_t hets, uint64_t idx, uint32_t depth) {
    Key k;
    k.hi = (total << 24) | (uint64_t)(0xFFFFFFu - hets);
    k.lo = (idx << 24) | (uint64_t)depth;
    return k;
}
DEVINL Win fresh_win() { return Win{0u, 0u, 0xFFFFFFFFu}; }

struct Ctx {
    const uint32_t *vlo, *vhi;
    const uint8_t* vflags;
    const uint32_t *rstart, *rend, *rword;
    const uint32_t* words;
    uint32_t N;
    // per-lane work counters
    uint64_t evals, cells;
};

struct Children {
    int n;
    uint32_t a1[4], a2[4];
    uint64_t frozen[4], total[4];
    uint32_t hets[4];
    Win w0[4];
    // shared by all children
    uint32_t depth, anc1, anc2;
    Win w1;
};

// weighted popcount: sum_b popc(M & Q_b) << b   (Horner over the 8 quality bit-planes)
DEVINL uint32_t wpop(uint32_t M, const uint32_t* Q) {
    uint32_t s = 0;
#pragma unroll
    for (int b = 7; b >= 0; --b) s = (s << 1) + (uint32_t)__popc(M & Q[b]);
    return s;
}

// new_extended_node (astar_phaser.rs:69-119) for all children of `cur` at once.
// Row r covering p contributes min(score(h1'), score(h2')) where h' = parent prefix + child allele:
//   score(h') = S(parent prefix over [max(start_r, off), p)) + (allele_r[p] != a ? qual_r[p] : 0)
// so the O(overlap) part is shared by the children; it is evaluated bit-parallel per 32-variant word.
DEVINL void score_children(Ctx& cx, const Cur& cur, uint32_t off, uint32_t p, const NodeRec* pool, Children& ch,
                           uint32_t (&sumF)[4], uint32_t (&sumL)[4]) {
    const uint32_t lane = lane_id();
    const uint32_t kp = p >> 5, bp = p & 31u;
    const uint32_t ck = cur.depth ? ((off + cur.depth - 1) >> 5) : (off >> 5);
    const bool trans = (ck != kp);  // the child opens a new 32-variant chunk
    const Win W0 = trans ? fresh_win() : cur.w0;
    const Win W1 = trans ? cur.w0 : cur.w1;
    const Win W2 = cur.w1;  // only meaningful when trans
    uint32_t accF[4] = {0, 0, 0, 0}, accL[4] = {0, 0, 0, 0};
    const uint32_t lo = cx.vlo[p], hi = cx.vhi[p];

    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t r = base + lane;
        bool valid = r < hi;
        uint32_t rs = 0, re = 0, rw = 0;
        if (valid) {
            rs = cx.rstart[r];
            re = cx.rend[r];
            rw = cx.rword[r];
        }
        valid = valid && re > p;  // start <= p by construction of vhi
        const uint32_t kr = rs >> 5;
        const uint32_t myj = kp - kr;  // words before the one holding p (garbage when !valid)
        uint32_t s1 = 0, s2 = 0, ap = 3, qp = 0;
        // haplotype-chain walker state (uniform)
        uint32_t chain_slot = cur.anc2, chain_phase = 0, chain_next = NONE32, guard = 0;
        Win cw1 = fresh_win();
        for (uint32_t j = 0;; ++j) {
            const bool need = valid && (j <= myj);
            if (!__any(need)) break;
            Win w;
            if (j == 0) w = W0;
            else if (j == 1) w = W1;
            else if (trans && j == 2) w = W2;
            else {
                if (chain_phase == 0) {
                    if (chain_slot == NONE32 || ++guard > (cx.N >> 6) + 4) break;  // nothing older: zero cost
                    NodeRec a = load_rec(pool + chain_slot);
                    w = a.w0;
                    cw1 = a.w1;
                    chain_next = a.anc2;
                    chain_phase = 1;
                } else {
                    w = cw1;
                    chain_slot = chain_next;
                    chain_phase = 0;
                }
            }
            if (need) {
                const uint4* pw = reinterpret_cast<const uint4*>(cx.words + (size_t)(rw + (kp - j - kr)) * WORD_DWORDS);
                const uint4 x0 = pw[0], x1 = pw[1], x2 = pw[2];
                const uint32_t aLo = x0.x, aHi = x0.y;
                const uint32_t Q[8] = {x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y};
                const uint32_t M1 = ~w.nv & ((aLo ^ w.h1) | aHi);
                const uint32_t M2 = ~w.nv & ((aLo ^ w.h2) | aHi);
                s1 += wpop(M1, Q);
                s2 += wpop(M2, Q);
                if (j == 0) {
                    ap = ((aLo >> bp) & 1u) | (((aHi >> bp) & 1u) << 1);
                    qp = 0;
#pragma unroll
                    for (int b = 7; b >= 0; --b) qp = (qp << 1) | ((Q[b] >> bp) & 1u);
                }
            }
        }
        if (valid) {
            const bool frozen = (re == p + 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < ch.n) {
                    const uint32_t m1 = (ch.a1[c] < 2 && ap != ch.a1[c]) ? qp : 0u;
                    const uint32_t m2 = (ch.a2[c] < 2 && ap != ch.a2[c]) ? qp : 0u;
                    const uint32_t cost = min(s1 + m1, s2 + m2);
                    if (frozen) accF[c] += cost; else accL[c] += cost;
                }
            }
            cx.evals += (uint64_t)ch.n;
            cx.cells += (uint64_t)ch.n * (uint64_t)(p + 1 - max(rs, off));
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sumF[c] = bcast32(wave_sum_u32(accF[c]));
        sumL[c] = bcast32(wave_sum_u32(accL[c]));
    }
}

// Builds the children of `cur` (astar_phaser.rs:348-392 / :517-561): order (0,1),(1,0),(0,0),(1,1);
// (1,0) skipped iff the parent haplotypes are identical (<=> hets == 0); an ignored variant yields one (2,2).
DEVINL void expand(Ctx& cx, const Cur& cur, uint32_t off, uint32_t p, uint64_t h_next, NodeRec* pool, Children& ch) {
    const bool bad = (cx.vflags[p] & HP_VAR_IGNORED) != 0;
    ch.n = 0;
    if (bad) {
        ch.a1[0] = 2; ch.a2[0] = 2; ch.n = 1;
    } else {
        ch.a1[ch.n] = 0; ch.a2[ch.n] = 1; ch.n++;
        if (cur.hets != 0) { ch.a1[ch.n] = 1; ch.a2[ch.n] = 0; ch.n++; }
        ch.a1[ch.n] = 0; ch.a2[ch.n] = 0; ch.n++;
        ch.a1[ch.n] = 1; ch.a2[ch.n] = 1; ch.n++;
    }
    const uint32_t kp = p >> 5, bp = p & 31u;
    const uint32_t ck = cur.depth ? ((off + cur.depth - 1) >> 5) : (off >> 5);
    const bool trans = (ck != kp);
    if (trans) {
        // cur becomes the holder of a complete chunk that its descendants link to: persist it
        NodeRec r;
        r.frozen = cur.frozen; r.depth = cur.depth; r.hets = cur.hets; r.anc1 = cur.anc1; r.anc2 = cur.anc2;
        r.w0 = cur.w0; r.w1 = cur.w1;
        store_rec(pool + cur.idx, r);
    }
    uint32_t sumF[4], sumL[4];
    score_children(cx, cur, off, p, pool, ch, sumF, sumL);
    ch.depth = cur.depth + 1;
    ch.anc1 = trans ? (uint32_t)cur.idx : cur.anc1;
    ch.anc2 = trans ? cur.anc1 : cur.anc2;
    ch.w1 = trans ? cur.w0 : cur.w1;
    const Win basew = trans ? fresh_win() : cur.w0;
    for (int c = 0; c < ch.n; ++c) {
        ch.frozen[c] = cur.frozen + sumF[c];
        ch.total[c] = ch.frozen[c] + sumL[c] + h_next;
        ch.hets[c] = cur.hets + (ch.a1[c] != ch.a2[c] ? 1u : 0u);
        Win w = basew;
        if (ch.a1[c] < 2) {
            w.nv &= ~(1u << bp);
            w.h1 |= ch.a1[c] << bp;
            w.h2 |= ch.a2[c] << bp;
        }
        ch.w0[c] = w;
    }
}

DEVINL Cur child_as_cur(const Children& ch, int c, uint64_t idx) {
    Cur n;
    n.frozen = ch.frozen[c]; n.total = ch.total[c]; n.idx = idx;
    n.depth = ch.depth; n.hets = ch.hets[c]; n.anc1 = ch.anc1; n.anc2 = ch.anc2;
    n.w0 = ch.w0[c]; n.w1 = ch.w1;
    return n;
}
DEVINL void store_child(NodeRec* pool, const Children& ch, int c, uint64_t idx) {
    NodeRec r;
    r.frozen = ch.frozen[c]; r.depth = ch.depth; r.hets = ch.hets[c]; r.anc1 = ch.anc1; r.anc2 = ch.anc2;
    r.w0 = ch.w0[c]; r.w1 = ch.w1;
    store_rec(pool + idx, r);
}
DEVINL Cur cur_from_pool(const NodeRec* pool, Key k) {
    const uint64_t idx = k.lo >> 24;
    NodeRec r = load_rec(pool + idx);
    Cur n;
    n.frozen = r.frozen; n.total = k.hi >> 24; n.idx = idx;
    n.depth = r.depth; n.hets = r.hets; n.anc1 = r.anc1; n.anc2 = r.anc2;
    n.w0 = r.w0; n.w1 = r.w1;
    return n;
}
DEVINL Cur root_node(uint64_t heur) {
    Cur n;
    n.frozen = 0; n.total = heur; n.idx = 0; n.depth = 0; n.hets = 0; n.anc1 = NONE32; n.anc2 = NONE32;
    n.w0 = fresh_win(); n.w1 = fresh_win();
    return n;
}

struct WaveCounters {
    uint64_t sub_pops, main_pops, nodes;
};

// LDS ring of the last 64 heuristic values: H[x] lives at ring[x & 63] while x in [v, v+64)
DEVINL uint64_t ring_get(const uint64_t* ring, uint32_t x) {
    uint64_t v = 0;
    if (lane_id() == 0) v = ring[x & 63u];
    return bcast64(v);
}
DEVINL void ring_set(uint64_t* ring, uint32_t x, uint64_t v) {
    if (lane_id() == 0) ring[x & 63u] = v;
}

// astar_subsolver (astar_phaser.rs:311-405). Returns status; outputs (max_cost_so_far, farthest).
DEVINL int32_t subsolve(Ctx& cx, const SolveParams& prm, uint32_t off, uint32_t ps, Heap& heap, NodeRec* pool,
                        const uint64_t* ring, WaveCounters& wc, uint64_t& est, uint32_t& solved) {
    heap_reset(heap);
    uint64_t next_idx = 1;
    Cur cur = root_node(ring_get(ring, off + 1));  // initial_estimate = H[off+1] (astar_phaser.rs:322)
    uint32_t next_expected = 0, visited = 0;
    uint64_t max_cost = 0;
    const uint32_t max_visits = prm.minq_sub + prm.qinc * ps;
    int32_t st = ST_OK;
    while (cur.depth < ps && visited < max_visits) {
        visited += 1;
        wc.sub_pops += 1;
        if (cur.depth == next_expected) {
            max_cost = max(max_cost, cur.total);
            next_expected += 1;
        }
        const uint32_t p = off + cur.depth;
        Children ch;
        expand(cx, cur, off, p, ring_get(ring, p + 1), pool, ch);
        wc.nodes += ch.n;
        if (next_idx + ch.n > prm.cap_sub) { st = ST_OVERFLOW; break; }
        if (ch.a1[0] == 2 && ch.total[0] != cur.total) { st = ST_INVARIANT; break; }  // astar_phaser.rs:360
        int best = 0;
        Key kbest = make_key(ch.total[0], ch.hets[0], next_idx, ch.depth);
        for (int c = 1; c < ch.n; ++c) {
            Key k = make_key(ch.total[c], ch.hets[c], next_idx + c, ch.depth);
            if (key_less(k, kbest)) { kbest = k; best = c; }
        }
        const bool take_child = key_less(kbest, heap.top);  // top is inf when the heap is empty
        for (int c = 0; c < ch.n; ++c) {
            if (take_child && c == best) continue;
            store_child(pool, ch, c, next_idx + c);
            heap_push(heap, make_key(ch.total[c], ch.hets[c], next_idx + c, ch.depth));
        }
        if (take_child) {
            cur = child_as_cur(ch, best, next_idx + best);
        } else {
            Key t = heap.top;
            heap_pop(heap);
            cur = cur_from_pool(pool, t);
        }
        next_idx += ch.n;
        if (__any(heap.ovf)) { st = ST_OVERFLOW; break; }
    }
    if (cur.depth == ps) {  // astar_phaser.rs:395-399 (peek, not pop)
        max_cost = max(max_cost, cur.total);
        next_expected += 1;
    }
    est = max_cost;
    solved = next_expected - 1;
    return st;
}

template <bool SUB_LDS>
DEVINL int32_t solve_block(const BatchDev& B, uint32_t blk, uint32_t slot, Key* lds_heap, uint64_t* ring) {
    const SolveParams& prm = B.prm;
    const BlockDesc d = B.desc[blk];
    const uint32_t N = d.n_vars;
    const uint32_t lane = lane_id();
    Ctx cx;
    cx.vlo = B.vlo + d.var_off; cx.vhi = B.vhi + d.var_off; cx.vflags = B.vflags + d.var_off;
    cx.rstart = B.rstart + d.read_off; cx.rend = B.rend + d.read_off; cx.rword = B.rword + d.read_off;
    cx.words = B.words + d.word_off * WORD_DWORDS;
    cx.N = N; cx.evals = 0; cx.cells = 0;
    uint64_t* H = B.H + d.h_off;
    NodeRec* sub_pool = B.sub_pool + (size_t)slot * prm.cap_sub;
    NodeRec* main_pool = B.main_pool + (size_t)slot * prm.cap_main;
    uint32_t* tracker = B.tracker + (size_t)slot * ((size_t)prm.max_n_vars + 1);
    WaveCounters wc{0, 0, 0};
    int32_t st = ST_OK;

    Heap sub;
    sub.base = SUB_LDS ? lds_heap : (B.sub_heap_g + (size_t)slot * prm.jcap_sub * 64);
    sub.jcap = prm.jcap_sub;
    sub.ovf = 0;

    // ---- calculate_astar_heuristic (astar_phaser.rs:246-292) -------------------------------------------
    if (lane == 0) H[N] = 0;
    ring_set(ring, N, 0);
    uint32_t clip = 1;
    for (uint32_t v = N; v-- > 0;) {
        ring_set(ring, v, 0);  // heuristic_costs[problem_offset] is still 0 (astar_phaser.rs:320)
        uint64_t est = 0;
        uint32_t solved = 0;
        st = subsolve(cx, prm, v, clip, sub, sub_pool, ring, wc, est, solved);
        if (st != ST_OK) break;
        if (solved < min(clip, 2u)) { st = ST_INVARIANT; break; }  // astar_phaser.rs:268
        const bool bad = (cx.vflags[v] & HP_VAR_IGNORED) != 0;
        const uint64_t hnext = ring_get(ring, v + 1);
        uint64_t hv;
        if (bad) hv = hnext;
        else {
            if (est < hnext) { st = ST_INVARIANT; break; }  // astar_phaser.rs:284
            hv = est;
        }
        ring_set(ring, v, hv);
        if (lane == 0) H[v] = hv;
        clip = min(solved + 1, prm.max_seg);
    }

    // ---- main pruned search (astar_phaser.rs:451-633) ---------------------------------------------------
    hp_phase_stats stats{};
    if (st == ST_OK) {
        Heap hq;
        hq.base = B.main_heap + (size_t)slot * prm.jcap_main * 64;
        hq.jcap = prm.jcap_main;
        hq.ovf = 0;
        heap_reset(hq);
        // PQueueHapTracker (astar_phaser.rs:171-231): per-length counts live in global scratch and are only
        // ever touched by lane 0 (plain same-thread read-modify-write); the running total is a register.
        if (lane == 0) for (uint32_t i = 0; i <= N; ++i) tracker[i] = 0;
        uint32_t trk_total = 0, trk_thr = 0;
        auto trk_add = [&](uint32_t len, uint32_t n) {
            if (lane == 0) tracker[len] = tracker[len] + n;
            if (len >= trk_thr) trk_total += n;
        };
        auto trk_remove = [&](uint32_t len) {
            if (lane == 0) tracker[len] = tracker[len] - 1u;
            if (len >= trk_thr) trk_total -= 1;
        };

        uint64_t thr = prm.minq_main;                      // curr_queue_size_threshold
        const uint64_t max_q = 10ull * prm.minq_main;      // max_queue_size
        uint32_t min_progress = 0, next_expected = 0;
        uint64_t pruned = 0, next_idx = 1, qlen = 1;
        const uint64_t h0 = bcast64(lane == 0 ? H[0] : 0);
        Cur cur = root_node(h0);
        trk_add(0, 1);

        while (cur.depth < N) {
            wc.main_pops += 1;
            qlen -= 1;
            trk_remove(cur.depth);
            if (cur.depth == next_expected) {
                next_expected += 1;
                if (pruned == 0) thr += prm.qinc;
            }
            if (cur.depth < min_progress) {  // astar_phaser.rs:507-515
                if (pruned == 0) thr = prm.minq_main;
                pruned += 1;
                if (heap_empty(hq)) { st = ST_INVARIANT; break; }
                Key t = hq.top;
                heap_pop(hq);
                cur = cur_from_pool(main_pool, t);
                continue;
            }
            const uint32_t p = cur.depth;
            const uint64_t hn = bcast64(lane == 0 ? H[p + 1] : 0);
            Children ch;
            expand(cx, cur, 0, p, hn, main_pool, ch);
            wc.nodes += ch.n;
            if (next_idx + ch.n > prm.cap_main) { st = ST_OVERFLOW; break; }
            if (ch.a1[0] == 2 && ch.total[0] != cur.total) { st = ST_INVARIANT; break; }  // astar_phaser.rs:529
            int best = 0;
            Key kbest = make_key(ch.total[0], ch.hets[0], next_idx, ch.depth);
            for (int c = 1; c < ch.n; ++c) {
                Key k = make_key(ch.total[c], ch.hets[c], next_idx + c, ch.depth);
                if (key_less(k, kbest)) { kbest = k; best = c; }
            }
            // push every child except the best one, which is held in registers (it is logically queued)
            trk_add(ch.depth, (uint32_t)ch.n);
            for (int c = 0; c < ch.n; ++c) {
                if (c == best) continue;
                store_child(main_pool, ch, c, next_idx + c);
                heap_push(hq, make_key(ch.total[c], ch.hets[c], next_idx + c, ch.depth));
            }
            qlen += ch.n;
            // astar_phaser.rs:564-585
            while (trk_total > thr && min_progress < next_expected) {
                min_progress += 1;
                {   // increase_threshold(min_progress)
                    uint32_t c = 0;
                    if (lane == 0) c = tracker[min_progress - 1];
                    trk_total -= bcast32(c);
                    trk_thr = min_progress;
                }
                if (qlen > max_q) {
                    // full prune: every queued node shorter than min_progress gets the cleared priority
                    // (cost 0, same hets, same index); each lane rewrites and re-heapifies its private heap
                    Key* hp = hq.base + lane;
                    for (uint32_t j = 0; j < hq.cnt; ++j) {
                        Key k = hp[(size_t)j * 64];
                        if ((uint32_t)(k.lo & 0xFFFFFFu) < min_progress) {
                            k.hi &= 0xFFFFFFull;
                            hp[(size_t)j * 64] = k;
                        }
                    }
                    for (uint32_t i = hq.cnt / 2; i-- > 0;) lane_sift_down(hp, i, hq.cnt, hp[(size_t)i * 64]);
                    heap_recompute_top(hq);
                    if (ch.depth < min_progress) kbest.hi &= 0xFFFFFFull;
                }
            }
            if (key_less(kbest, hq.top)) {
                cur = child_as_cur(ch, best, next_idx + best);
                cur.total = kbest.hi >> 24;
            } else {
                store_child(main_pool, ch, best, next_idx + best);
                heap_push(hq, kbest);
                Key t = hq.top;
                heap_pop(hq);
                cur = cur_from_pool(main_pool, t);
            }
            next_idx += ch.n;
            if (__any(hq.ovf)) { st = ST_OVERFLOW; break; }
        }

        if (st == ST_OK) {
            // ---- emit the solution (astar_phaser.rs:588-628): walk the window chain from the last chunk down
            uint8_t* o1 = B.h1 + d.var_off;
            uint8_t* o2 = B.h2 + d.var_off;
            uint64_t phased = 0, snvs = 0, skipped = 0;
            const uint32_t last_ck = (N - 1) >> 5;
            uint32_t chunk = last_ck;
            Win w = cur.w0, wn = cur.w1;
            uint32_t slot_next = cur.anc2;
            bool have_wn = true;
            for (;;) {
                const uint32_t pos = chunk * 32 + (lane & 31u);
                const bool inb = lane < 32 && pos < N;
                const uint32_t bit = lane & 31u;
                const uint32_t nvb = (w.nv >> bit) & 1u, b1 = (w.h1 >> bit) & 1u, b2 = (w.h2 >> bit) & 1u;
                if (inb) {
                    o1[pos] = nvb ? 2 : (uint8_t)b1;
                    o2[pos] = nvb ? 2 : (uint8_t)b2;
                }
                const uint64_t m_in = __ballot(inb);
                const uint64_t m_het = __ballot(inb && !nvb && b1 != b2);
                const uint64_t m_skip = __ballot(inb && nvb);
                const uint64_t m_snv = __ballot(inb && (cx.vflags[inb ? pos : 0] & HP_VAR_SNV));
                (void)m_in;
                phased += __popcll(m_het);
                skipped += __popcll(m_skip);
                snvs += __popcll(m_het & m_snv);
                if (chunk == 0) break;
                chunk -= 1;
                if (have_wn) {
                    w = wn;
                    have_wn = false;
                } else {
                    if (slot_next == NONE32) { st = ST_INVARIANT; break; }
                    NodeRec a = load_rec(main_pool + slot_next);
                    w = a.w0;
                    wn = a.w1;
                    slot_next = a.anc2;
                    have_wn = true;
                }
            }
            stats.pruned_solutions = pruned;
            stats.estimated_cost = h0;
            stats.actual_cost = cur.total;
            stats.phased_variants = phased;
            stats.phased_snvs = snvs;
            stats.skipped_variants = skipped;
            stats.homozygous_variants = (uint64_t)N - phased - skipped;
            if (cur.total < h0) st = ST_INVARIANT;  // phase_stats.rs:163
        }
    }

    const uint64_t evals = wave_sum_u64(cx.evals), cells = wave_sum_u64(cx.cells);
    if (lane == 0) {
        B.stats[blk] = stats;
        hp_work_counters c{};
        c.sub_pops = wc.sub_pops; c.main_pops = wc.main_pops; c.evals = evals; c.cells = cells; c.nodes_created = wc.nodes;
        B.counters[blk] = c;
        B.status[blk] = st;
    }
    return st;
}

template <bool SUB_LDS>
__global__ void __launch_bounds__(64) hp_astar_kernel(BatchDev B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* ring = reinterpret_cast<uint64_t*>(smem);            // 64 x u64
    Key* lds_heap = reinterpret_cast<Key*>(smem + 64 * sizeof(uint64_t));
    const uint32_t slot = blockIdx.x;
    const uint32_t G = gridDim.x;
    // Static "snake" assignment over the LPT-sorted work list: workgroup w takes ranks w, 2G-1-w, 2G+w, ...
    // (uniform control flow, no atomics; every workgroup gets a similar mix of large and small blocks).
    for (uint32_t round = 0;; ++round) {
        const uint32_t base = round * G;
        if (base >= B.n_items) break;
        const uint32_t i = base + ((round & 1u) ? (G - 1u - slot) : slot);
        if (i < B.n_items) solve_block<SUB_LDS>(B, B.order[i], slot, lds_heap, ring);
    }
}

// explicit instantiations used by the host code
template __global__ void hp_astar_kernel<true>(BatchDev);
template __global__ void hp_astar_kernel<false>(BatchDev);

}  // namespace hp
