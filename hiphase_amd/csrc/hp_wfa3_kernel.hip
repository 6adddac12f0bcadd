// hp_wfa3_kernel.hip — graph-WFA (unit costs, end-to-end) on gfx950, third generation: A ROUND'S WAVES AS ONE FLAT SORTED LIST.
//
// Replaces, bit-identically, reference src/wfa_graph.rs:350-650 `edit_distance_with_pruning`, as hp_wfa2_kernel.hip does; same
// batch layout, same job lists, same hand-over protocol between the graph-size classes, same capped-diagonal records (W2Batch,
// hp_wfa2_dev.h) - a drop-in for the class launches of hp_wfa2.hip. What changed is the shape of a lockstep step.
//
// Second generation: a step = one NODE's diagonals for each group of G lanes. A HiFi read's round holds a dozen live (node,
// diagonal) slots spread over three or four nodes: three or four steps a round with a third of the lanes holding a wave, and a
// state machine per group (entries, clusters, hulls, items, pending queue) that is uniform within a group but differs between the
// groups of a wavefront - vector code, 290 instructions per group and node visit (profiles/round3/wfa2_phases.txt).
// Here (W3Cfg, hp_wfa2_dev.h; CPU model: tests/cpp/wfa2_model.cpp `model_wfa3`, pinned to the oracle):
//   * a round's live waves are ONE list sorted by key = node << 19 | diagonal + 2^18, 8 bytes a slot in LDS (key, offset << 3 | kind),
//     their traversed-node sets in HBM at the same index;
//   * the next round's TARGETS are built from it in one pass: slot (n, d) emits (n, d - 1), (n, d), (n, d + 1) unless its
//     predecessor in the list did; a target's candidates (wfa_graph.rs:555-573 turned into a pull: d + 1 -> offset + 1,
//     d -> offset + 1, d - 1 -> offset) are among the three list entries from its emitter on;
//   * a step is a TILE of the next G targets whatever nodes they belong to: the node descriptor, the sequence pointer and the
//     capped-diagonal record are per LANE (a vector load costs the same whether its lanes read one address or eight);
//   * a wave that finishes its node (wfa_graph.rs:527-553: the children take it up THIS round at offset 0) becomes a target
//     (child, diagonal + node length) inserted into the sorted target list, or joins the target that is already there. Nodes must
//     still be taken in index order within a round: a tile that holds a finishing wave of node q COMMITS only its slots of nodes
//     below q's first child (ids are topological - nothing before that child can be reached from q this round); the rest stay
//     targets and are computed again, with what the finished waves hand them, by the next tile. Progress: q itself commits.
// Nothing else is new: candidates, tie rule, capped diagonals instead of max_wavefronts, pruning, finals are the second
// generation's, per (node, diagonal) - the results are the same set of slots with the same contents, found in fewer steps.
// Integer / byte work, no MFMA; bound by instruction issue, hence fewer, fuller steps.
#include "hp_common.h"
#include "hp_wfa2_dev.h"

namespace hp {

#ifndef W3_STATS
#define W3_STATS 0
#endif
#if W3_STATS
#define W3C(i, v) do { w3c[i] += (uint64_t)(v); } while (0)
#else
#define W3C(i, v)
#endif
#ifndef W3_PROF
#define W3_PROF 0
#endif
// Round 6: a slot carries the RING INDEX of its traversed-node set instead of owning a copy of it. Until round 5 every committed slot
// wrote its set (W words) to HBM every round and every tile loaded up to five of them - 24 KB written per read, all of it copies:
// a wave that only advances keeps its set. Now the group's set arena is a list of IMMUTABLE entries; a slot's word carries the entry's index (10 bits beside offset and kind), a tile forwards the index of its
// tied candidate, and an entry is written only where a set really changes: a wave taken up by a child node (parents' sets + the
// node, wfa_graph.rs:535-541) and a tie between waves whose sets differ (:476-510) - a few hundred times per read instead of
// eight thousand. The reference hash-conses its sets for the same reason (wfa_graph.rs:363-370). Entries may duplicate each other
// (no look-up for an equal set): a tie between two copies writes a third - harmless. A job's arena holds W3_ARENA (1 023) entries,
// written once each in order; a job that needs more is handed on like one that outgrows its slot lists (why = 8).
// -DW3_SETIDS=0: the round-5 kernel.
// (W3_SETIDS / W3_ARENA: hp_wfa2_dev.h - the host sizes the groups' scratch from them)
#if W3_PROF   // s_memtime between the phases of the lockstep step (a separate build: the timers cost registers)
#define W3T(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); w3t[i] += t_ - w3tl; w3tl = t_; } while (0)
#else
#define W3T(i)
#endif

// A group's ballot as a 32-bit word (G <= 16): the group's bits of the wavefront's ballot - one select and one bit-field extract, and
// everything computed from it (counts, first set bit, tests) stays 32-bit vector code (the 64-bit shift / and / count it replaces
// were two passes each).
template <int G> W2DEV uint32_t w3_gb(bool pred, uint32_t gbase) {
    const uint64_t b = __ballot(pred);
    const uint32_t half = (gbase & 32u) ? (uint32_t)(b >> 32) : (uint32_t)b;
    return __builtin_amdgcn_ubfe(half, gbase & 31u, (uint32_t)G);
}
// bits of `m` (a group's ballot) below this lane; lmask = (1 << gl) - 1
W2DEV uint32_t w3_below(uint32_t m, uint32_t lmask) { return (uint32_t)__popc(m & lmask); }

// Lanes with done == false have matched their first n bytes of their node (at global offset nb + o) against read[pos..] and may match
// up to maxlen: the group serves them one after the other, G x 32 bytes per pass. The serving lane's node address travels (two
// dwords), its read offset and what is left. `on`: this group takes part (group-uniform).
template <int G> W2DEV uint32_t w3_match_rest(const uint8_t* seq, uint64_t nb, const uint8_t* readp, uint32_t o, int32_t pos, uint32_t maxlen, uint32_t n, bool done,
                                              bool on, uint32_t gbase, uint32_t gl) {
    constexpr uint32_t LB = W2_MATCH_LANE_BYTES;
    bool pending = on && !done;
    while (__any(pending)) {
        const uint32_t gb = w3_gb<G>(pending, gbase);
        const bool active = gb != 0;
        const uint32_t L = active ? (uint32_t)__builtin_ctz(gb) : 0u;
        const uint64_t a = nb + o + n;
        const uint32_t alo = w2_gsel<G>((uint32_t)a, gl, L), ahi = w2_gsel<G>((uint32_t)(a >> 32), gl, L);
        const uint32_t sp = w2_gsel<G>((uint32_t)pos + n, gl, L), rem = w2_gsel<G>(maxlen - n, gl, L);
        const uint8_t* pa = seq + (((uint64_t)ahi << 32) | alo);
        const uint8_t* pb = readp + sp;
        const uint32_t off = gl * LB;
        uint32_t m = LB;
        if (active) {
            if (off < rem) {
                uint4 x[LB / 16], y[LB / 16];
#pragma unroll
                for (uint32_t k = 0; k < LB / 16; ++k) { x[k] = w2_ld16(pa + off + 16 * k); y[k] = w2_ld16(pb + off + 16 * k); }
                m = 0;
#pragma unroll
                for (uint32_t k = 0; k < LB / 16; ++k) if (m == 16u * k) m += w2_pfx16(x[k], y[k]);
                if (m > rem - off) m = rem - off;
            } else m = 0u;   // beyond the end: acts as a stop
        }
        const uint32_t stop = w3_gb<G>(active && m < LB, gbase);
        uint32_t got = (uint32_t)G * LB;
        if (stop) {
            const uint32_t S = (uint32_t)__builtin_ctz(stop);
            got = S * LB + w2_gsel<G>(m, gl, S);
        }
        if (got > rem) got = rem;
        if (active && gl == L) {
            n += got;
            if (stop || n >= maxlen) pending = false;
        }
    }
    return n;
}

template <int G, int W, bool WIDE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W3Cfg<W>::WAVES_PER_SIMD, W3Cfg<W>::WAVES_PER_SIMD))) hp_wfa3_kernel(W2Batch B) {
    using C = W3Cfg<W, WIDE>;
    static_assert(G >= 8 && G <= 16 && (G & (G - 1)) == 0, "group size");
    static_assert(W <= G, "one lane per set word in the cooperative set merges");
    constexpr uint32_t NG = 64 / G;
    constexpr uint32_t SL = (uint32_t)C::SLOTS;
    const uint32_t lane = w2_lane(), gid = lane / G, gl = lane % G, gbase = gid * G;
    const uint32_t lmask = (1u << gl) - 1u;
    unsigned char* R = w2_smem + (size_t)gid * C::BYTES;
    uint2* A = reinterpret_cast<uint2*>(R + C::O_A);                 // [parity * SLOTS + i]: targets (from ip on) and live slots (from 0 on)
    uint32_t* outset = reinterpret_cast<uint32_t*>(R + C::O_MISC);

    const uint32_t slot = blockIdx.x * NG + gid;
    uint64_t* htab = B.htab + ((size_t)slot << B.hcap_log2);
    uint32_t* gs = B.gsets + (size_t)slot * B.set_stride;            // [(parity * SLOTS + s) * W], then the capped records
    const uint32_t hmask = (1u << B.hcap_log2) - 1u;
    const uint32_t prune32 = B.prune_distance > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)B.prune_distance;
    const uint32_t maxed32 = B.max_ed > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (uint32_t)B.max_ed;

    // ---- group-uniform state ---------------------------------------------------------------------------------------------------
    enum : uint32_t { S_JOB = 0, S_ROUND = 1, S_WAIT = 2, S_TILE = 3, S_DONE = 4 };
    uint32_t state = S_JOB, jround = 0, job = 0, ticket = 0, idle_polls = 0;
    bool have_ticket = false;
    uint32_t poll_div = 0;
    uint32_t n_nodes = 0, last = 0, other_len = 0, tag = 0;
    uint64_t ref_off = 0;                                            // of the job's reference window in seq[]
    const uint8_t* readp = B.seq;
    const W2Node* gnode = B.nodes; const uint16_t* gedge = B.edges;
    uint32_t ed = 0, c = 0, p = 1, ip = 0, np = 0, nl = 0, nf = 0, nl_prev = 0;
    uint32_t lastkey = 0;                                            // key of the last target of the round's list (np > ip)
#if W3_SETIDS
    constexpr uint32_t CAP = (uint32_t)C::SET_ENTRIES;               // entries of a job's set arena: written once each, in order
    uint32_t ring_next = 0;                                          // where the next entry goes
#endif
    uint32_t farthest = 0, min_prog = 0;
    bool final_found = false;
    int32_t status = W2_ST_PENDING;
    uint32_t score = 0, steps = 0, why = 0;
    uint32_t lane_far = 0, lane_upd = 0;                             // per lane
#if W3_STATS
    uint64_t w3c[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // wave steps, control passes, group tiles, act lanes, has lanes, committed lanes, jobs, rounds, long extensions (lanes), inserts, tiles with a discard, build chunks
#endif
    const uint32_t n_class = *B.n_items_dev;
    if (B.esc_role == 1u) (void)atomicAdd(B.esc + 4, lane == 0 ? 1u : 0u);   // a producer workgroup has started
#ifndef W3_PREFETCH
#define W3_PREFETCH 0   // measured (round 4): no gain - 24.6-27.2 ms against 24.3-25.2 ms per launch stage over three runs each, and 18 more registers
#endif
#if W3_PREFETCH
    uint32_t pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0;   // see "next round's bases" below
#endif
#if W3_PROF
    uint64_t w3t[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t w3tl = __builtin_amdgcn_s_memtime();
    const uint64_t w3t0 = w3tl;
    uint32_t w3steps = 0;
#endif

    for (;;) {
        // ============================ 1. control: advance every group to its next tile ===========================================
        uint32_t spins = 0;
        W3T(0);
        if (B.esc_role == 2u && __any(state == S_WAIT) && (!__any(state == S_TILE) || (++poll_div & 15u) == 0u)) {
            const uint32_t gone = atomicAdd(B.esc + 2, 0u);
            W2_WAIT_VM();
            const uint32_t pub = atomicAdd(B.esc + 1, 0u);
            bool alone = false;
            if (idle_polls > 750u && gone == 0u) alone = atomicAdd(B.esc + 4, 0u) == 0u;
            if (state == S_WAIT) {
                if (pub > ticket) { state = S_JOB; have_ticket = true; }
                else if (gone >= B.esc_producers || alone) state = S_DONE;
            }
        }
        while (state != S_TILE && state != S_DONE && state != S_WAIT) {
            W3C(1, 1);
            if (++spins > (1u << 20)) { state = S_DONE; break; }   // cannot happen; never hang the device
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (state == S_JOB) {
                if (status != W2_ST_PENDING) {   // results of the job that just ended
                    const uint32_t upd = w2_gsum<G>(lane_upd);
                    bool handed_over = false;
                    // worth handing over: what the largest class's bigger tables fix (the slot lists) - not a full capped set
                    // ... and not a read that is heading for the neighbourhood of max_edit_distance anyway (the workload's noisy tail: at
                    // 5 % noise a read holds fifty diagonals per node and outgrows every class's lists): by the rate it has made so far -
                    // edits per read base - it would end beyond hopeless_pct (50) per cent of the cap. Those skip the largest class - where
                    // they would crawl on to outgrow its lists too, as the tail of the launch set: at 125 per cent a hundred reads a set
                    // still did, and the class ended 12-15 ms after the other two - and go straight to the host's pass (the exact
                    // reference-window verdict of hp_wfa2_bound_kernel, then the dense band). Routing only: every road computes the same
                    // result. (A dozen rounds in: the bench's noisy reads outgrow the lists around their twelfth round; an ordinary HiFi
                    // read over a structural variant projects to 20-30 per cent of the cap and is handed over as before.)
                    const bool hopeless = ed >= 12u && (uint64_t)ed * other_len * 100u > ((uint64_t)farthest + 1u) * (uint64_t)maxed32 * B.hopeless_pct;
                    const bool hand = status == W2_ST_NEED_BIG && why == 8u && !hopeless;
                    if (B.esc_role == 1u && __any(hand)) {
                        const uint32_t taken = atomicAdd(B.esc, 0u);
                        const bool me = hand && gl == 0 && taken < B.esc_limit;
                        const uint32_t pos = atomicAdd(B.esc, me ? 1u : 0u);
                        (void)atomicExch(me ? B.esc_order + pos : B.esc + 3, me ? job : 0u);
                        if (me) B.handed[job] = 1;   // (never written by anyone else: hp_wfa2_map_kernel)
                        W2_WAIT_VM();
                        bool pend = me;
                        for (uint32_t s = 0; s < (1u << 16) && __any(pend); ++s) {
                            const uint32_t seen = atomicCAS(pend ? B.esc + 1 : B.esc + 3, pend ? pos : 0xFFFFFFFFu, pend ? pos + 1u : 0xFFFFFFFFu);
                            if (pend && seen == pos) pend = false;
                        }
                        handed_over = w3_gb<G>(me, gbase) != 0;
                    }
                    if (!handed_over && gl == 0) { B.status[job] = status; B.out_score[job] = status == W2_ST_NEED_BIG ? (uint64_t)(why | (ed << 8)) : score; B.out_work[(size_t)job * 2] = upd; }
                    if (!handed_over && gl < (uint32_t)W) B.out_sets[(size_t)job * W2_SET_STRIDE + gl] = outset[gl];
                    status = W2_ST_PENDING;
                    W3C(6, __popcll(__ballot(gl == 0)));
                }
                if (B.group_jobs != 0u && B.esc_role != 2u && jround >= B.group_jobs) { state = S_DONE; break; }
                const uint32_t mine = atomicAdd(B.next, (gl == 0 && !have_ticket) ? 1u : 0u);   // (every lane takes part: no one-lane branch)
                const uint32_t k = have_ticket ? ticket : w2_gsel<G>(mine, gl, 0u);
                jround++;
                if (B.esc_role == 2u) {
                    if (!have_ticket && k >= n_class) {
                        const uint32_t pub = atomicAdd(B.esc + 1, 0u);
                        if (k >= pub) { ticket = k; state = S_WAIT; break; }
                    }
                    have_ticket = false;
                    job = k < n_class ? B.order[k] : atomicAdd(B.esc_order + k, 0u);
                } else {
                    if (k >= n_class) { state = S_DONE; break; }
                    job = B.order[k];
                }
                const W2Job jd = B.jobs[job];
                const W2Info ji = B.info[job];
                n_nodes = ji.n_nodes; last = n_nodes - 1u; other_len = jd.read_len;
                ref_off = jd.ref_off; readp = B.seq + jd.read_off;
                tag = B.tag_base + job + 1u;
                if (gl < (uint32_t)W) outset[gl] = 0u;
                score = 0;
                if (n_nodes == 0 || n_nodes > (uint32_t)C::MAXN || ji.n_edges > (uint32_t)C::MAXE || other_len >= (uint32_t)W2_DIAG_LIM) {
                    status = W2_ST_NEED_BIG, why = 4u;
                    continue;   // stays in S_JOB: the next pass writes this status and fetches the next job
                }
                gnode = B.nodes + jd.node_off;
                gedge = B.edges + jd.edge_off;
                // the start wave (wfa_graph.rs:366-378): the only target of round 0
                ed = 0; c = 0; p = 1; ip = 0; np = 1; nl = 0; nf = 0; nl_prev = 0; steps = 0;
#if W3_SETIDS
                ring_next = 0;
#endif
                if (gl == 0) A[0] = make_uint2(w3_key(0u, 0), w3_aux(W3_NONE, W3_NONE, W3_NONE) | W3_START);
                lastkey = w3_key(0u, 0);
                farthest = 0; min_prog = 0; final_found = false; lane_far = 0; lane_upd = 0;
                state = S_TILE;
                continue;
            }
            // ---- state == S_ROUND: end of round (wfa_graph.rs:633-648), then the next round's targets ----
            {
                const uint32_t far = (uint32_t)w2_gmax<G>((int32_t)lane_far);
                lane_far = 0;
                if (final_found) { status = W2_ST_OK; score = ed; state = S_JOB; continue; }
                if (far > farthest) farthest = far;
                if (farthest > prune32) min_prog = farthest - prune32;
                if (ed + 1u > maxed32) { status = W2_ST_MAX_ED; score = maxed32; state = S_JOB; continue; }
                if (nl == 0u) { status = W2_ST_INTERNAL; state = S_JOB; continue; }
                ++ed; p = c; c ^= 1u; nl_prev = nl; nl = 0; nf = 0; ip = 0; np = 0;
                W3C(7, __popcll(__ballot(gl == 0)));
                // every live slot (n, d) emits the targets (n, d - 1), (n, d), (n, d + 1) its predecessor has not emitted
                uint32_t carry = 0xFFFFFFFFu;
                bool over = false;

#pragma clang loop unroll(disable)
                for (uint32_t base = 0; base < nl_prev; base += (uint32_t)G) {
                    W3C(11, 1);
                    const uint32_t i = base + gl;
                    const bool valid = i < nl_prev;
                    const uint32_t key = valid ? A[p * SL + i].x : 0xFFFFFFFFu;
                    uint32_t prevkey = (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)key, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
                    if (gl == 0) prevkey = carry;
                    const uint32_t nn = w3_key_node(key);
                    const int32_t d = w3_key_diag(key);
                    const bool same = prevkey != 0xFFFFFFFFu && w3_key_node(prevkey) == nn;
                    const int32_t gap = same ? d - w3_key_diag(prevkey) : 3;
                    const bool c1 = valid && gap >= 2, c2 = valid && gap >= 3;
                    const uint32_t mv = w3_gb<G>(valid, gbase), m1 = w3_gb<G>(c1, gbase), m2 = w3_gb<G>(c2, gbase);
                    const uint32_t excl = w3_below(mv, lmask) + w3_below(m1, lmask) + w3_below(m2, lmask);
                    const uint32_t total = (uint32_t)(__popc(mv) + __popc(m1) + __popc(m2));
                    if (np + total > SL) { over = true; break; }
                    if (valid) {
                        const uint32_t cnt = 1u + (c1 ? 1u : 0u) + (c2 ? 1u : 0u);
                        const uint32_t at = c * SL + np + excl;
                        const uint32_t aux = w3_aux(i, W3_NONE, W3_NONE);
                        // cnt 3: d - 1, d, d + 1; cnt 2: d, d + 1; cnt 1: d + 1
                        if (c2) A[at] = make_uint2(w3_key(nn, d - 1), aux);
                        if (c1) A[at + cnt - 2u] = make_uint2(w3_key(nn, d), aux);
                        A[at + cnt - 1u] = make_uint2(w3_key(nn, d + 1), aux);
                        if (d - 1 <= -W2_DIAG_LIM || d + 1 >= W2_DIAG_LIM) over = true;
                    }
                    np += total;
                    lastkey = (uint32_t)w2_gmax<G>(valid ? (int32_t)w3_key(nn, d + 1) : 0);   // (keys ascend: the chunk's last target)
                    carry = w2_gsel<G>(key, gl, (uint32_t)G - 1u);   // (0xFFFFFFFF when the chunk is not full: it was the last one)
                }
                if (w3_gb<G>(over, gbase)) { status = W2_ST_NEED_BIG, why = 8u; state = S_JOB; continue; }
                state = S_TILE;
            }
        }
        W3T(1);
        if (!__any(state == S_TILE)) {
            if (!__any(state == S_WAIT)) break;   // every group is done
            if (++idle_polls > 60000u) {          // (~1.5 s of nothing to do: never hang the device; unclaimed jobs stay for the host's pass)
#ifdef W3_DEBUG_TIMEOUT
                if (lane == 0) printf("w3 timeout: wg %u role %u gone %u producers %u started %u pub %u taken %u ticket %u n_class %u\n", blockIdx.x, B.esc_role, atomicAdd(B.esc + 2, 0u), B.esc_producers, atomicAdd(B.esc + 4, 0u), atomicAdd(B.esc + 1, 0u), atomicAdd(B.esc, 0u), ticket, n_class);
#endif
                break;
            }
            for (int z = 0; z < 8; ++z) __builtin_amdgcn_s_sleep(127);
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        // ============================ 2. one tile: the next G targets of the round ===============================================
        const bool run = state == S_TILE;
        const uint32_t pbase = p * SL, cbase = c * SL;
        const bool act = run && ip + gl < np;
        W3C(0, 1); W3C(2, __popcll(__ballot(run && gl == 0))); W3C(3, __popcll(__ballot(act)));
        const uint2 tgt = act ? A[cbase + ip + gl] : make_uint2(0u, w3_aux(W3_NONE, W3_NONE, W3_NONE));
        const uint32_t n = w3_key_node(tgt.x);
        const int32_t d = act ? w3_key_diag(tgt.x) : 0;
        const uint32_t back = tgt.y & 0x3FFu, src0 = (tgt.y >> 10) & 0x3FFu, src1 = (tgt.y >> 20) & 0x3FFu;
        // The node of this lane's target (16 bytes from HBM, L2-resident), its capped-diagonal record (neighbouring nodes share a line),
        // the three candidate slots and the candidates' sets: every load is issued UNCONDITIONALLY on an address that is valid whatever
        // the lane holds (node 0, slot 0 for a lane without a target) and the results are masked - round 5: as `if (act) load` chains the
        // compiler built nested exec-mask regions with a wait after each of the three list reads (three serialised trips to the LDS
        // and a dozen branches per tile).
        // (a lane without a target reads ONE line of the whole device - the first node, the first words of gsets: always in the
        // compute unit's own cache - not its group's slot 0: with per-group dummies every masked load was a request to L2, 16 % more
        // of them than the predicated loads issued, and 15 % more HBM traffic through the pressure on it)
        const uint4 nd = *reinterpret_cast<const uint4*>(act ? gnode + n : B.nodes);
        const uint4 crec = *reinterpret_cast<const uint4*>(act ? gs + C::SET_DWORDS + 4u * n : B.gsets);
        // ---- candidates from the previous round: the three live slots from the emitter on ----
        int32_t oA = -1, oB = -1, oC = -1;
        int32_t sA = -1, sB = -1, sC = -1;   // their slots (= set indices)
#if W3_SETIDS
        uint32_t idA = 0, idB = 0, idC = 0;  // ring indices of their sets
#endif
        {
            const bool cand = act && back != W3_NONE;
            uint2 e[3];
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) e[k] = A[pbase + ((cand && back + k < nl_prev) ? back + k : 0u)];
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                const uint32_t j = back + k;
                const bool ok = cand && j < nl_prev;
                const int32_t rel = (int32_t)e[k].x - (int32_t)tgt.x;   // same node: the difference of the diagonals (keys are node << 19 | diagonal + bias)
                const uint32_t kd = e[k].y & 7u;
#if W3_SETIDS
                const int32_t off = (int32_t)((e[k].y >> 3) & 0x3FFFFu);
                const uint32_t eid = e[k].y >> 21;
#else
                const int32_t off = (int32_t)(e[k].y >> 3);
#endif
                const bool a = ok && rel == 1 && (kd & 1u) != 0u;
                const bool b = ok && rel == 0 && kd == W2_KIND_INTERIOR_READ;
                const bool c_ = ok && rel == -1 && (kd == W2_KIND_INTERIOR_READ || kd == W2_KIND_END_LAST);
                oA = a ? off + 1 : oA; sA = a ? (int32_t)j : sA;
                oB = b ? off + 1 : oB; sB = b ? (int32_t)j : sB;
                oC = c_ ? off : oC; sC = c_ ? (int32_t)j : sC;
#if W3_SETIDS
                idA = a ? eid : idA; idB = b ? eid : idB; idC = c_ ? eid : idC;
#endif
            }
        }
#if W3_SETIDS
        // ---- waves that finished a parent THIS round (offset 0; wfa_graph.rs:527-553), and the start wave: src0 / src1 are the ring
        // indices of the parents' sets ----
        const bool hinj = act && ((tgt.y & W3_START) != 0u || src0 != W3_NONE);
        const bool h0 = act && src0 != W3_NONE, h1 = act && src1 != W3_NONE;
        // (the parents' sets are asked for NOW, with the tile's other loads, on addresses that are valid whatever the lane holds - entry 0
        // of the device's first group for a lane without a parent: the one place a new entry is made in nearly every tile must not
        // cost a trip to memory of its own in the middle of it. Measured: with these loads where the entry is put together the class
        // kernels were 4 % SLOWER than round 5's - 18.0 against 17.3 ms a set - for all the stores they no longer make.)
        W2Set<W> qD = w2_ldset<W>(h0 ? gs + (size_t)src0 * W : B.gsets, true);
        {
            const W2Set<W> t = w2_ldset<W>(h1 ? gs + (size_t)src1 * W : B.gsets, true);
#pragma unroll
            for (int w = 0; w < W; ++w) qD.w[w] = (h0 ? qD.w[w] : 0u) | (h1 ? t.w[w] : 0u);
        }
#else
        const W2Set<W> qA = w2_ldset<W>(sA >= 0 ? gs + (size_t)(pbase + (uint32_t)sA) * W : B.gsets, true);
        const W2Set<W> qB = w2_ldset<W>(sB >= 0 ? gs + (size_t)(pbase + (uint32_t)sB) * W : B.gsets, true);
        const W2Set<W> qC = w2_ldset<W>(sC >= 0 ? gs + (size_t)(pbase + (uint32_t)sC) * W : B.gsets, true);
        // ---- waves that finished a parent THIS round (offset 0; wfa_graph.rs:527-553), and the start wave ----
        const bool hinj = act && ((tgt.y & W3_START) != 0u || src0 != W3_NONE);
        const bool h0 = act && src0 != W3_NONE, h1 = act && src1 != W3_NONE;
        W2Set<W> qD = w2_ldset<W>(h0 ? gs + (size_t)(cbase + src0) * W : B.gsets, true);
        {
            const W2Set<W> t = w2_ldset<W>(h1 ? gs + (size_t)(cbase + src1) * W : B.gsets, true);
#pragma unroll
            for (int w = 0; w < W; ++w) qD.w[w] = (h0 ? qD.w[w] : 0u) | (h1 ? t.w[w] : 0u);
        }
#endif
        W3T(2);
        const uint32_t len = nd.y & ~W2_IS_REF;
        const uint64_t nb = ((nd.y & W2_IS_REF) ? ref_off : B.alt_off) + nd.x;   // the node's sequence in seq[]
        const uint8_t* nseq = B.seq + nb;
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
        // (when the furthest is this diagonal's own wave - B: offset + 1 of a wave that stopped on a mismatch at offset omax - 1 of
        // this very diagonal - every other candidate would have to match through that position: none ties, nothing to check)
        const bool chk = has && !tB;
        const bool nA = chk && oA >= 0 && oA < omax, nB = chk && oB >= 0 && oB < omax, nC = chk && oC >= 0 && oC < omax;
        const bool nD = chk && hinj && omax > 0;
        // (a lane without a wave reads the first bytes of seq[]: one line for the whole device, always in cache - the start of ITS node or
        // read would be a cold line, and so would offset 0 for a tie check that is not needed: those read where the extension reads.
        // Measured: with cold dummy addresses the launch set fetched 70 % more from HBM, 137 against 90 KB per read)
        const uint8_t* ra = has ? readp + pos0 : B.seq;
        const uint8_t* na_ = has ? nseq + omax : B.seq;
#if W3_PREFETCH && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" :: "v"(pf0), "v"(pf1), "v"(pf2), "v"(pf3));   // (the previous step's look-ahead loads have landed long since: nothing waits here)
#endif
        // (unconditional loads on addresses that are valid for every lane - the start of the node / of the read for a lane that needs
        // none; seq[] ends in 256 bytes of padding - and results that are only looked at under the lane's own flag: no exec-mask region each)
        const W2Pre pm = w2_pre(na_, ra, true);
        const W2Pre8 pA = w2_pre8(nA ? nseq + oA : na_, nA ? readp + (d + oA) : ra, true);
        const W2Pre8 pB = w2_pre8(nB ? nseq + oB : na_, nB ? readp + (d + oB) : ra, true);
        const W2Pre8 pC = w2_pre8(nC ? nseq + oC : na_, nC ? readp + (d + oC) : ra, true);
        const W2Pre8 pD = w2_pre8(nD ? nseq : na_, nD ? readp + d : ra, true);
        W3T(3);
        uint32_t E;
        {
            uint32_t n0 = 0;
            bool done = (room == 0);
            if (!done) {
                uint32_t m = w2_pfx16(pm.a, pm.b);
                if (m > room) m = room;
                n0 = m;
                if (m < 16 || n0 >= room) done = true;
            }
            W3C(8, __popcll(__ballot(run && !done)));
            E = (uint32_t)omax + w3_match_rest<G>(B.seq, nb, readp, (uint32_t)omax, pos0, room, n0, done, run, gbase, gl);
        }
        W3T(4);
        {
            bool pdA = false, pdB = false, pdC = false, pdD = false;
            auto quick = [&](const W2Pre8& q, bool nX, int32_t oX, bool& pend8) -> bool {
                if (!nX) return false;
                const uint32_t g = (uint32_t)(omax - oX), m = w2_pfx8(q.a, q.b);
                if (g <= 8u) return m >= g;
                pend8 = (m == 8u);
                return false;
            };
            const bool xA = quick(pA, nA, oA, pdA), xB = quick(pB, nB, oB, pdB), xC = quick(pC, nC, oC, pdC), xD = quick(pD, nD, 0, pdD);
            tA = tA || xA; tB = tB || xB; tC = tC || xC; tD = tD || xD;
            if (__any(pdA || pdB || pdC || pdD)) {
                // the long compare only decides whether the candidate's traversed nodes join the slot's set: one whose set adds
                // nothing to what the tied candidates bring already (the usual case: the same path, a diagonal over) needs none
#if W3_SETIDS
                // (by index: a candidate that names the entry a tied one names brings nothing new; different indices may still hold equal
                // sets - then the long compare runs for nothing, the result is the same)
                const bool addA = !((tB && idB == idA) || (tC && idC == idA));
                const bool addB = !((tA && idA == idB) || (tC && idC == idB));
                const bool addC = !((tA && idA == idC) || (tB && idB == idC));
                const bool addD = true;
#else
                bool addA = false, addB = false, addC = false, addD = false;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    uint32_t dset = qD.w[w];
                    if ((n >> 5) == (uint32_t)w) dset |= 1u << (n & 31u);
                    const uint32_t known = (tA ? qA.w[w] : 0u) | (tB ? qB.w[w] : 0u) | (tC ? qC.w[w] : 0u) | (tD ? dset : 0u);
                    addA = addA || (qA.w[w] & ~known); addB = addB || (qB.w[w] & ~known); addC = addC || (qC.w[w] & ~known); addD = addD || (dset & ~known);
                }
#endif
                pdA = pdA && addA; pdB = pdB && addB; pdC = pdC && addC; pdD = pdD && addD;
            }
            if (__any(pdA || pdB || pdC || pdD)) {   // rare: a long alternative run onto the furthest wave's diagonal
                auto slow = [&](bool pd, int32_t oX) -> bool {
                    const uint32_t g = pd ? (uint32_t)(omax - oX) : 0u;
                    const uint32_t mr = w3_match_rest<G>(B.seq, nb, readp, (uint32_t)(pd ? oX : 0), pd ? d + oX : 0, g, pd ? 8u : 0u, !pd, run, gbase, gl);
                    return pd && mr == g;
                };
                const bool yA = slow(pdA, oA), yB = slow(pdB, oB), yC = slow(pdC, oC), yD = slow(pdD, 0);
                tA = tA || yA; tB = tB || yB; tC = tC || yC; tD = tD || yD;
            }
        }
        W3T(5);
        // ---- capped-diagonal set: is (n, d) recorded? ---------------------------------------------------------------------------
        const bool rec_live = crec.x == tag;
        const uint32_t rel = (uint32_t)(d - (int32_t)crec.y + 32);
        const bool in_win = rec_live && rel < 64u;
        bool capped = in_win && (((rel < 32u ? crec.z : crec.w) >> (rel & 31u)) & 1u);
        bool hfull = false;
        const bool use_hash = has && rec_live && !in_win;
        if (__any(use_hash)) {
            if (use_hash) {
                const uint64_t key = ((uint64_t)tag << 32) | ((uint64_t)(n & 0x3FFu) << 18) | (uint64_t)((uint32_t)d & 0x3FFFFu);
                uint32_t hp = (n * 0x9E3779B1u + (uint32_t)d) & hmask, probes = 0;
                uint64_t e = htab[hp];
#pragma clang loop unroll(disable)
                while (e != key && (uint32_t)(e >> 32) == tag) {
                    if (++probes > 24u) { hfull = true; break; }
                    hp = (hp + 1u) & hmask;
                    e = htab[hp];
                }
                capped = e == key;
            }
        }
        // ---- decide (wfa_graph.rs:463-474) -------------------------------------------------------------------------------------
        const int32_t pos_end = has ? d + (int32_t)E : 0;
        const int32_t cap = min((int32_t)len, (int32_t)other_len - d);
        const bool is_final = has && n == last && E == len && (uint32_t)pos_end == other_len;
        uint32_t kind = W2_KIND_NONE;
        bool ins = false, counts_far = false;
        if (has) {
            const bool skip = (capped && (int32_t)E < cap) || ((uint32_t)pos_end < min_prog);
            if (!skip) {
                counts_far = true;
                ins = (int32_t)E == cap && !capped && !hfull;
                if (E == len) {
                    if (n == last) { if ((uint32_t)pos_end < other_len) kind = W2_KIND_END_LAST; }
                    else kind = W2_KIND_FINISHED;
                } else kind = ((uint32_t)pos_end < other_len) ? W2_KIND_INTERIOR_READ : W2_KIND_INTERIOR;
            }
        }
#if W3_PREFETCH
        // ---- next round's bases: a wave that ran a long stretch and stopped inside node and read is the alignment's front; next round
        // it goes on right behind the mismatch (on this diagonal, or one beside it: the same lines). Those lines are asked for NOW -
        // two loads per sequence, 128 bytes apart, into registers nobody reads - so that next round's extension (two dependent trips
        // to memory: the first 16 bytes, then the cooperative 256) finds them in L2 instead of HBM. ----
        {
            const bool ahead = kind == W2_KIND_INTERIOR_READ && E >= (uint32_t)omax + 16u;
            if (ahead) {
                const uint8_t* a = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(nseq + E + 1u) & ~(uintptr_t)3);
                const uint8_t* b = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(readp + pos_end + 1) & ~(uintptr_t)3);
                pf0 = *reinterpret_cast<const uint32_t*>(a); pf1 = *reinterpret_cast<const uint32_t*>(a + 128);
                pf2 = *reinterpret_cast<const uint32_t*>(b); pf3 = *reinterpret_cast<const uint32_t*>(b + 128);
            }
        }
#endif
        W3T(6);
        // ---- commit: the slots of nodes below the first child of every node that finished in this tile (a prefix of the tile) ----
        // (round 5: by KEY, not by node - everything below the smallest target a wave that finished in this tile hands to a child is
        // final: what could still reach a slot comes from a parent's slot with a smaller key, and those are all in this prefix or
        // before it, their own children's targets at or beyond that smallest key. 2.7 % fewer tiles than the cut at the child's node.)
        const bool fin = kind == W2_KIND_FINISHED;
        const int32_t tdl = d + (int32_t)len;   // the diagonal a wave that finished its node hands to the node's children
        if (w3_gb<G>(fin && (tdl <= -W2_DIAG_LIM || tdl >= W2_DIAG_LIM), gbase)) status = W2_ST_NEED_BIG, why = 8u;
        const uint32_t X = (uint32_t)w2_gmin<G>(fin ? (int32_t)w3_key(nd.w & 0xFFFFu, tdl) : 0x7FFFFFFF);
        const bool commit = act && tgt.x < X;
        W3C(4, __popcll(__ballot(has))); W3C(5, __popcll(__ballot(commit))); W3C(10, __popcll(__ballot(act && !commit)));
        const uint32_t ncommit = (uint32_t)__popc(w3_gb<G>(commit, gbase));
        const bool live_k = commit && kind != W2_KIND_NONE && !fin, fin_k = commit && fin;
        const uint32_t ml = w3_gb<G>(live_k, gbase), mf = w3_gb<G>(fin_k, gbase);
        const uint32_t lpos = nl + w3_below(ml, lmask), fpos = SL - 1u - (nf + w3_below(mf, lmask));
        const uint32_t nlive = (uint32_t)__popc(ml), nfin = (uint32_t)__popc(mf);
#if W3_SETIDS
        if (run && nl + nlive > SL) status = W2_ST_NEED_BIG, why = 8u;
#else
        if (run && nl + nlive + nf + nfin > SL) status = W2_ST_NEED_BIG, why = 8u;
#endif
        if (__any(hfull && commit)) { if (w3_gb<G>(hfull && commit, gbase)) status = W2_ST_NEED_BIG, why = 9u; }
        const bool ok = status == W2_ST_PENDING;
#if W3_SETIDS
        // ---- this round's slots: offset | kind | the ring index of the union of the tied sets. One tied candidate (or several that name
        // one entry): its index travels on. A wave taken up from a parent, or tied candidates with different entries: a new entry. ----
        uint32_t bid = tA ? idA : tB ? idB : idC;
        W2Set<W> made = w2_set0<W>();
        bool words = false;   // this lane holds the words of its wave's set in `made`
        {
            // (bid is the first tied candidate's entry: the others differ from IT or from nothing)
            const bool differ = (tB && idB != bid) || (tC && idC != bid);
            // (a wave that ends the alignment keeps no slot - its set only goes into the job's result - but it may be one that was taken up
            // from a parent in this very tile: it needs the words, not an entry)
            const bool kept = kind != W2_KIND_NONE;   // (commit && kept = live_k || fin_k)
            words = ok && commit && (kept || is_final) && (tD || differ);
            const bool need = words && kept;
            // the usual new entry - a wave taken up from its parents and tied with nothing else: the parents' sets are in hand
            const bool slow = words && (differ || (tD && (tA || tB || tC)));   // (a wave taken up at offset 0 ties with a further one when it matches the read up to there)
            if (words && !slow) {
#pragma unroll
                for (int w = 0; w < W; ++w) made.w[w] = qD.w[w] | ((n >> 5) == (uint32_t)w ? 1u << (n & 31u) : 0u);   // + the node itself (wfa_graph.rs:535-541)
            }
            if (__any(slow)) {   // rare: tied waves whose sets differ - their words are fetched here
                if (slow) {
                    auto add = [&](uint32_t idx) { const W2Set<W> t = w2_ldset<W>(gs + (size_t)idx * W, true);
#pragma unroll
                                                   for (int w = 0; w < W; ++w) made.w[w] |= t.w[w]; };
                    if (tA) add(idA);
                    if (tB && !(tA && idB == idA)) add(idB);
                    if (tC && !(tA && idC == idA) && !(tB && idC == idB)) add(idC);
                    if (tD) {
#pragma unroll
                        for (int w = 0; w < W; ++w) made.w[w] |= qD.w[w] | ((n >> 5) == (uint32_t)w ? 1u << (n & 31u) : 0u);
                    }
                }
            }
            {
                const uint32_t nm = w3_gb<G>(need, gbase);
                const uint32_t at = ring_next + w3_below(nm, lmask);
                ring_next += (uint32_t)__popc(nm);
                if (ring_next > CAP) status = W2_ST_NEED_BIG, why = 8u;   // (group-uniform: the job has used up its arena - it is handed on like one that outgrows its slot lists)
                if (need && ring_next <= CAP) { w2_stset<W>(gs + (size_t)at * W, made); bid = at; }
            }
        }
        const bool ok2 = status == W2_ST_PENDING;
        if (ok2 && live_k) A[cbase + lpos] = make_uint2(tgt.x, (E << 3) | kind | (bid << 21));
#else
        // ---- this round's slots: offset | kind and the union of the tied sets -------------------------------------------------
        W2Set<W> best;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint32_t dset = qD.w[w];
            if (hinj && (n >> 5) == (uint32_t)w) dset |= 1u << (n & 31u);   // best + the successor (wfa_graph.rs:535-541)
            best.w[w] = (tA ? qA.w[w] : 0u) | (tB ? qB.w[w] : 0u) | (tC ? qC.w[w] : 0u) | (tD ? dset : 0u);
        }
        if (ok && live_k) {
            A[cbase + lpos] = make_uint2(tgt.x, (E << 3) | kind);
            w2_stset<W>(gs + (size_t)(cbase + lpos) * W, best);
        }
        if (ok && fin_k) w2_stset<W>(gs + (size_t)(cbase + fpos) * W, best);
#endif
        if (commit) {
            lane_upd += has ? 1u : 0u;
            if (counts_far && (uint32_t)pos_end > lane_far) lane_far = (uint32_t)pos_end;
        }
        W3T(7);
        // ---- record newly capped diagonals (committed slots only; the lanes of one node share its record) ----------------------
        bool later = false;
        {
            bool pend_ins = commit && ins && ok;
            while (__any(pend_ins)) {
                const uint32_t im = w3_gb<G>(pend_ins, gbase);
                if (im) {   // (group-uniform)
                    const uint32_t L = (uint32_t)__builtin_ctz(im);
                    const uint32_t nL = w2_gsel<G>(n, gl, L);
                    const uint32_t liveL = w2_gsel<G>(rec_live ? 1u : 0u, gl, L);
                    const int32_t anchor = (int32_t)w2_gsel<G>(liveL ? crec.y : (uint32_t)d, gl, L);
                    const bool mine = pend_ins && n == nL;
                    const uint32_t r2 = (uint32_t)(d - anchor + 32);
                    const bool inw = mine && r2 < 64u;
                    later = later || (mine && !inw);
                    const uint32_t lo = w2_gor<G>(inw && r2 < 32u ? 1u << r2 : 0u), hi = w2_gor<G>(inw && r2 >= 32u ? 1u << (r2 - 32u) : 0u);
                    if (gl == L) *reinterpret_cast<uint4*>(gs + C::SET_DWORDS + 4u * n) = make_uint4(tag, (uint32_t)anchor, (rec_live ? crec.z : 0u) | lo, (rec_live ? crec.w : 0u) | hi);
                    pend_ins = pend_ins && !mine;
                }
            }
        }
        if (__any(later)) {
            while (__any(later)) {
                const uint64_t lm = __ballot(later);
                const int L = __builtin_ctzll(lm);
                if ((int)lane == L) {
                    const uint64_t key = ((uint64_t)tag << 32) | ((uint64_t)(n & 0x3FFu) << 18) | (uint64_t)((uint32_t)d & 0x3FFFFu);
                    uint32_t hp = (n * 0x9E3779B1u + (uint32_t)d) & hmask, probes = 0;
                    uint64_t e = htab[hp];
#pragma clang loop unroll(disable)
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
            if (__any(hfull && commit)) { if (w3_gb<G>(hfull && commit, gbase)) status = W2_ST_NEED_BIG, why = 9u; }
        }
        W3T(8);
        // ---- finals (wfa_graph.rs:576-629): every wave of the last node that consumed node and read ---------------------------
#if W3_SETIDS
        if (__any(is_final && commit)) {
            const bool gf = w3_gb<G>(is_final && commit, gbase) != 0;
            W2Set<W> fs = made;   // (a lane that has just put its set together holds the words)
            if (is_final && commit && !words) fs = w2_ldset<W>(gs + (size_t)bid * W, true);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t o = w2_gor<G>(is_final && commit ? fs.w[w] : 0u);
                if (gf && gl == 0) outset[w] |= o;
            }
            if (gf) final_found = true;
        }
#else
        if (__any(is_final && commit)) {
            const bool gf = w3_gb<G>(is_final && commit, gbase) != 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t o = w2_gor<G>(is_final && commit ? best.w[w] : 0u);
                if (gf && gl == 0) outset[w] |= o;
            }
            if (gf) final_found = true;
        }
#endif
        W3T(9);
        // ---- the round's lists move on; finished waves become targets of their node's children --------------------------------
#if W3_PROF
        ++w3steps;
#endif
        if (run) {
#if W3_SETIDS
            nl += nlive; ip += ncommit;   // (finished waves hold no slot: their children's targets carry the index of their set)
#else
            nl += nlive; nf += nfin; ip += ncommit;
#endif
            // ---- finished waves become targets (child, diagonal + node length) of this round, kept sorted ----
            // one target: key K, the finished wave's set `si` (group-uniform)
            auto insert_slow = [&](const uint32_t K, const uint32_t si) {
                W3C(12, 1);
                // (the general road: any list length, a third wave onto one target, no space below the targets)
                // where K belongs among the targets still to come: [ip, np) is sorted
                uint32_t pos = ip, hit = 0xFFFFFFFFu;
#pragma clang loop unroll(disable)
                for (uint32_t base = ip; base < np; base += (uint32_t)G) {
                    W3C(13, 1);
                    const uint32_t i = base + gl;
                    const uint32_t k = i < np ? A[cbase + i].x : 0xFFFFFFFFu;
                    const uint32_t lt = w3_gb<G>(k < K, gbase), eq = w3_gb<G>(k == K, gbase);
                    const uint32_t nlt = (uint32_t)__popc(lt);
                    pos += nlt;
                    if (eq) { hit = base + (uint32_t)__builtin_ctz(eq); break; }
                    if (nlt < (uint32_t)G) break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (hit != 0xFFFFFFFFu) {
                    const uint32_t y = A[cbase + hit].y;
                    uint32_t ny = y;
                    if (((y >> 10) & 0x3FFu) == W3_NONE) ny = (y & ~(0x3FFu << 10)) | (si << 10);
                    else if (((y >> 20) & 0x3FFu) == W3_NONE) ny = (y & ~(0x3FFu << 20)) | (si << 20);
                    else {
                        // a third wave onto one target (rare): its set and the second one's merge into a fresh entry of the arena
#if W3_SETIDS
                        if (ring_next >= CAP) { status = W2_ST_NEED_BIG, why = 8u; return; }
                        const uint32_t sm = ring_next++;
#else
                        if (nl + nf + 1u > SL) { status = W2_ST_NEED_BIG, why = 8u; return; }
                        const uint32_t sm = SL - 1u - nf;
                        ++nf;
#endif
                        const uint32_t s1 = (y >> 20) & 0x3FFu;
#if W3_SETIDS
                        if (gl < (uint32_t)W) gs[(size_t)sm * W + gl] = gs[(size_t)s1 * W + gl] | gs[(size_t)si * W + gl];
#else
                        if (gl < (uint32_t)W) gs[(size_t)(cbase + sm) * W + gl] = gs[(size_t)(cbase + s1) * W + gl] | gs[(size_t)(cbase + si) * W + gl];
#endif
                        ny = (y & ~(0x3FFu << 20)) | (sm << 20);
                    }
                    if (gl == 0) A[cbase + hit].y = ny;
                } else {
                    if (np >= SL) { status = W2_ST_NEED_BIG, why = 8u; return; }
                    // shift [pos, np) up by one, from the top down, G entries at a time
#pragma clang loop unroll(disable)
                    for (uint32_t top = np; top > pos;) {
                        W3C(14, 1);
                        const uint32_t lo = top - pos > (uint32_t)G ? top - (uint32_t)G : pos;
                        const uint32_t i = lo + gl;
                        uint2 v = make_uint2(0, 0);
                        if (i < top) v = A[cbase + i];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (i < top) A[cbase + i + 1u] = v;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        top = lo;
                    }
                    if (gl == 0) A[cbase + pos] = make_uint2(K, w3_aux(W3_NONE, si, W3_NONE));
                    ++np;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            };
            // One target K with the finished wave's set `si` (group-uniform), ONE code shape for the three things that happen to it
            // (round 5; the one-at-a-time roads above cost a quarter of a step because the eight groups of a wavefront took
            // different ones): the next G targets still to come are looked at once, a lane each;
            //   * K is there: the wave joins that target (second source);
            //   * K lies beyond everything listed (the front of the alignment): appended;
            //   * K belongs among those G: the targets BELOW it move one slot DOWN into the space the consumed targets left
            //     (ip - 1 >= nl: the live list grows from 0 by at most one entry per consumed target) - a child's target lands
            //     right behind the tile that finished its parent, so this moves one or two entries where shifting the list's
            //     tail up moved all of it;
            // anything else - the key further up a long list, a third wave onto one target, no space below - takes the general road.
            auto insert = [&](const uint32_t K, const uint32_t si) {
                W3C(9, 1);
                const uint32_t rem = np - ip;
                const bool in = gl < rem;
                const uint32_t i = cbase + ip + gl;
                const uint2 v = in ? A[i] : make_uint2(0xFFFFFFFFu, 0u);
                const bool mine = v.x == K;
                const uint32_t free0 = ((v.y >> 10) & 0x3FFu) == W3_NONE ? 1u : 0u, free1 = ((v.y >> 20) & 0x3FFu) == W3_NONE ? 1u : 0u;
                const uint32_t lt = w3_gb<G>(v.x < K, gbase), eq = w3_gb<G>(mine, gbase), jn = w3_gb<G>(mine && (free0 | free1) != 0u, gbase);
                const uint32_t pos = (uint32_t)__popc(lt);
                const bool app = rem == 0u || K > lastkey;
                const bool join = jn != 0u;
                const bool front = !eq && !app && pos < (uint32_t)G && ip > nl;
                const bool full = !eq && app && np >= SL;
                const bool slow = (eq && !join) || (!eq && !app && !front);
                if (join && mine) A[i].y = free0 ? ((v.y & ~(0x3FFu << 10)) | (si << 10)) : ((v.y & ~(0x3FFu << 20)) | (si << 20));
                if (front && gl < pos) A[i - 1u] = v;
                if ((front || (app && !eq && !full)) && gl == 0) A[front ? cbase + ip + pos - 1u : cbase + np] = make_uint2(K, w3_aux(W3_NONE, si, W3_NONE));
                if (front) --ip;
                if (app && !eq && !full) { ++np; lastkey = K; }
                if (full) status = W2_ST_NEED_BIG, why = 8u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (slow) insert_slow(K, si);
            };
            W3T(11);
            const uint32_t fm = status == W2_ST_PENDING ? mf : 0u;
            if (fm) {   // (group-uniform) a wave finished its node in this tile
                W3C(15, __popc(fm));
                // every finished lane puts the targets of its node's first two children (they come with the node's descriptor) into the
                // group's scratch itself - no broadcast per lane; then one pass per target
                const uint32_t nch = nd.z & 0xFFFFu;
                const bool f2 = fin_k && nch >= 2u;
                const uint32_t m2 = w3_gb<G>(f2, gbase);
                const uint32_t qpos = w3_below(fm, lmask) + w3_below(m2, lmask);
                const uint32_t m = (uint32_t)(__popc(fm) + __popc(m2));
                uint2* qbuf = reinterpret_cast<uint2*>(R + C::O_Q);
#if W3_SETIDS
                const uint32_t fset = bid;    // the finished wave's set: what its children's targets name
#else
                const uint32_t fset = fpos;
#endif
                if (fin_k) {
                    qbuf[qpos] = make_uint2(w3_key(nd.w & 0xFFFFu, tdl), fset);
                    if (f2) qbuf[qpos + 1u] = make_uint2(w3_key(nd.w >> 16, tdl), fset);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // one pass per target. (Measured and left out, round 4: all of a tile's new targets merged into the list's last G entries
                // at once by rank - a broadcast read and two ballots per key. A third of the tiles fell outside what that handles -
                // a key before the look, two waves onto one new target - and the one-at-a-time road they then took on top of the
                // attempt made the class kernels 15 % slower: 29 against 25 ms.)
#pragma clang loop unroll(disable)
                for (uint32_t k = 0; k < m && status == W2_ST_PENDING; ++k) {
                    const uint2 q = qbuf[k];
                    insert(q.x, q.y);
                }
                // a third and later child (a node several variants reconnect at) is read from the overflow list in HBM: per lane, rare.
                // (Kept apart from the common case: a load on the way to it made every step wait for its own set stores - vmcnt
                // counts loads and stores in one order.)
#ifndef W3_NO_OVERFLOW_CHILDREN   // (experiment switch: timing without the rare road's loads)
                uint32_t m3 = w3_gb<G>(fin_k && nch > 2u, gbase);
#pragma clang loop unroll(disable)
                while (m3 && status == W2_ST_PENDING) {
                    const uint32_t L = (uint32_t)__builtin_ctz(m3);
                    m3 &= m3 - 1u;
                    const uint32_t qn = w2_gsel<G>(n, gl, L), qz = w2_gsel<G>(nd.z, gl, L);
                    const int32_t td = (int32_t)w2_gsel<G>((uint32_t)tdl, gl, L);
#if W3_SETIDS
                    const uint32_t si = w2_gsel<G>(bid, gl, L);
#else
                    const uint32_t si = w2_gsel<G>(fpos, gl, L);
#endif
                    uint32_t scan = qz >> 16;
#pragma clang loop unroll(disable)
                    for (uint32_t j = 2; j < (qz & 0xFFFFu) && status == W2_ST_PENDING; ++j) insert(w3_key(w2_next_child(gedge, qn, scan), td), si);
                }
#endif
            }
            if (ip >= np) state = S_ROUND;
            if (++steps > W2_MAX_STEPS) status = W2_ST_INTERNAL;
            if (status != W2_ST_PENDING) state = S_JOB;
        }
        W3T(10);
    }
#if W3_PROF
    if (lane == 0 && (blockIdx.x % 61) == 3)
        printf("w3prof G%d W%d wg %u: total %llu steps %u | between %llu control %llu candidates %llu issue %llu extension %llu ties %llu capped-decide %llu commit-write %llu capins %llu finals %llu inject %llu pre-inject %llu\n", G, W, blockIdx.x,
               (unsigned long long)(w3tl - w3t0), w3steps, (unsigned long long)w3t[0], (unsigned long long)w3t[1], (unsigned long long)w3t[2], (unsigned long long)w3t[3], (unsigned long long)w3t[4],
               (unsigned long long)w3t[5], (unsigned long long)w3t[6], (unsigned long long)w3t[7], (unsigned long long)w3t[8], (unsigned long long)w3t[9], (unsigned long long)w3t[10], (unsigned long long)w3t[11]);
#endif
    if (B.esc_role == 1u) {   // a producer workgroup is gone (everything it hands over has been published)
        W2_WAIT_VM();
        (void)atomicAdd(B.esc + 2, lane == 0 ? 1u : 0u);
    }
#if W3_STATS
    if (lane == 0 && (blockIdx.x % 61) == 3)
        printf("w3 G%d W%d wg %u: wave-steps %llu control-passes %llu group-tiles %llu act %llu has %llu committed %llu jobs %llu rounds %llu long-ext-lanes %llu inserts %llu discarded-lanes %llu build-chunks %llu slow-inserts %llu scan-iters %llu shift-iters %llu finished-waves %llu\n", G, W, blockIdx.x,
               (unsigned long long)w3c[0], (unsigned long long)w3c[1], (unsigned long long)w3c[2], (unsigned long long)w3c[3], (unsigned long long)w3c[4], (unsigned long long)w3c[5],
               (unsigned long long)w3c[6], (unsigned long long)w3c[7], (unsigned long long)w3c[8], (unsigned long long)w3c[9], (unsigned long long)w3c[10], (unsigned long long)w3c[11],
               (unsigned long long)w3c[12], (unsigned long long)w3c[13], (unsigned long long)w3c[14], (unsigned long long)w3c[15]);
#endif
}

}  // namespace hp
