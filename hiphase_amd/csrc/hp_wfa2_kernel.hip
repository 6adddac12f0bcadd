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
//   * the reference's max_wavefronts map (:360, :464-470) is replaced by the set of CAPPED diagonals (a tagged 16-byte
//     record per node in HBM, a tagged hash set for the odd diagonal far from a node's others): a wave on (node, d) is stale <=> (node, d) once reached cap = min(node length,
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
#ifndef W2_PROF
#define W2_PROF 0
#endif
#ifndef W2_STATS
#define W2_STATS 0
#endif
#ifndef W2_QUEUE_ASM
#define W2_QUEUE_ASM 0
#endif
#ifndef W2_MATCH_LANE_BYTES
#define W2_MATCH_LANE_BYTES 32
#endif
#if W2_PROF
#define W2PT(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); w2pc[i] += t_ - w2tl; w2tl = t_; } while (0)
#define W2PC(i, v) do { w2pn[i] += (v); } while (0)
#else
#define W2PT(i)
#define W2PC(i, v)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define W2_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")   // the atomics above have returned (= been performed)
#else
#define W2_WAIT_VM()
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char w2_smem[];

W2DEV uint32_t w2_lane() { return __lane_id(); }
W2DEV uint64_t w2_ld8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// ---- 16-byte windows -------------------------------------------------------------------------------------------------
W2DEV uint4 w2_ld16(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
// common prefix (0..16) of two 16-byte windows
W2DEV uint32_t w2_pfx16(const uint4& a, const uint4& b) {
    const uint32_t x0 = a.x ^ b.x, x1 = a.y ^ b.y, x2 = a.z ^ b.z, x3 = a.w ^ b.w;
    if (x0) return (uint32_t)__builtin_ctz(x0) >> 3;
    if (x1) return 4u + ((uint32_t)__builtin_ctz(x1) >> 3);
    if (x2) return 8u + ((uint32_t)__builtin_ctz(x2) >> 3);
    if (x3) return 12u + ((uint32_t)__builtin_ctz(x3) >> 3);
    return 16u;
}
// 8-byte windows for the tie checks (the gap to the furthest candidate is one or two bytes almost always; 16-byte windows
// for the four of them kept 16 more registers live through the extension)
struct W2Pre8 { uint64_t a, b; };
W2DEV W2Pre8 w2_pre8(const uint8_t* a, const uint8_t* b, bool on) {
    W2Pre8 p; p.a = 0; p.b = 0;
    if (on) { p.a = w2_ld8(a); p.b = w2_ld8(b); }
    return p;
}
W2DEV uint32_t w2_pfx8(uint64_t a, uint64_t b) { const uint64_t x = a ^ b; return x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u; }
struct W2Pre { uint4 a, b; };
W2DEV W2Pre w2_pre(const uint8_t* a, const uint8_t* b, bool on) {
    W2Pre p;
    p.a = make_uint4(0, 0, 0, 0); p.b = p.a;
    if (on) { p.a = w2_ld16(a); p.b = w2_ld16(b); }
    return p;
}

// ---- group collectives (G consecutive lanes; every lane of the group must be active) -------------------------------
template <int G> W2DEV uint64_t w2_gballot(bool pred, uint32_t gbase) {
    const uint64_t b = __ballot(pred);
    if (G == 64) return b;
    return (b >> gbase) & ((1ull << (G & 63)) - 1ull);
}
// Reductions over the G lanes of a group on the vector ALU (DPP lane exchanges; no trip through the LDS crossbar as
// __shfl would take): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror (i <-> 7-i), row_mirror (i <-> 15-i).
// `old` = the lane's own value, so a lane whose partner is masked off just keeps what it has.
#define W2_DPP(v, ctrl) __builtin_amdgcn_update_dpp((int)(v), (int)(v), (ctrl), 0xF, 0xF, false)
template <int G> W2DEV int32_t w2_gmax(int32_t v) {
    v = max(v, W2_DPP(v, 0xB1)); v = max(v, W2_DPP(v, 0x4E)); v = max(v, W2_DPP(v, 0x141));
    if (G >= 16) v = max(v, W2_DPP(v, 0x140));
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
template <int G> W2DEV int32_t w2_gmin(int32_t v) {
    v = min(v, W2_DPP(v, 0xB1)); v = min(v, W2_DPP(v, 0x4E)); v = min(v, W2_DPP(v, 0x141));
    if (G >= 16) v = min(v, W2_DPP(v, 0x140));
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
template <int G> W2DEV uint32_t w2_gor(uint32_t v) {
    v |= (uint32_t)W2_DPP(v, 0xB1); v |= (uint32_t)W2_DPP(v, 0x4E); v |= (uint32_t)W2_DPP(v, 0x141);
    if (G >= 16) v |= (uint32_t)W2_DPP(v, 0x140);
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v |= (uint32_t)__shfl_xor((int)v, m);
    return v;
}
template <int G> W2DEV uint32_t w2_gsum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    if (G >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v += (uint32_t)__shfl_xor((int)v, m);
    return v;
}
// the value lane `L` of the group holds (L group-uniform)
template <int G> W2DEV uint32_t w2_gsel(uint32_t v, uint32_t gl, uint32_t L) { return w2_gor<G>(gl == L ? v : 0u); }
// Lanes with done == false have matched their first n bytes of node[o..] against read[pos..] and may match up to
// maxlen: the group serves them one after the other, G x 32 bytes per step. Only the serving lane's two offsets travel
// (the sequences' base addresses are group-uniform). `on`: this group takes part (group-uniform).
template <int G> W2DEV uint32_t w2_match_rest(const uint8_t* nseq, const uint8_t* readp, uint32_t o, int32_t pos, uint32_t maxlen, uint32_t n, bool done,
                                              bool on, uint32_t gbase, uint32_t gl) {
    // bytes a lane compares per pass (-DW2_MATCH_LANE_BYTES). A pass is one dependent round trip to memory for the whole wavefront
    // (every group waits for the slowest), and 88 % of the steps have one: HiFi reads match ~200 bases between two errors, so with
    // 32 bytes a lane (256 a pass) a quarter of the served extensions need a second pass. 64 bytes a lane would leave 7 % - but
    // the sixteen more registers spill (18-27 of them at three wavefronts per SIMD) and the launch set gets SLOWER: 43 vs 38 ms.
    constexpr uint32_t LB = W2_MATCH_LANE_BYTES;
    bool pending = on && !done;
    while (__any(pending)) {
        const uint64_t gb = w2_gballot<G>(pending, gbase);
        const bool active = gb != 0;
        const uint32_t L = active ? (uint32_t)__builtin_ctzll(gb) : 0u;
        const uint32_t so = w2_gsel<G>(o + n, gl, L), sp = w2_gsel<G>((uint32_t)pos + n, gl, L), rem = w2_gsel<G>(maxlen - n, gl, L);
        const uint8_t* pa = nseq + so;
        const uint8_t* pb = readp + sp;
        const uint32_t off = gl * LB;
        uint32_t m = LB;
        if (active) {
            if (off < rem) {
                uint4 a[LB / 16], b[LB / 16];
#pragma unroll
                for (uint32_t k = 0; k < LB / 16; ++k) { a[k] = w2_ld16(pa + off + 16 * k); b[k] = w2_ld16(pb + off + 16 * k); }
                m = 0;
#pragma unroll
                for (uint32_t k = 0; k < LB / 16; ++k) if (m == 16u * k) m += w2_pfx16(a[k], b[k]);
                if (m > rem - off) m = rem - off;
            } else m = 0u;   // beyond the end: acts as a stop
        }
        const uint64_t stop = w2_gballot<G>(active && m < LB, gbase);
        uint32_t got = (uint32_t)G * LB;
        if (stop) {
            const uint32_t S = (uint32_t)__builtin_ctzll(stop);
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

template <int W> struct W2Set { uint32_t w[W]; };
template <int W> W2DEV W2Set<W> w2_set0() { W2Set<W> s; for (int i = 0; i < W; ++i) s.w[i] = 0; return s; }
template <int W> W2DEV W2Set<W> w2_ldset(const uint32_t* p, bool on) {   // 16-byte aligned, global memory
    W2Set<W> s = w2_set0<W>();
    if (on) {
        if (W == 2) { const uint2 v = *reinterpret_cast<const uint2*>(p); s.w[0] = v.x; s.w[1] = v.y; }
        else {
#pragma unroll
            for (int i = 0; i < W; i += 4) { const uint4 v = *reinterpret_cast<const uint4*>(p + i); s.w[i] = v.x; s.w[i + 1] = v.y; s.w[i + 2] = v.z; s.w[i + 3] = v.w; }
        }
    }
    return s;
}
template <int W> W2DEV void w2_stset(uint32_t* p, const W2Set<W>& s) {
    if (W == 2) *reinterpret_cast<uint2*>(p) = make_uint2(s.w[0], s.w[1]);
    else {
#pragma unroll
        for (int i = 0; i < W; i += 4) *reinterpret_cast<uint4*>(p + i) = make_uint4(s.w[i], s.w[i + 1], s.w[i + 2], s.w[i + 3]);
    }
}

// =====================================================================================================================
template <int G, int W, bool WIDE = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W2Cfg<W>::WAVES_PER_SIMD, W2Cfg<W>::WAVES_PER_SIMD))) hp_wfa2_kernel(W2Batch B) {
    using C = W2Cfg<W, WIDE>;
    static_assert(G >= 8 && G <= 64 && (G & (G - 1)) == 0, "group size");
    static_assert(C::MAXQ <= G, "one lane per pending-queue entry");
    constexpr uint32_t NG = 64 / G;
    const uint32_t lane = w2_lane(), gid = lane / G, gl = lane % G, gbase = gid * G;
    unsigned char* R = w2_smem + (size_t)gid * C::BYTES;
    uint4* live = reinterpret_cast<uint4*>(R + C::O_LIVE);          // [parity * MAXL + i]
    uint4* fin = reinterpret_cast<uint4*>(R + C::O_FIN);            // [i]
    uint32_t* ek = reinterpret_cast<uint32_t*>(R + C::O_EK);        // [parity * SLOTS + s]
    uint32_t* outset = reinterpret_cast<uint32_t*>(R + C::O_MISC);
    int2* srcs = reinterpret_cast<int2*>(R + C::O_SRC);

    const uint32_t slot = blockIdx.x * NG + gid;
    uint64_t* htab = B.htab + ((size_t)slot << B.hcap_log2);
    uint32_t* gs = B.gsets + (size_t)slot * B.set_stride;           // [(parity * SLOTS + s) * W]
    const uint32_t hmask = (1u << B.hcap_log2) - 1u;
    const uint32_t prune32 = B.prune_distance > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)B.prune_distance;
    const uint32_t maxed32 = B.max_ed > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (uint32_t)B.max_ed;
    const uint8_t* altp = B.seq + B.alt_off;

    // ---- group-uniform state (every lane of a group holds the same value) ------------------------------------------
    enum : uint32_t { S_JOB = 0, S_NEXT = 1, S_WAIT = 2, S_TILE = 3, S_DONE = 4 };   // S_WAIT: holds a ticket for a list position not yet published
    uint32_t state = S_JOB, jround = 0, job = 0, ticket = 0, idle_polls = 0;
    bool have_ticket = false;
    uint32_t poll_div = 0;
    uint32_t n_live_node = 0;   // live entries the current node has got this round
#if W2_STATS
    uint32_t mx_l = 0, mx_f = 0, mx_top = 0;   // how far a job fills its tables (sizing study, scripts/prof_wfa2.sh)
#endif
    uint32_t n_nodes = 0, last = 0, other_len = 0, tag = 0;
    const uint8_t* refp = altp; const uint8_t* readp = altp;
    const W2Node* gnode = B.nodes; const uint16_t* gedge = B.edges;   // the job's node / edge tables in HBM
    uint32_t ed = 0, c = 0, p = 1, lcnt_prev = 0, lcnt_cur = 0, fcnt = 0, top = 0, pp = 0;
    uint32_t farthest = 0, min_prog = 0;
    bool final_found = false, round_live = false;
    int32_t status = W2_ST_PENDING;
    uint32_t score = 0, steps = 0;
    uint4 ph = make_uint4(0xFFFFu, 0, 0, 0);   // header of the previous round's entry at pp (node 0xFFFF: none left)
    // nodes waiting for their turn with waves handed over by parents this round: lane i of the group keeps entry i in
    // registers (node 0xFFFF = free; up to MAXPAR parent-entry codes, one byte each)
    uint32_t pq_node = 0xFFFFu, pq_cnt = 0, pq_c0 = 0, pq_c1 = 0;
    // current node / item
    uint32_t n = 0, len = 0, child_off = 0, n_child = 0, c01 = 0, n_items = 0, item = 0, npar = 0, pc0 = 0, pc1 = 0;
    const uint8_t* nseq = altp;
    bool use_list = false;
    // previous round's live entries of n (NP of them at most: two for the smaller classes, four for the largest, which takes over
    // the reads whose paths lie far apart after a structural variant): live diagonals [pa, pb], slot of diagonal d = po + d
    constexpr int NP = C::MAXPREV;
    uint32_t np = 0;
    int32_t pa[NP], pb[NP], po[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { pa[k] = 0; pb[k] = 0; po[k] = 0; }
    int32_t lo = 0, hi = 0, base = 0;
    const uint32_t n_class = *B.n_items_dev;   // jobs of this graph-size class
    if (B.esc_role == 1u) (void)atomicAdd(B.esc + 4, lane == 0 ? 1u : 0u);   // a producer workgroup has started
    uint32_t coff = 0;
    int32_t clo = 0, chi = INT32_MIN, cvlo = INT32_MAX, cvhi = INT32_MIN, cflo = INT32_MAX, cfhi = INT32_MIN;   // cluster being formed
    uint32_t lane_far = 0;   // per lane
    uint32_t why = 0;        // which limit handed the job back (reported in place of the score; HP_DEBUG prints the histogram)
    uint32_t lane_upd = 0;   // per lane: (node, diagonal) wave updates of this job (work counter)

#if W2_PROF
    uint64_t w2pc[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; uint32_t w2pn[8] = {0,0,0,0,0,0,0,0};
    uint64_t w2tl = __builtin_amdgcn_s_memtime();
    const uint64_t w2t0 = w2tl;
#endif

    // hands the waves that finished entry `code` of the current node to child `cid` (group-uniform call)
    auto pq_append = [&](uint32_t cid, uint32_t code) {
        const uint64_t m = w2_gballot<G>(pq_node == cid, gbase);
        const uint64_t f = w2_gballot<G>(pq_node == 0xFFFFu && gl < (uint32_t)C::MAXQ, gbase);
        const uint32_t idx = m ? (uint32_t)__builtin_ctzll(m) : (f ? (uint32_t)__builtin_ctzll(f) : 0xFFu);
        bool over = idx == 0xFFu;
        if (gl == idx) {
            if (!m) { pq_node = cid; pq_cnt = 0; pq_c0 = 0; pq_c1 = 0; }
            if (pq_cnt >= (uint32_t)C::MAXPAR) over = true;
            else {
                if (pq_cnt < 4u) pq_c0 |= code << (8u * pq_cnt); else pq_c1 |= code << (8u * (pq_cnt - 4u));
                ++pq_cnt;
            }
        }
        if (w2_gballot<G>(over, gbase)) status = W2_ST_NEED_BIG, why = 1u;
    };
    // publishes the cluster [clo, chi] of the current item as an entry of this round (group-uniform)
    auto emit_cluster = [&]() {
        const bool lv = cvlo <= cvhi, fn = cflo <= cfhi;
        uint32_t code;
        uint4 h;
        h.x = n | ((coff + (uint32_t)(clo - lo)) << 16);
        h.y = (uint32_t)clo;
        h.z = (lv ? ((uint32_t)(cvlo - clo) | ((uint32_t)(cvhi - clo) << 8)) : 0x00FFu) | ((fn ? ((uint32_t)(cflo - clo) | ((uint32_t)(cfhi - clo) << 8)) : 0x00FFu) << 16);
        h.w = len;
        if (lv) {
            if (lcnt_cur >= (uint32_t)C::MAXL) { status = W2_ST_NEED_BIG, why = 2u; return; }
            code = lcnt_cur;
            if (gl == 0) live[c * C::MAXL + lcnt_cur] = h;
            lcnt_cur++;
            ++n_live_node;
            round_live = true;
#if W2_STATS
            mx_l = max(mx_l, lcnt_cur);
#endif
        } else {
            if (fcnt >= (uint32_t)C::MAXF) { status = W2_ST_NEED_BIG, why = 3u; return; }
            code = 128u + fcnt;
            if (gl == 0) fin[fcnt] = h;
            fcnt++;
#if W2_STATS
            mx_f = max(mx_f, fcnt);
#endif
        }
        if (fn)
#pragma clang loop unroll(disable)
            for (uint32_t j = 0, scan = child_off; j < n_child; ++j)
                pq_append(j == 0 ? (c01 & 0xFFFFu) : (j == 1 ? (c01 >> 16) : w2_next_child(gedge, n, scan)), code);
        chi = INT32_MIN; cvlo = INT32_MAX; cvhi = INT32_MIN; cflo = INT32_MAX; cfhi = INT32_MIN;
    };

    for (;;) {
        // ============================ 1. control: advance every group to its next tile ===============================
        uint32_t spins = 0;
        W2PT(0); W2PC(0, 1);
        if (B.esc_role == 2u && __any(state == S_WAIT) && (!__any(state == S_TILE) || (++poll_div & 15u) == 0u)) {   // (busy wavefronts look less often)
            // producers publish before they exit: a ticket still unpublished once every producer workgroup is gone stays so
            const uint32_t gone = atomicAdd(B.esc + 2, 0u);
            W2_WAIT_VM();
            const uint32_t pub = atomicAdd(B.esc + 1, 0u);
            // (a profiler or a shared hardware queue may run the kernels one after the other, this one first: when no producer
            // workgroup has so much as started after ~20 ms of idling, leave - what they hand over then stays for the host's pass)
            bool alone = false;
            if (idle_polls > 750u && gone == 0u) alone = atomicAdd(B.esc + 4, 0u) == 0u;
            if (state == S_WAIT) {
                if (pub > ticket) { state = S_JOB; have_ticket = true; }
                else if (gone >= B.esc_producers || alone) state = S_DONE;
            }
        }
        while (state != S_TILE && state != S_DONE && state != S_WAIT) {
            W2PC(1, 1);
            if (++spins > (1u << 20)) { state = S_DONE; break; }   // cannot happen; never hang the device
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (state == S_JOB) {
                if (status != W2_ST_PENDING) {   // results of the job that just ended
                    const uint32_t upd = w2_gsum<G>(lane_upd);
                    bool handed_over = false;
                    // Worth handing over: what the largest class's bigger tables fix (pending queue, live / finished-only entries,
                    // source intervals, slots) - not a full capped set or three entries of one node, which it shares. And only
                    // while that class has room: reads with a few per cent of noise fail by the thousand, and a few hundred
                    // groups working through them one after the other (to fail again) took a second; those go straight
                    // to the dense-band pass, as does everything past the budget.
                    const bool hand = status == W2_ST_NEED_BIG && (why == 1u || why == 2u || why == 3u || why == 6u || why == 7u || why == 8u);
                    if (B.esc_role == 1u && __any(hand)) {
                        // hand the job to the largest class: reserve a list position, store the job there, publish positions in
                        // order. Every lane takes part in every atomic (idle ones on a scratch word): no one-lane branches.
                        const uint32_t taken = atomicAdd(B.esc, 0u);
                        const bool me = hand && gl == 0 && taken < B.esc_limit;
                        const uint32_t pos = atomicAdd(B.esc, me ? 1u : 0u);
                        (void)atomicExch(me ? B.esc_order + pos : B.esc + 3, me ? job : 0u);
#if W2_STATS
                        if (me) B.handed[job] = (uint8_t)(1u + min(254u, ed / 2u));   // sizing study: the round it gave up in
#else
                        if (me) B.handed[job] = 1;   // (never written by anyone else: hp_wfa2_map_kernel)
#endif
                        W2_WAIT_VM();
                        bool pend = me;
                        for (uint32_t s = 0; s < (1u << 16) && __any(pend); ++s) {
                            const uint32_t seen = atomicCAS(pend ? B.esc + 1 : B.esc + 3, pend ? pos : 0xFFFFFFFFu, pend ? pos + 1u : 0xFFFFFFFFu);
                            if (pend && seen == pos) pend = false;
                        }
                        // its results are the consumer's to write (two XCDs' L2s must not both hold dirty copies of one word).
                        // A position that could not be published leaves the job PENDING: the host's dense-band pass takes it.
                        handed_over = w2_gballot<G>(me, gbase) != 0;
                    }
                    if (!handed_over && gl == 0) { B.status[job] = status; B.out_score[job] = status == W2_ST_NEED_BIG ? (uint64_t)(why | (ed << 8)) /* the limit it ran into, the round it was in */ : score; B.out_work[(size_t)job * 2] = upd; }
                    if (!handed_over && gl < (uint32_t)W) B.out_sets[(size_t)job * W2_SET_STRIDE + gl] = outset[gl];
#if W2_STATS
                    if (W <= 4 && gl == 0) { B.out_sets[(size_t)job * W2_SET_STRIDE + 6] = mx_l | (mx_f << 8) | (mx_top << 16); }
                    mx_l = 0; mx_f = 0; mx_top = 0;
#endif
                    status = W2_ST_PENDING;
                }
                if (B.group_jobs != 0u && B.esc_role != 2u && jround >= B.group_jobs) { state = S_DONE; break; }   // this group's share is done
                // dynamic work queue: the group's first lane pulls the next index. The instruction is written out by hand:
                // hipcc's atomic optimiser + structuriser turned an atomicAdd under a one-lane branch inside a persistent
                // loop into a wavefront that never came back (round 1, DESIGN.md bring-up notes)
#if W2_QUEUE_ASM
                uint32_t mine = 0;
                if (gl == 0) {
                    uint32_t* qp = B.next;
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of hipcc parses kernel bodies too; gfx950 assembly means nothing to it)
                    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(mine) : "v"(qp), "v"(1u) : "memory");
#else
                    (void)qp;
#endif
                }
#else
                // every lane of the group takes part (lane 0 adds one, the others nothing): no one-lane branch around the atomic
                const uint32_t mine = atomicAdd(B.next, (gl == 0 && !have_ticket) ? 1u : 0u);
#endif
                const uint32_t k = have_ticket ? ticket : w2_gsel<G>(mine, gl, 0u);
                jround++;
                if (B.esc_role == 2u) {
                    if (!have_ticket && k >= n_class) {   // past the class's own jobs: a ticket for a job another class may hand over
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
                refp = B.seq + jd.ref_off; readp = B.seq + jd.read_off;
                tag = B.tag_base + job + 1u;
                if (gl < (uint32_t)W) outset[gl] = 0u;
                score = 0;
                if (n_nodes == 0 || n_nodes > (uint32_t)C::MAXN || ji.n_edges > (uint32_t)C::MAXE || other_len >= (uint32_t)W2_DIAG_LIM) {
                    status = W2_ST_NEED_BIG, why = 4u;
                    continue;   // stays in S_JOB: the next pass writes this status and fetches the next job
                }
                gnode = B.nodes + jd.node_off;
                gedge = B.edges + jd.edge_off;
                // the start wave (wfa_graph.rs:366-378): node 0 waits for its turn in round 0, with no parent
                pq_node = gl == 0 ? 0u : 0xFFFFu; pq_cnt = 0; pq_c0 = 0; pq_c1 = 0;
                ed = 0; c = 0; p = 1; lcnt_prev = 0; lcnt_cur = 0; fcnt = 0; top = 0; pp = 0; steps = 0;
                ph = make_uint4(0xFFFFu, 0, 0, 0);
                farthest = 0; min_prog = 0; final_found = false; round_live = false; lane_far = 0; lane_upd = 0;
                n_items = 0; item = 0;
                state = S_NEXT;
                continue;
            }
            // ---- state == S_NEXT: the next item of the current node, or the next node, or the end of the round ----
            if (item >= n_items) {
                const uint32_t na = ph.x & 0xFFFFu;
                const uint32_t nb = (uint32_t)w2_gmin<G>((int32_t)pq_node);
                n = min(na, nb);
                if (n == 0xFFFFu) {
                    // ---- end of round (wfa_graph.rs:633-648) ----
                    const uint32_t far = (uint32_t)w2_gmax<G>((int32_t)lane_far);
                    lane_far = 0;
                    if (final_found) { status = W2_ST_OK; score = ed; state = S_JOB; continue; }
                    if (far > farthest) farthest = far;
                    if (farthest > prune32) min_prog = farthest - prune32;
                    if (ed + 1u > maxed32) { status = W2_ST_MAX_ED; score = maxed32; state = S_JOB; continue; }
                    if (!round_live) { status = W2_ST_INTERNAL; state = S_JOB; continue; }
                    ++ed; p = c; c ^= 1u; lcnt_prev = lcnt_cur; lcnt_cur = 0; fcnt = 0; top = 0; pp = 0; round_live = false;
                    ph = live[p * C::MAXL];   // lcnt_prev >= 1 here
                    continue;
                }
                {
                    // (a copy of the descriptors in LDS, direct-mapped, was measured: the visit no longer waits for L2, the launch set
                    // is no faster - three wavefronts per SIMD hide that wait already, what the kernel lacks is issue slots - and
                    // the tables it displaced doubled the hand-overs)
                    const uint4 nd = *reinterpret_cast<const uint4*>(gnode + n);   // seq_off, len | is_ref, children, first two children
                    len = nd.y & ~W2_IS_REF;
                    nseq = ((nd.y & W2_IS_REF) ? refp : altp) + nd.x;
                    n_child = nd.z & 0xFFFFu; child_off = nd.z >> 16; c01 = nd.w;   // (child_off: first overflow entry)
                }
                // ---- sources: previous entries of n grown by one diagonal a side, finished parents, the start wave ----
                lo = INT32_MAX; hi = INT32_MIN;
                np = 0;
                if (na == n) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        if ((ph.x & 0xFFFFu) == n) {   // (entries of one node are consecutive in the live list)
                            pa[k] = (int32_t)ph.y + (int32_t)(ph.z & 0xFFu); pb[k] = (int32_t)ph.y + (int32_t)((ph.z >> 8) & 0xFFu); po[k] = (int32_t)(ph.x >> 16) - (int32_t)ph.y;
                            lo = min(lo, pa[k] - 1); hi = max(hi, pb[k] + 1);
                            np = (uint32_t)k + 1u;
                            ++pp;
                            ph = pp < lcnt_prev ? live[p * C::MAXL + pp] : make_uint4(0xFFFFu, 0, 0, 0);
                        }
                    }
                    if ((ph.x & 0xFFFFu) == n) { status = W2_ST_NEED_BIG, why = 6u; state = S_JOB; continue; }   // more entries of one node than the class holds
                }
                npar = 0; pc0 = 0; pc1 = 0;
                if (nb == n) {
                    const bool mine = pq_node == n;
                    npar = w2_gor<G>(mine ? pq_cnt : 0u); pc0 = w2_gor<G>(mine ? pq_c0 : 0u); pc1 = w2_gor<G>(mine ? pq_c1 : 0u);
                    if (mine) pq_node = 0xFFFFu;
#pragma clang loop unroll(disable)
                    for (uint32_t i = 0; i < npar; ++i) {
                        const uint32_t code = ((i < 4u ? pc0 >> (8u * i) : pc1 >> (8u * (i - 4u))) & 0xFFu);
                        const uint4 h = code < 128u ? live[c * C::MAXL + code] : fin[code - 128u];
                        const int32_t f0 = (int32_t)h.y + (int32_t)h.w + (int32_t)((h.z >> 16) & 0xFFu), f1 = (int32_t)h.y + (int32_t)h.w + (int32_t)(h.z >> 24);
                        lo = min(lo, f0); hi = max(hi, f1);
                    }
                }
                if (ed == 0 && n == 0) { lo = min(lo, 0); hi = max(hi, 0); }
                n_items = 1; item = 0; use_list = false; n_live_node = 0;
                if (hi - lo >= 2 * (int32_t)G) {
                    // far-apart sources (a structural variant upstream puts two paths hundreds of diagonals apart): walk
                    // them again, merge what overlaps or touches, every remaining interval is an item of its own
                    uint32_t ni = 0;
                    const uint32_t ns = np + npar + ((ed == 0 && n == 0) ? 1u : 0u);
#pragma clang loop unroll(disable)
                    for (uint32_t i = 0; i < ns; ++i) {
                        int2 iv = make_int2(0, 0);
                        if (i < np) {
#pragma unroll
                            for (int k = 0; k < NP; ++k) if (i == (uint32_t)k) iv = make_int2(pa[k] - 1, pb[k] + 1);
                        } else if (i - np < npar) {
                            const uint32_t q = i - np;
                            const uint32_t code = ((q < 4u ? pc0 >> (8u * q) : pc1 >> (8u * (q - 4u))) & 0xFFu);
                            const uint4 h = code < 128u ? live[c * C::MAXL + code] : fin[code - 128u];
                            iv = make_int2((int32_t)h.y + (int32_t)h.w + (int32_t)((h.z >> 16) & 0xFFu), (int32_t)h.y + (int32_t)h.w + (int32_t)(h.z >> 24));
                        }
                        uint32_t j = 0;
#pragma clang loop unroll(disable)
                        while (j < ni) {
                            const int2 it = srcs[j];
                            if (it.x <= iv.y + 1 && iv.x <= it.y + 1) {
                                iv.x = min(iv.x, it.x); iv.y = max(iv.y, it.y);
                                --ni;
                                const int2 mv = srcs[ni];
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                if (gl == 0) srcs[j] = mv;
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                j = 0;
                            } else ++j;
                        }
                        if (ni >= (uint32_t)C::MAXS) { status = W2_ST_NEED_BIG, why = 7u; break; }
                        if (gl == 0) srcs[ni] = iv;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        ++ni;
                    }
                    if (status != W2_ST_PENDING) { state = S_JOB; continue; }
                    n_items = ni;
                    use_list = true;
                }
            }
            {
                if (use_list) { const int2 iv = srcs[item]; lo = iv.x; hi = iv.y; }
                ++item;
                const uint32_t cnt = (uint32_t)(hi - lo + 1);
                if (top + cnt > (uint32_t)C::SLOTS || lo <= -W2_DIAG_LIM || hi >= W2_DIAG_LIM) { status = W2_ST_NEED_BIG, why = 8u; state = S_JOB; continue; }
                coff = top; top += cnt; base = lo;
#if W2_STATS
                mx_top = max(mx_top, top);
#endif
                chi = INT32_MIN; cvlo = INT32_MAX; cvhi = INT32_MIN; cflo = INT32_MAX; cfhi = INT32_MIN;
                state = S_TILE;
            }
        }
        W2PT(1);
        if (!__any(state == S_TILE)) {
            if (!__any(state == S_WAIT)) break;   // every group is done
            if (++idle_polls > 60000u) break;     // (~1.5 s of nothing to do: never hang the device; unclaimed jobs stay for the host's pass)
            for (int z = 0; z < 8; ++z) __builtin_amdgcn_s_sleep(127);
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        W2PC(2, __popcll(__ballot(state == S_TILE)));

        // ============================ 2. one tile: G diagonals of the current item ===================================
        const bool run = state == S_TILE;
        const int32_t d = base + (int32_t)gl;
        const bool act = run && d <= hi;
        // ---- candidates from the previous round: A from d+1 (offset+1), B from d (offset+1), C from d-1 (offset) ----
        int32_t oA = -1, oB = -1, oC = -1;
        int32_t sA = -1, sB = -1, sC = -1;   // their slots
        if (act) {
#pragma unroll
            for (int k = 0; k < NP; ++k)
                if ((uint32_t)k < np) {
                    if (d + 1 >= pa[k] && d + 1 <= pb[k]) sA = po[k] + d + 1;
                    if (d >= pa[k] && d <= pb[k]) sB = po[k] + d;
                    if (d - 1 >= pa[k] && d - 1 <= pb[k]) sC = po[k] + d - 1;
                }
        }
        const uint32_t pbase = p * C::SLOTS, cbase = c * C::SLOTS;
        {
            const uint32_t eA = ek[pbase + (uint32_t)max(sA, 0)], eB = ek[pbase + (uint32_t)max(sB, 0)], eC = ek[pbase + (uint32_t)max(sC, 0)];
            if (sA >= 0 && (eA & 1u)) oA = (int32_t)(eA >> 3) + 1; else sA = -1;
            if (sB >= 0 && (eB & 7u) == W2_KIND_INTERIOR_READ) oB = (int32_t)(eB >> 3) + 1; else sB = -1;
            if (sC >= 0 && ((eC & 7u) == W2_KIND_INTERIOR_READ || (eC & 7u) == W2_KIND_END_LAST)) oC = (int32_t)(eC >> 3); else sC = -1;
        }
        // their traversed-node sets are only needed for the union at the end of the step: the loads go out now (issued after
        // the tie checks instead they save 7 spilled registers at 3 wavefronts per SIMD but expose their latency: 3 % slower)
        const W2Set<W> qA = w2_ldset<W>(gs + (size_t)(pbase + (uint32_t)max(sA, 0)) * W, sA >= 0);
        const W2Set<W> qB = w2_ldset<W>(gs + (size_t)(pbase + (uint32_t)max(sB, 0)) * W, sB >= 0);
        const W2Set<W> qC = w2_ldset<W>(gs + (size_t)(pbase + (uint32_t)max(sC, 0)) * W, sC >= 0);
        // ---- waves that finished a parent THIS round (offset 0; wfa_graph.rs:527-553) ----
        bool hinj = false;
        W2Set<W> qD = w2_set0<W>();   // (a second or later finished parent is ORed in at once: rare, and it saves W registers)
        if (run) {
#pragma clang loop unroll(disable)
            for (uint32_t i = 0; i < npar; ++i) {
                const uint32_t code = ((i < 4u ? pc0 >> (8u * i) : pc1 >> (8u * (i - 4u))) & 0xFFu);
                const uint4 h = code < 128u ? live[c * C::MAXL + code] : fin[code - 128u];
                const int32_t rel = d - (int32_t)h.w - (int32_t)h.y;   // relative to the parent entry's first diagonal
                bool hit = false;
                uint32_t s = 0;
                if (act && rel >= (int32_t)((h.z >> 16) & 0xFFu) && rel <= (int32_t)(h.z >> 24)) {
                    s = cbase + (h.x >> 16) + (uint32_t)rel;
                    hit = (ek[s] & 7u) == W2_KIND_FINISHED;
                }
                hinj = hinj || hit;
                if (i == 0) qD = w2_ldset<W>(gs + (size_t)s * W, hit);
                else {
                    const W2Set<W> t = w2_ldset<W>(gs + (size_t)s * W, hit);
#pragma unroll
                    for (int w = 0; w < W; ++w) qD.w[w] |= t.w[w];
                }
            }
            if (act && ed == 0 && n == 0 && d == 0) hinj = true;   // the start wave (wfa_graph.rs:366-378)
        }
        W2PT(2);
        const bool has = act && (oA >= 0 || oB >= 0 || oC >= 0 || hinj);
        lane_upd += has ? 1u : 0u;
        W2PC(3, __popcll(__ballot(has)));
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
        // A candidate behind the furthest one ties iff it matches the read up to omax. When the furthest is this diagonal's own wave
        // (B: offset + 1 of a wave that stopped INSIDE node and read, i.e. on a mismatch at offset omax - 1 of this diagonal), every
        // other candidate would have to match through that very position: none ties, nothing to check.
        const bool chk = has && !tB;
        const bool nA = chk && oA >= 0 && oA < omax, nB = chk && oB >= 0 && oB < omax, nC = chk && oC >= 0 && oC < omax;
        const bool nD = chk && hinj && omax > 0;
        // every global load of the step goes out together: the extension's first 16 bytes, the first 16 bytes of every
        // tie check, and the probe of the capped-diagonal set
        const uint8_t* ra = readp + (has ? pos0 : 0);
        const uint8_t* na_ = nseq + (has ? omax : 0);
        // the node's capped-diagonal record: one 16-byte load per group (neighbouring nodes share a line); the hash set is only
        // probed for a diagonal outside the record's window
        uint4 crec = make_uint4(0, 0, 0, 0);
        if (run) crec = *reinterpret_cast<const uint4*>(gs + C::SET_DWORDS + 4u * n);   // (every lane of the group: the first lane writes it back)
        const W2Pre pm = w2_pre(na_, ra, room > 0);
        const W2Pre8 pA = w2_pre8(nseq + (nA ? oA : 0), readp + (nA ? d + oA : 0), nA);
        const W2Pre8 pB = w2_pre8(nseq + (nB ? oB : 0), readp + (nB ? d + oB : 0), nB);
        const W2Pre8 pC = w2_pre8(nseq + (nC ? oC : 0), readp + (nC ? d + oC : 0), nC);
        const W2Pre8 pD = w2_pre8(nseq, readp + (nD ? d : 0), nD);
        W2PT(3);
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
            W2PC(4, __any(!done && run) ? 1 : 0);
            E = (uint32_t)omax + w2_match_rest<G>(nseq, readp, (uint32_t)omax, pos0, room, n0, done, run, gbase, gl);
        }
        W2PT(4);
        {
            // a tie check whose gap exceeds 8 bytes and whose first 8 match goes on with the cooperative compare
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
                bool addA = false, addB = false, addC = false, addD = false;
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    uint32_t dset = qD.w[w];
                    if ((n >> 5) == (uint32_t)w) dset |= 1u << (n & 31u);
                    const uint32_t known = (tA ? qA.w[w] : 0u) | (tB ? qB.w[w] : 0u) | (tC ? qC.w[w] : 0u) | (tD ? dset : 0u);
                    addA = addA || (qA.w[w] & ~known); addB = addB || (qB.w[w] & ~known); addC = addC || (qC.w[w] & ~known); addD = addD || (dset & ~known);
                }
                pdA = pdA && addA; pdB = pdB && addB; pdC = pdC && addC; pdD = pdD && addD;
            }
            W2PC(5, __any(pdA || pdB || pdC || pdD) ? 1 : 0);
            if (__any(pdA || pdB || pdC || pdD)) {   // rare: a long alternative run onto the furthest wave's diagonal
                auto slow = [&](bool pd, int32_t oX) -> bool {
                    const uint32_t g = pd ? (uint32_t)(omax - oX) : 0u;
                    const uint32_t mr = w2_match_rest<G>(nseq, readp, (uint32_t)(pd ? oX : 0), pd ? d + oX : 0, g, pd ? 8u : 0u, !pd, run, gbase, gl);
                    return pd && mr == g;
                };
                const bool yA = slow(pdA, oA), yB = slow(pdB, oB), yC = slow(pdC, oC), yD = slow(pdD, 0);
                tA = tA || yA; tB = tB || yB; tC = tC || yC; tD = tD || yD;
            }
        }
        W2PT(5);
        // ---- capped-diagonal set: is (n, d) recorded? ---------------------------------------------------------------------
        const bool rec_live = crec.x == tag;                       // (else: nothing of this job recorded for the node yet)
        const uint32_t rel = (uint32_t)(d - (int32_t)crec.y + 32);   // bit of diagonal d in the record's window
        const bool in_win = rec_live && rel < 64u;
        bool capped = in_win && (((rel < 32u ? crec.z : crec.w) >> (rel & 31u)) & 1u);
        bool hfull = false;
        const bool use_hash = has && rec_live && !in_win;          // out of the window: the hash set (rare)
        W2PC(6, __any(use_hash) ? 1 : 0);
        if (__any(use_hash)) {
            if (use_hash) {
                const uint64_t key = ((uint64_t)tag << 32) | ((uint64_t)(n & 0x3FFu) << 18) | (uint64_t)((uint32_t)d & 0x3FFFFu);
                uint32_t hp = (n * 0x9E3779B1u + (uint32_t)d) & hmask, probes = 0;
                uint64_t e = htab[hp];
                // (rolled: unrolled 24 deep, each level of the probe holds a saved exec mask - fifty scalar registers the
                // rest of the step then spills)
#pragma clang loop unroll(disable)
                while (e != key && (uint32_t)(e >> 32) == tag) {
                    if (++probes > 24u) { hfull = true; break; }
                    hp = (hp + 1u) & hmask;
                    e = htab[hp];
                }
                capped = e == key;
            }
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
                ins = (int32_t)E == cap && !capped && !hfull;
                if (E == len) {
                    if (n == last) { if ((uint32_t)pos_end < other_len) kind = W2_KIND_END_LAST; }
                    else kind = W2_KIND_FINISHED;
                } else kind = ((uint32_t)pos_end < other_len) ? W2_KIND_INTERIOR_READ : W2_KIND_INTERIOR;
            }
        }
        // ---- record newly capped diagonals: in the node's record (the group's first lane stores it; an empty record takes
        // the first such diagonal as its anchor), or - outside its window - in the hash set, one lane at a time so that
        // two of them never take the same empty slot --------------------------------------------------------------------
        bool later = false;
        W2PC(7, __any(ins) ? 1 : 0);
        if (__any(ins)) {
            const uint64_t im = w2_gballot<G>(ins, gbase);
            if (im) {   // (group-uniform)
                int32_t anchor = (int32_t)crec.y;
                if (!rec_live) anchor = (int32_t)w2_gsel<G>((uint32_t)d, gl, (uint32_t)__builtin_ctzll(im));
                const uint32_t r2 = (uint32_t)(d - anchor + 32);
                const bool mine = ins && r2 < 64u;
                later = ins && !mine;
                const uint32_t lo = w2_gor<G>(mine && r2 < 32u ? 1u << r2 : 0u), hi = w2_gor<G>(mine && r2 >= 32u ? 1u << (r2 - 32u) : 0u);
                if (gl == 0) *reinterpret_cast<uint4*>(gs + C::SET_DWORDS + 4u * n) = make_uint4(tag, (uint32_t)anchor, (rec_live ? crec.z : 0u) | lo, (rec_live ? crec.w : 0u) | hi);
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
        }
        if (__any(hfull)) { if (w2_gballot<G>(hfull, gbase)) status = W2_ST_NEED_BIG, why = 9u; }
        W2PT(6);
        // ---- write this round's slot: offset | kind and the union of the tied sets -----------------------------------
        W2Set<W> best;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint32_t dset = qD.w[w];
            if (hinj && (n >> 5) == (uint32_t)w) dset |= 1u << (n & 31u);   // best + the successor (wfa_graph.rs:535-541)
            best.w[w] = (tA ? qA.w[w] : 0u) | (tB ? qB.w[w] : 0u) | (tC ? qC.w[w] : 0u) | (tD ? dset : 0u);
        }
        if (act) {
            const uint32_t s = cbase + coff + (uint32_t)(d - lo);
            ek[s] = has ? ((E << 3) | kind) : 0u;
            if (kind != W2_KIND_NONE) w2_stset<W>(gs + (size_t)s * W, best);
        }
        // ---- finals (wfa_graph.rs:576-629): every wave of the last node that consumed node and read -------------------
        if (__any(is_final)) {
            const bool gf = w2_gballot<G>(is_final, gbase) != 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t o = w2_gor<G>(is_final ? best.w[w] : 0u);
                if (gf && gl == 0) outset[w] |= o;
            }
            if (gf) final_found = true;
        }
        W2PT(7);
        // ---- clusters of non-empty diagonals become this round's entries ----------------------------------------------
        if (run) {
            const uint64_t fm = w2_gballot<G>(kind == W2_KIND_FINISHED, gbase);
            uint64_t rem = fm | w2_gballot<G>(kind == W2_KIND_INTERIOR || kind == W2_KIND_INTERIOR_READ || kind == W2_KIND_END_LAST, gbase);
            // one pass per run of diagonals (gaps of at most two inside a run); the item's last tile flushes the open cluster
            bool end_pending = base + (int32_t)G > hi;
#pragma clang loop unroll(disable)
            while (rem || end_pending) {
                int32_t first = INT32_MAX / 2, lastd = INT32_MAX / 2;
                uint64_t rm = 0;
                const bool bit = rem != 0;
                if (bit) {
                    const int f = __builtin_ctzll(rem);
                    const uint64_t z = ~(rem >> f);
                    const int e = __builtin_ctzll(z & (z >> 1) & (z >> 2));   // the run is bits [f, f + e)
                    rm = rem & ((1ull << (f + e)) - 1ull);
                    rem &= ~rm;
                    first = base + f; lastd = base + f + e - 1;
                } else end_pending = false;
                // (a node keeps at most two live entries per round - the next round reads at most two: once it has one, the
                // rest of its diagonals stay one cluster, gaps included)
                if (chi != INT32_MIN && ((first - chi >= 3 && n_live_node == 0u) || lastd - clo >= (int32_t)C::MAXW)) emit_cluster();   // (n_live_node: see MAXPREV)
                if (bit) {
                    if (chi == INT32_MIN) clo = first;
                    chi = lastd;
                    const uint64_t fr = fm & rm, lr = rm & ~fm;
                    if (fr) { cflo = min(cflo, base + (int32_t)__builtin_ctzll(fr)); cfhi = max(cfhi, base + 63 - (int32_t)__builtin_clzll(fr)); }
                    if (lr) { cvlo = min(cvlo, base + (int32_t)__builtin_ctzll(lr)); cvhi = max(cvhi, base + 63 - (int32_t)__builtin_clzll(lr)); }
                }
            }
            base += (int32_t)G;
            if (base > hi) state = S_NEXT;
            if (++steps > W2_MAX_STEPS) status = W2_ST_INTERNAL;
            if (status != W2_ST_PENDING) state = S_JOB;
        }
        W2PT(8);
    }
    if (B.esc_role == 1u) {   // a producer workgroup is gone (everything it hands over has been published)
        W2_WAIT_VM();
        (void)atomicAdd(B.esc + 2, lane == 0 ? 1u : 0u);
    }
#if W2_PROF
    if (lane == 0 && (blockIdx.x % 97) == 3)
        printf("wg %u total %llu | idle->ctl %llu control %llu cand %llu issue %llu ext %llu ties %llu hash %llu write+final %llu cluster %llu | iters %u ctlpasses %u tile-lanes %u has-lanes %u ext2 %u tieslow %u hashprobe %u capins %u\n",
               blockIdx.x, (unsigned long long)(w2tl - w2t0), (unsigned long long)w2pc[0], (unsigned long long)w2pc[1], (unsigned long long)w2pc[2], (unsigned long long)w2pc[3],
               (unsigned long long)w2pc[4], (unsigned long long)w2pc[5], (unsigned long long)w2pc[6], (unsigned long long)w2pc[7], (unsigned long long)w2pc[8], w2pn[0], w2pn[1], w2pn[2], w2pn[3], w2pn[4], w2pn[5], w2pn[6], w2pn[7]);
#endif
}

// ---- graph-size classes: one thread per job, in longest-read-first order --------------------------------------------
// A job goes to the smallest class whose tables hold its graph (W2Cfg<2/4/8>), or stays NEED_BIG for the dense-band
// path. The class lists keep the longest-read-first order exactly (it steers the work queues: longest jobs first keeps
// the tail of the persistent kernels short - a list in atomic-arrival order cost 6 ms of 22): kernel 1 classifies and
// counts per workgroup of 256 jobs, kernel 2 turns the counts into offsets and scatters.
struct W2ClassArgs {
    const W2Job* jobs;
    const W2Info* info;
    const uint32_t* len_order;
    uint32_t n_jobs;
    uint8_t* cls;         // [n_jobs] class of the job at each position of len_order (3 = no class)
    uint32_t* blockcnt;   // [workgroups][4]
    uint32_t* order;      // [3][n_jobs]
    uint32_t* counts;     // [4]: jobs per class, [3] = jobs no class takes
    int32_t* status;
    uint32_t* esc;        // W2Batch::esc: [0] = [1] = jobs of the largest class, [2] = [3] = 0
    uint32_t use_w2;      // 0: graphs of at most 64 nodes join the middle class (one queue, one tail)
    uint8_t* job_cls;     // [n_jobs] by job id: 0..2, 3 = no class
    const uint8_t* fmt;   // block mode (else null): bit 7 = the host routed this record past the compact kernels (W2_FMT_SUSPECT):
                          // its way out - reference-window test, dense band - started when the set's launch set did
};
__global__ void __launch_bounds__(256) hp_wfa2_classify_kernel(W2ClassArgs A) {
    __shared__ uint32_t wcnt[4][4];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool on = t < A.n_jobs;
    int k = 3;
    if (on) {
        const uint32_t i = A.len_order[t];
        const W2Info in = A.info[i];
        const bool routed = A.fmt != nullptr && (A.fmt[i] & W2_FMT_SUSPECT) != 0u;
        if (in.status == W2B_OK && A.jobs[i].read_len < (uint32_t)W2_DIAG_LIM && !routed) {
            if (A.use_w2 && in.n_nodes <= (uint32_t)W2Cfg<2>::MAXN && in.n_edges <= (uint32_t)W2Cfg<2>::MAXE) k = 0;
            else if (in.n_nodes <= (uint32_t)W2Cfg<4>::MAXN && in.n_edges <= (uint32_t)W2Cfg<4>::MAXE) k = 1;
            else if (in.n_nodes <= (uint32_t)W2Cfg<8>::MAXN && in.n_edges <= (uint32_t)W2Cfg<8>::MAXE) k = 2;
        }
        A.status[i] = k < 3 ? W2_ST_PENDING : W2_ST_NEED_BIG;
        A.cls[t] = (uint8_t)k;
        A.job_cls[i] = (uint8_t)k;
    }
    for (int c = 0; c < 4; ++c) {
        const uint64_t m = __ballot(on && k == c);
        if (lane == 0) wcnt[wave][c] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 4) A.blockcnt[(size_t)blockIdx.x * 4 + threadIdx.x] = wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
}
__global__ void __launch_bounds__(256) hp_wfa2_scatter_kernel(W2ClassArgs A) {
    __shared__ uint32_t part[4][4];    // per wave: partial sums of the counts of the workgroups before this one
    __shared__ uint32_t wcnt[4][4];    // per wave: members of each class in this workgroup
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t s[4] = {0, 0, 0, 0};
    const uint32_t upto = blockIdx.x + ((blockIdx.x + 1 == gridDim.x) ? 1u : 0u);   // the last workgroup also totals the counts
    for (uint32_t b = threadIdx.x; b < upto; b += 256u) {
        const uint4 c = *reinterpret_cast<const uint4*>(A.blockcnt + (size_t)b * 4);
        s[0] += c.x; s[1] += c.y; s[2] += c.z; s[3] += c.w;
    }
    for (int c = 0; c < 4; ++c) {
        uint32_t v = s[c];
        for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
        if (lane == 0) part[wave][c] = v;
    }
    const bool on = t < A.n_jobs;
    const int k = on ? (int)A.cls[t] : 3;
    uint32_t pre = 0;
    for (int c = 0; c < 4; ++c) {
        const uint64_t m = __ballot(on && k == c);
        if (lane == 0) wcnt[wave][c] = (uint32_t)__popcll(m);
        if (k == c) pre = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (blockIdx.x + 1 == gridDim.x && threadIdx.x < 4) {
        const uint32_t tot = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        A.counts[threadIdx.x] = tot;
        if (threadIdx.x == 2) { A.esc[0] = tot; A.esc[1] = tot; A.esc[2] = 0; A.esc[3] = 0; }
    }
    if (on && k < 3) {
        uint32_t base = part[0][k] + part[1][k] + part[2][k] + part[3][k];
        if (blockIdx.x + 1 == gridDim.x) base -= A.blockcnt[(size_t)blockIdx.x * 4 + k];   // (its own count went into the total)
        for (uint32_t w = 0; w < wave; ++w) base += wcnt[w][k];
        A.order[(size_t)k * A.n_jobs + base + pre] = A.len_order[t];
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
    W2Info* info;
};
__global__ void __launch_bounds__(64) hp_wfa2_build_kernel(W2BuildArgs A) {
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    if (j >= A.n_jobs) return;
    const W2Job J = A.jobs[j];
    w2_build(J, A.vars, A.nodes + J.node_off, A.edges + J.edge_off, A.tags + J.tag_off, A.info + j);
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
    const W2Node* nodes;
    uint32_t* out_work;
    // first == 1: the largest class's kernel may still be running (it aligns its own jobs and the ones handed over). Only
    // what a kernel that has COMPLETED wrote is read: jobs of the two smaller classes that were not handed over, and jobs
    // no class took. The others are reported PENDING in `seen` and mapped by the second launch.
    const uint8_t* job_cls;
    const uint8_t* handed;
    int32_t* seen;        // the status this launch acted on
    uint32_t first;
};
__global__ void __launch_bounds__(64) hp_wfa2_map_kernel(W2MapArgs A) {
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    if (j >= A.n_jobs) return;
    if (A.first && (A.job_cls[j] == 2u || A.handed[j])) { A.seen[j] = W2_ST_PENDING; return; }
    const int32_t st = A.status[j];
    A.seen[j] = st;
    const W2Job J = A.jobs[j];
    const uint32_t* set = A.out_sets + (size_t)j * W2_SET_STRIDE;
    const bool ok = st == W2_ST_OK;
    w2_map_alleles(A.tags + J.tag_off, A.info[j].n_tags, set, ok, A.alleles + J.allele_off, J.n_hets);
    uint32_t bytes = 0;   // work counter: bytes of the nodes the best alignment(s) traverse
    if (ok && A.info[j].n_nodes <= 32u * W2_SET_STRIDE)
        for (uint32_t n = 0; n < A.info[j].n_nodes; ++n)
            if ((set[n >> 5] >> (n & 31u)) & 1u) bytes += A.nodes[J.node_off + n].len_ref & ~W2_IS_REF;
    A.out_work[(size_t)j * 2 + 1] = bytes;
}

// The second collection (two phases): the same mapping for the few jobs the largest class's kernel delivered, gathered - one
// 32-byte record and the allele row per job, back to back - so that what crosses PCIe is kilobytes, not the whole batch again.
struct W2HeldRec { int32_t status; uint32_t work_updates; uint64_t score; uint32_t work_bytes; uint32_t pad[3]; };
struct W2MapHeldArgs {
    W2MapArgs M;
    const uint32_t* held;    // job ids
    const uint32_t* hoff;    // [n_held + 1] offsets of their allele rows in `rows`
    uint32_t n_held;
    W2HeldRec* rec;
    uint8_t* rows;
    const uint64_t* out_score;
};
__global__ void __launch_bounds__(64) hp_wfa2_map_held_kernel(W2MapHeldArgs A) {
    const uint32_t k = blockIdx.x * 64u + threadIdx.x;
    if (k >= A.n_held) return;
    const uint32_t j = A.held[k];
    const int32_t st = A.M.status[j];
    const W2Job J = A.M.jobs[j];
    const uint32_t* set = A.M.out_sets + (size_t)j * W2_SET_STRIDE;
    const bool ok = st == W2_ST_OK;
    w2_map_alleles(A.M.tags + J.tag_off, A.M.info[j].n_tags, set, ok, A.rows + A.hoff[k], J.n_hets);
    uint32_t bytes = 0;
    if (ok && A.M.info[j].n_nodes <= 32u * W2_SET_STRIDE)
        for (uint32_t n = 0; n < A.M.info[j].n_nodes; ++n)
            if ((set[n >> 5] >> (n & 31u)) & 1u) bytes += A.M.nodes[J.node_off + n].len_ref & ~W2_IS_REF;
    W2HeldRec r{};
    r.status = st; r.work_updates = A.M.out_work[(size_t)j * 2]; r.score = A.out_score[j]; r.work_bytes = bytes;
    A.rec[k] = r;
}

}  // namespace hp

namespace hp {

// ---- read bases into the layout the alignment kernels read: one wavefront per read -----------------------------------
// The host stages every read as the caller holds it (HP_SEQ_ASCII bytes, or the BAM record's 4-bit codes, HP_SEQ_BAM4:
// half the bytes across PCIe and no decode on the host); this kernel writes one byte per base at the job's read_off,
// decoding 4-bit codes with htslib's table "=ACMGRSVTWYHKDBN" (what read.seq().as_bytes() yields, read_parsing.rs:738).
// A lane expands 8 source bytes to 16 bases per iteration; a read's first base may sit in a low nibble (odd read_offset).
struct W2UnpackArgs {
    const W2Job* jobs;
    const uint64_t* src_off;    // [n_jobs] byte offset in `packed` of the byte that holds the read's first base
    const uint8_t* fmt_nib;     // [n_jobs] seq_format | first base in the low nibble << 4
    uint32_t n_jobs;
    const uint8_t* packed;
    uint8_t* seq;
    uint64_t tail_off;          // 256 bytes of zeros behind the last read (the 32-byte compares may run past its last base)
};
__global__ void __launch_bounds__(256) hp_wfa2_unpack_kernel(W2UnpackArgs A) {
    __shared__ uint16_t lut[256];   // source byte -> two bases (first base = high nibble = low byte of the pair)
    {
        const char* tab = "=ACMGRSVTWYHKDBN";
        const uint32_t t = threadIdx.x;
        lut[t] = (uint16_t)((uint32_t)(uint8_t)tab[t >> 4] | ((uint32_t)(uint8_t)tab[t & 15u] << 8));
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 64) reinterpret_cast<uint32_t*>(A.seq + A.tail_off)[threadIdx.x] = 0u;
    const uint32_t j = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (j >= A.n_jobs) return;
    const W2Job J = A.jobs[j];
    const uint8_t* src = A.packed + A.src_off[j];
    uint8_t* dst = A.seq + J.read_off;
    const uint32_t fm = A.fmt_nib[j], len = J.read_len;
    if ((fm & 15u) == HP_SEQ_ASCII) {
        for (uint32_t o = lane * 16u; o < len; o += 64u * 16u) {   // (source and destination slots are both padded to 16 bytes)
            uint4 v; __builtin_memcpy(&v, src + o, 16);
            *reinterpret_cast<uint4*>(dst + o) = v;
        }
        return;
    }
    const uint32_t odd = (fm >> 4) & 1u;   // (bit 7: routed past the compact kernels, W2_FMT_SUSPECT - not this kernel's business)
    for (uint32_t o = lane * 16u; o < len; o += 64u * 16u) {
        uint64_t lo, hi;
        __builtin_memcpy(&lo, src + (o >> 1), 8);
        __builtin_memcpy(&hi, src + (o >> 1) + 8, 8);
        // odd: the first base is the low nibble of byte 0: drop a nibble - within each byte the high nibble comes first, so base
        // k pairs with base k + 1 of the NEXT byte's high nibble: byte' = (byte << 4) | (next byte >> 4)
        if (odd) {
            const uint64_t nx = (lo >> 8) | (hi << 56);
            lo = ((lo << 4) & 0xF0F0F0F0F0F0F0F0ull) | ((nx >> 4) & 0x0F0F0F0F0F0F0F0Full);
        }
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t b0 = (uint32_t)(lo >> (16 * k)) & 0xFFu, b1 = (uint32_t)(lo >> (16 * k + 8)) & 0xFFu;
            w[k] = (uint32_t)lut[b0] | ((uint32_t)lut[b1] << 16);
        }
        *reinterpret_cast<uint4*>(dst + o) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

}  // namespace hp

namespace hp {

// ---- a cheap exact verdict for hopeless reads ------------------------------------------------------------------------------------
// A read that leaves the compact kernels deep into its alignment is usually one that will exhaust max_edit_distance (a noisy
// read, a mis-placed supplementary record): the dense-band pass then walks the whole graph for max_edit_distance rounds only to
// report Err(MaxEditDistance) (wfa_graph.rs:645-648). This kernel settles most of them first, exactly: plain two-sequence WFA
// (unit costs, end to end) of the read against the job's REFERENCE window alone, one wavefront per job, up to a threshold
// T = max_edit_distance + D, D = the sum over the job's variants of max(reference length, allele lengths). Every path P of the
// job's graph is the reference window with some variants' reference spans replaced by an allele, so ed(window, P) <= D and, by
// the triangle inequality, ed(read, P) >= ed(read, window) - D: if ed(read, window) > T no path is within max_edit_distance of
// the read, and the reference's search - pruned or not, it only ever reports the cost of an alignment it found - ends in
// MaxEditDistance. Nothing is decided when the distance is <= T: those jobs take the dense-band pass as before.
struct W2BoundArgs {
    const W2Job* jobs;
    const uint32_t* ids;       // jobs to test
    const uint32_t* thresh;    // T per tested job (<= W2_BOUND_MAX_T)
    uint32_t n;
    const uint8_t* seq;
    uint8_t* exceeds;          // out: 1 = ed(read, reference window) > T
    uint32_t max_t;            // the largest threshold of the launch (sizes the wavefront arrays)
    uint32_t lds_seq;          // bytes of LDS behind them for a job's read + reference window (0: compare in place)
};
constexpr uint32_t W2_BOUND_MAX_T = 1000;   // the test costs T^2 / 64 tiles: 3 ms of a wavefront at 1 000, 30 at 3 000 (where a structural
                                            // variant in the window has made D so large that the test rarely settles anything)
constexpr uint32_t W2_BOUND_LDS_SEQ = 96 * 1024;   // read + reference window are staged in LDS when they fit in this many bytes
// NT threads per job (64 or 256): a round's diagonals (up to 2 T + 1 of them) are independent, the jobs are a few hundred, and what
// the rows stage waits for is their latency. Four wavefronts need a free slot on every SIMD of one CU at once, though - beside a
// resident launch set that may be a long wait (hp_wfa2.hip picks; HP_BOUND_THREADS).
template <uint32_t W2_BOUND_THREADS>
__global__ void __launch_bounds__(W2_BOUND_THREADS) hp_wfa2_bound_kernel(W2BoundArgs A) {
    extern __shared__ int32_t w2b_lds[];
    __shared__ int w2b_done;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (q >= A.n) return;
    const W2Job J = A.jobs[A.ids[q]];
    const int32_t T = (int32_t)A.thresh[q];
    const int32_t n = (int32_t)J.read_len, m = (int32_t)J.ref_len;
    const uint8_t* a = A.seq + J.read_off;
    const uint8_t* b = A.seq + J.ref_off;
    // The extensions are a chain of dependent loads (a noisy read mismatches within a few bases, 5 600 tiles of 64 diagonals for
    // T = 600): from HBM that is 8 ms a read, from LDS under 1. Both sequences are staged behind the wavefront arrays when they
    // fit (A.lds_seq = the launch's room for them; reads beyond ~45 kb compare in place).
    {
        const uint32_t na = ((uint32_t)n + 16u + 15u) & ~15u, nb = ((uint32_t)m + 16u + 15u) & ~15u;
        if (na + nb <= A.lds_seq) {
            uint8_t* la = reinterpret_cast<uint8_t*>(w2b_lds + 2 * (2 * A.max_t + 3) + 2);
            la = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(la) + 15) & ~(uintptr_t)15);
            uint8_t* lb = la + na;
            // (16-byte copies; the sources are 16-byte aligned only for the read - the window starts anywhere in its hull)
            for (uint32_t o = tid * 16u; o < na; o += W2_BOUND_THREADS * 16u) { uint4 v; __builtin_memcpy(&v, a + o, 16); *reinterpret_cast<uint4*>(la + o) = v; }
            for (uint32_t o = tid * 16u; o < nb; o += W2_BOUND_THREADS * 16u) { uint4 v; __builtin_memcpy(&v, b + o, 16); *reinterpret_cast<uint4*>(lb + o) = v; }
            a = la; b = lb;
        }
    }
    constexpr int32_t NONE = INT32_MIN / 2;
    const int32_t W = 2 * T + 3;                 // diagonals -T-1 .. T+1 (the rim stays NONE)
    int32_t* prev = w2b_lds;
    int32_t* cur = w2b_lds + W;
    for (int32_t i = (int32_t)tid; i < 2 * W; i += (int32_t)W2_BOUND_THREADS) w2b_lds[i] = NONE;
    if (tid == 0) w2b_done = 0;
    __syncthreads();
    auto extend = [&](int32_t i, int32_t k) {   // furthest i' >= i with a[i..i') == b[i + k .. i' + k)
        int32_t j = i + k;
        while (i < n && j < m) {
            if (i + 8 <= n && j + 8 <= m) {
                uint64_t x, y;
                __builtin_memcpy(&x, a + i, 8); __builtin_memcpy(&y, b + j, 8);
                const uint64_t d = x ^ y;
                if (d == 0) { i += 8; j += 8; continue; }
                const int32_t e = (int32_t)(__builtin_ctzll(d) >> 3);
                return i + e;
            }
            if (a[i] != b[j]) break;
            ++i; ++j;
        }
        return i;
    };
    const int32_t kend = m - n;                  // the diagonal of the end cell
    if (tid == 0) { const int32_t f = extend(0, 0); prev[T + 1] = f; if (kend == 0 && f == n) w2b_done = 1; }
    __syncthreads();
    bool done = w2b_done != 0;
    int32_t s = 0;
    while (!done && s < T) {
        ++s;
        bool mine = false;
        // Only diagonals that can still reach the end cell's within the T - s edits that are left take part: |kend - k| <= T - s (every
        // edit moves a path by at most one diagonal). Exact - a cell outside feeds only cells outside (its successors lie on k - 1, k,
        // k + 1 one round later, where the allowance is one smaller), so every cell inside sees the predecessors it would have seen -
        // and half the work: the triangle |k| <= s becomes the diamond between the start and the end cell. The cells a round reads
        // just outside its range are last round's (computed: the range shrinks by one a side) or the rim (never written: NONE).
        const int32_t left = T - s;
        const int32_t klo = kend - left > -s ? kend - left : -s, khi = kend + left < s ? kend + left : s;
        for (int32_t base = klo; base <= khi; base += (int32_t)W2_BOUND_THREADS) {
            const int32_t k = base + (int32_t)tid;
            if (k <= khi) {
                const int32_t p0 = prev[k + T + 1], p1 = prev[k + 1 + T + 1], pm = prev[k - 1 + T + 1];
                int32_t f = NONE;
                if (p0 >= 0 && p0 + 1 <= n && p0 + 1 + k <= m) f = p0 + 1;          // substitution
                if (p1 >= 0 && p1 + 1 <= n && p1 + 1 > f) f = p1 + 1;               // a read base of its own
                if (pm >= 0 && pm + k <= m && pm > f) f = pm;                       // a reference base of its own
                if (f >= 0) f = extend(f, k);
                cur[k + T + 1] = f;
                if (k == kend && f == n) mine = true;
            }
        }
        if (mine) w2b_done = 1;
        __syncthreads();                         // the round's diagonals are written; everyone sees whether the end cell was reached
        done = w2b_done != 0;
        int32_t* t = prev; prev = cur; cur = t;
    }
    if (tid == 0) A.exceeds[q] = done ? 0 : 1;
}

}  // namespace hp
