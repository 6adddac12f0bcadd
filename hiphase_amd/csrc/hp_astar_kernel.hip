// hp_astar_kernel.hip — the A* MEC phasing solver for gfx950 (MI355X), one wavefront per phase block.
//
// Replaces, bit-identically, reference src/astar_phaser.rs:
//   new_extended_node (:69-119)  -> expand(): lanes = rows covering the new variant; each lane scores
//                                   its row against both haplotype windows with bit-sliced AND/popcount
//                                   (v_bcnt_u32_b32) over 32-variant plane words; the 4 children share the
//                                   O(overlap) part; 8 partial sums are reduced with a DPP transpose-butterfly.
//   PriorityQueue (:316,460)     -> 64-way sharded binary heap (one private heap per lane, wave arg-min over
//                                   the 64 tops by DPP); the priority is a total order so any heap pops alike.
//                                   Sub-solver heap: packed 64-bit keys in LDS; main heap: 128-bit keys in HBM.
//   astar_subsolver (:311-405), calculate_astar_heuristic (:246-292), astar_solver (:426-633)
//                                -> subsolve()/solve_block(), same statement order, same `<`/`<=`.
// Integer-only (u32 per-row scores, u64 costs); no MFMA (there is no dense contraction on this path).
// Node state is O(1): a chain of 32-variant haplotype windows (see hp_astar_dev.h), not O(len) copies.
//
// Rules learnt on hardware (see DESIGN.md "Bring-up notes"):
//  * no atomics under a lane-0 branch: hipcc's atomic optimiser + the structuriser produced a wave that never
//    left the work-queue loop on gfx950 -> blocks are assigned statically (snake order over the LPT list);
//  * every global/LDS location is written and later read by the SAME lane (lane 0 for node records / H /
//    tracker, the owning lane for its private heap) and then broadcast with v_readfirstlane / v_readlane,
//    so no cross-lane visibility fences are needed inside the wave;
//  * no dynamically indexed private arrays (they become scratch memory = global-latency accesses).
#include "hp_astar_dev.h"
#include "hp_common.h"

namespace hp {

#define DEVINL __device__ __forceinline__
#ifndef HP_MAIN_PROF
#define HP_MAIN_PROF 0   // tuning aid: ticks of the main search by phase, packed over the counters' reserved words
#endif

// LDS layout (bytes): [0,512) H ring (64 x u64) | [512,768) variant ring (64 x (lo | flags << 28)) | [768,...) sub heap
// 768 + 11 x 512 B of heap = 6400 B at default parameters = exactly five 1280-byte LDS allocation granules of gfx950
// (measured, scripts/lds_probe.hip) -> 25 workgroups per CU by LDS, 24 by registers
constexpr uint32_t LDS_HRING_OFF = 0;
constexpr uint32_t LDS_VRING_OFF = 512;
constexpr uint32_t LDS_HEAP_OFF = 768;
extern __shared__ __attribute__((aligned(16))) unsigned char hp_smem[];

DEVINL uint32_t lane_id() { return __lane_id(); }
DEVINL uint32_t bcast32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DEVINL uint64_t bcast64(uint64_t v) {
    const uint32_t lo = bcast32((uint32_t)v), hi = bcast32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
DEVINL uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// ---- DPP helpers -----------------------------------------------------------------------------------------
// quad_perm [1,0,3,2] = 0xB1 (lane^1), [2,3,0,1] = 0x4E (lane^2), row_ror:8 = 0x128 (lane^8 inside a 16-lane
// row), row_shr:4 = 0x114, row_shl:4 = 0x104, row_half_mirror = 0x141, row_mirror = 0x140.
// old = 0 with bound_ctrl lets the backend fold the lane permutation into the consuming add (v_add_u32_dpp): every
// lane of these patterns has a valid source, so the zero is never observed.
template <int CTRL> DEVINL uint32_t dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// v + v[lane ^ 4]: lanes with bit2 set take lane-4 (banks 1,3), the others lane+4 (banks 0,2); a disabled bank reads 0
DEVINL uint32_t add_xor4(uint32_t v) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xA, true);
    const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, true);
    return v + up + dn;
}

// Sum of 8 per-lane values over the 64 lanes. Transpose-butterfly: three halving exchanges (lane^1, lane^2, lane^8)
// leave one value per lane (index = b0*4 + b1*2 + b3 of the lane number), one more exchange (lane^4) completes the
// 16-lane row, and the four rows are folded with the gfx950 row/half swaps (v_permlane16_swap: odd rows of the first
// operand <-> even rows of the second; v_permlane32_swap: upper half <-> lower half). Result: EVERY lane l holds the
// total of value j(l) = (l & 1) * 4 + ((l >> 1) & 1) * 2 + ((l >> 3) & 1); SUM8_LANE(j) is the lowest such lane.
#define SUM8_LANE(j) ((((j) >> 2) & 1) | ((((j) >> 1) & 1) << 1) | (((j) & 1) << 3))
DEVINL uint32_t wave_sum8(const uint32_t (&a)[8]) {
    const uint32_t lane = lane_id();
    const bool b0 = (lane & 1u) != 0, b1 = (lane & 2u) != 0, b3 = (lane & 8u) != 0;
    uint32_t b[4], c[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t keep = b0 ? a[k + 4] : a[k], send = b0 ? a[k] : a[k + 4];
        b[k] = keep + dpp<0xB1>(send);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t keep = b1 ? b[k + 2] : b[k], send = b1 ? b[k] : b[k + 2];
        c[k] = keep + dpp<0x4E>(send);
    }
    const uint32_t keep = b3 ? c[1] : c[0], send = b3 ? c[0] : c[1];
    const uint32_t d = keep + dpp<0x128>(send);
    const uint32_t e = add_xor4(d);
    const auto r16 = __builtin_amdgcn_permlane16_swap(e, e, false, false);
    const uint32_t f = r16[0] + r16[1];
    const auto r32 = __builtin_amdgcn_permlane32_swap(f, f, false, false);
    return r32[0] + r32[1];
}
DEVINL uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
template <int CTRL> DEVINL uint64_t dpp64(uint64_t v) {
    return ((uint64_t)dpp<CTRL>((uint32_t)(v >> 32)) << 32) | dpp<CTRL>((uint32_t)v);
}
DEVINL uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }

// minimum over the four slot groups of the wave_sum8 lane layout (lane ^ 2 and lane ^ 8 exchanges): every lane ends
// up with the minimum of the four distinct per-slot values
DEVINL uint64_t lane_min4(uint64_t v) {
    v = umin64(v, dpp64<0x4E>(v));
    v = umin64(v, dpp64<0x128>(v));
    return v;
}

// minimum of a u64 over the wave, left in EVERY lane (no scalar instructions): rows by DPP, then the gfx950 row / half
// swaps. `opaque()` hides from the compiler that a value is wave-uniform, so that what is computed from it stays on
// the vector ALU (the scalar unit is the bottleneck of the sub-solver loop).
DEVINL uint64_t wave_min_u64_v(uint64_t v) {
    v = umin64(v, dpp64<0xB1>(v));
    v = umin64(v, dpp64<0x4E>(v));
    v = umin64(v, dpp64<0x141>(v));
    v = umin64(v, dpp64<0x140>(v));
    {
        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)v, (uint32_t)v, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(v >> 32), (uint32_t)(v >> 32), false, false);
        v = umin64(((uint64_t)hi[0] << 32) | lo[0], ((uint64_t)hi[1] << 32) | lo[1]);
    }
    {
        const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)v, (uint32_t)v, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(v >> 32), (uint32_t)(v >> 32), false, false);
        v = umin64(((uint64_t)hi[0] << 32) | lo[0], ((uint64_t)hi[1] << 32) | lo[1]);
    }
    return v;
}
DEVINL uint64_t opaque(uint64_t x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return ((uint64_t)hi << 32) | lo;
}

// ---- keys ------------------------------------------------------------------------------------------------
DEVINL bool key_less(const Key& a, const Key& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
DEVINL Key key_inf() { return Key{~0ull, ~0ull}; }
DEVINL Key make_key(uint64_t total, uint32_t hets, uint64_t idx, uint32_t rank, uint32_t depth) {
    return Key{(total << 24) | (uint64_t)(0xFFFFFFu - hets), (idx << 26) | ((uint64_t)rank << 24) | (uint64_t)depth};
}
DEVINL uint64_t key_idx(const Key& k) { return k.lo >> 26; }
DEVINL uint32_t key_rank(const Key& k) { return (uint32_t)(k.lo >> 24) & 3u; }
DEVINL Key wave_min_key(Key k) {  // 128-bit lexicographic minimum (main heap; slow path only)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Key o;
        o.hi = __shfl_xor(k.hi, m);
        o.lo = __shfl_xor(k.lo, m);
        if (key_less(o, k)) k = o;
    }
    return Key{bcast64(k.hi), bcast64(k.lo)};
}
// Sub-solver key, one u64 (astar_phaser.rs:131-133 restricted to a <= 62-variant sub-problem):
//   cost | (63 - hets):6 | node_index:B | depth:6 | rank:2, B = SolveParams::sub_idx_bits:
//     B = 14: cost:36 (host checks cost < 2^36, nodes < 2^14) - every --phase-min-queue-size up to 39 730;
//     B = 20: cost:30 (cost < 2^30, nodes < 2^20) - the wide-index form for larger queues (min_queue_size / 10 +
//             queue_increment x max_segment_size visits, four nodes each);
//   rank = creation rank among the siblings; it sits below node_index, which is unique, so it never decides.
//   key = total << (B + 14) | (63 - hets) << (B + 8) | node_index << 8 | depth << 2 | rank   (built per lane by lane_subkey)
DEVINL uint64_t subkey_total(uint64_t k, uint32_t B) { return k >> (B + 14u); }
DEVINL uint32_t subkey_idx(uint64_t k, uint32_t B) { return (uint32_t)(k >> 8) & ((1u << B) - 1u); }
DEVINL uint32_t subkey_rank(uint64_t k) { return (uint32_t)k & 3u; }
DEVINL uint32_t subkey_depth(uint64_t k) { return (uint32_t)(k >> 2) & 63u; }
// The key of the child in slot `lslot` (a per-lane value) of an expansion, computed entirely on the vector ALU:
// sumT = the lane's frozen + fluid increment of that slot; slots that do not exist get ~0.
//   slot 0 = (0,1)  1 = (1,0) [only if the parent's haplotypes differ]  2 = (0,0)  3 = (1,1);  ignored variant: slot 0 only
DEVINL uint64_t lane_subkey(uint32_t lslot, bool bad, bool has1, uint64_t tbase, uint32_t sumT, uint32_t hets_hom,
                            uint32_t first_idx, uint32_t depth, uint32_t B) {
    // per-slot predicates as arithmetic on the lane's slot number and three uniform words (no per-lane-set masks,
    // which the compiler would hoist out of the loop and then spill)
    const uint32_t valid_set = bad ? 0x1u : (has1 ? 0xFu : 0xDu);      // bit s: slot s exists
    const uint32_t no10 = has1 ? 0u : 1u, real = bad ? 0u : 1u;
    const uint32_t lrank = lslot - ((lslot >> 1) & no10);              // creation rank among the siblings
    const uint32_t lhets = hets_hom + ((~lslot >> 1) & real);          // slots 0/1 of a real expansion: one more het
    const uint64_t total = tbase + sumT;
    const uint64_t low = ((uint64_t)(63u - lhets) << (B + 8u)) | ((first_idx + lrank) << 8) | (depth << 2) | lrank;   // < 2^(B + 14)
    const uint64_t k = (total << (B + 14u)) | low;
    return ((valid_set >> lslot) & 1u) ? k : ~0ull;
}

// ---- record I/O: lane 0 writes, lane 0 reads, broadcast (same-lane rule) ---------------------------------------
DEVINL void store_chunk(ChunkRec* dst, const Win& w0, const Win& w1, uint32_t anc2) {
    if (lane_id() == 0) {
        uint4* d = reinterpret_cast<uint4*>(dst);
        d[0] = make_uint4(w0.h1, w0.h2, w0.nv, w1.h1);
        d[1] = make_uint4(w1.h2, w1.nv, anc2, 0u);
    }
}
DEVINL ChunkRec load_chunk(const ChunkRec* src) {
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (lane_id() == 0) {
        const uint4* s = reinterpret_cast<const uint4*>(src);
        a = s[0];
        b = s[1];
    }
    ChunkRec r;
    r.w0.h1 = bcast32(a.x); r.w0.h2 = bcast32(a.y); r.w0.nv = bcast32(a.z);
    r.w1.h1 = bcast32(a.w); r.w1.h2 = bcast32(b.x); r.w1.nv = bcast32(b.y);
    r.anc2 = bcast32(b.z); r.pad = 0;
    return r;
}
DEVINL FamRec load_fam(const FamRec* src) {
    uint4 a = make_uint4(0, 0, 0, 0), b = a, c = a, d = a, e = a;
    if (lane_id() == 0) {
        const uint4* s = reinterpret_cast<const uint4*>(src);
        a = s[0]; b = s[1]; c = s[2]; d = s[3]; e = s[4];
    }
    FamRec f;
    f.frozen = ((uint64_t)bcast32(a.y) << 32) | bcast32(a.x);
    f.depth_flags = bcast32(a.z); f.hets = bcast32(a.w);
    f.anc1 = bcast32(b.x); f.anc2 = bcast32(b.y); f.base.h1 = bcast32(b.z); f.base.h2 = bcast32(b.w);
    f.base.nv = bcast32(c.x); f.w1.h1 = bcast32(c.y); f.w1.h2 = bcast32(c.z); f.w1.nv = bcast32(c.w);
    f.sumF[0] = bcast32(d.x); f.sumF[1] = bcast32(d.y); f.sumF[2] = bcast32(d.z); f.sumF[3] = bcast32(d.w);
    f.tot[0] = bcast32(e.x); f.tot[1] = bcast32(e.y); f.tot[2] = bcast32(e.z); f.tot[3] = bcast32(e.w);
    return f;
}

// ---- 64-way sharded heaps: lane l owns elements [j*64 + l] -------------------------------------------------
// (a) sub-solver heap: u64 keys, in LDS (or in HBM when the queue parameters make it too large for LDS)
template <bool LDS> struct SubHeap {
    uint64_t* gbase;    // used when !LDS
    uint32_t jcap;      // per-lane capacity
    uint32_t cnt;       // per lane
    uint64_t top;       // cache of the global minimum (~0 when empty): wave-uniform, but kept in vector registers
    uint32_t top_lane;  // its owner lane, likewise
    uint32_t ovf;       // per lane
    uint32_t dealt;     // uniform: keys pushed so far (round-robin target lane)
    DEVINL uint64_t ld(uint32_t j) const {
        if (LDS) return reinterpret_cast<const uint64_t*>(hp_smem + LDS_HEAP_OFF)[j * 64 + lane_id()];
        return gbase[(size_t)j * 64 + lane_id()];
    }
    DEVINL void st(uint32_t j, uint64_t k) {
        if (LDS) reinterpret_cast<uint64_t*>(hp_smem + LDS_HEAP_OFF)[j * 64 + lane_id()] = k;
        else gbase[(size_t)j * 64 + lane_id()] = k;
    }
    DEVINL void reset() { cnt = 0; top = opaque(~0ull); top_lane = 0; dealt = 0; }
    // The queue holds one key per family (expansion) that still has an unpopped child. Keys are dealt to the lanes
    // round-robin by push order (uniform counter `dealt`); every visit deals at most one key, so a lane never
    // holds more than ceil(max_visits / 64) entries.
    DEVINL void deal(uint64_t k) {   // k != ~0
        const uint32_t tgt = dealt & 63u;
        dealt += 1;
        if (lane_id() == tgt) {
            if (cnt >= jcap) ovf = 1;
            else {
                uint32_t j = cnt;
                while (j > 0) {
                    const uint32_t pj = (j - 1) >> 1;
                    const uint64_t pk = ld(pj);
                    if (k < pk) { st(j, pk); j = pj; } else break;
                }
                st(j, k);
                cnt += 1;
            }
        }
        if (k < top) { top = k; top_lane = tgt; }
    }
    // pops the minimum and queues ka (present: it re-uses the freed slot, one sift-down) and kb (~0 = absent, dealt)
    DEVINL void replace_push(uint64_t ka, uint64_t kb) {
        if (lane_id() == top_lane) {
            uint32_t i = 0;
            for (;;) {
                uint32_t ch = 2 * i + 1;
                if (ch >= cnt) break;
                uint64_t ck = ld(ch);
                if (ch + 1 < cnt) {
                    const uint64_t c2 = ld(ch + 1);
                    if (c2 < ck) { ck = c2; ch += 1; }
                }
                if (ck < ka) { st(i, ck); i = ch; } else break;
            }
            st(i, ka);
        }
        const uint64_t root = cnt > 0 ? ld(0) : ~0ull;
        top = wave_min_u64_v(root);
        const uint64_t who = __ballot(cnt > 0 && root == top);
        top_lane = who ? (uint32_t)__builtin_ctzll(who) : 0u;
        if (__any(kb != ~0ull)) deal(kb);
    }
};

// (b) main heap: 128-bit keys in HBM scratch, one heap per lane (a key lives in lane node_index % 64, so the up to three
// children an expansion queues land in three different lanes: one store instruction). Every level of a sift is a dependent
// HBM / L2 round trip and nothing else runs on this wavefront meanwhile, so the heaps are as flat as the wavefront is wide:
// 64-ary. A pop concerns ONE lane's heap (the one that holds the minimum) and the whole wavefront works on it: the 64 children
// of a node are one coalesced 1 KB load (a lane's heap is contiguous), their minimum one DPP reduction - two levels hold 4 161
// keys, i.e. 266 k keys over the 64 lanes in two round trips. The roots are mirrored in registers, so the minimum over the lanes
// after a pop costs no load. (Until round 6: 4-ary heaps sifted by their own lane alone - five or six dependent trips a pop and
// another for the new root, 4 450 shader-clock ticks of a jump's 7 000, scripts/r6_mainprof.py.)
// Insertion is LAZY: a pushed key is appended behind the lane's heap-ordered prefix (one store, no load) and only the
// running minimum `top` is updated; the appended keys are sifted into place when the heap order is actually needed - at
// the next pop or full prune. A dive that never pops from the queue (the common case: the heuristic is exact on clean
// data) never pays the dependent parent loads of a sift-up; a search that does pop pays exactly what it paid before.
// Memory order: a heap's entries are written by its own lane and, in a pop, read by all of them: the pop waits for the
// wavefront's outstanding stores after its own-lane part (workgroup-scope fence = s_waitcnt on a single-wavefront workgroup; the
// wavefront's accesses go through one L1 in issue order) - by then they are a round trip old and the wait is free.
// 128-bit lexicographic minimum over the wave and the first lane that holds it (m in scalar registers): one DPP reduction over
// the high words; the low words only where two lanes tie in the high word (cost and het count equal: uncommon)
DEVINL uint32_t wave_argmin_key(const Key& k, Key& m) {
    const uint64_t mh = wave_min_u64_v(k.hi);
    uint64_t tie = __ballot(k.hi == mh);
    if (__popcll(tie) > 1) {
        const uint64_t ml = wave_min_u64_v(k.hi == mh ? k.lo : ~0ull);
        tie = __ballot(k.hi == mh && k.lo == ml);
    }
    const int u = __builtin_ctzll(tie);
    m.hi = ((uint64_t)rdlane((uint32_t)(k.hi >> 32), u) << 32) | rdlane((uint32_t)k.hi, u);
    m.lo = ((uint64_t)rdlane((uint32_t)(k.lo >> 32), u) << 32) | rdlane((uint32_t)k.lo, u);
    return (uint32_t)u;
}
struct MainHeap {
    Key* base;
    uint32_t jcap, cnt;
    uint32_t hcnt;   // per lane: entries [0, hcnt) are heap-ordered, [hcnt, cnt) are appended and pending
    // per lane: entry 0 of the heap-ordered prefix (the infinite key while it is empty) is mirrored in LDS - where the sub-solver's
    // heap lay during the heuristic phase (1 KB; registers held across the whole search would be spilled in the sub-solver's loop)
    DEVINL Key root() const {
        const uint64_t* r = reinterpret_cast<const uint64_t*>(hp_smem + LDS_HEAP_OFF);
        return Key{r[lane_id()], r[64 + lane_id()]};
    }
    DEVINL void set_root(const Key& k) {
        uint64_t* r = reinterpret_cast<uint64_t*>(hp_smem + LDS_HEAP_OFF);
        r[lane_id()] = k.hi; r[64 + lane_id()] = k.lo;
    }
    Key top;
    uint32_t top_lane, ovf;
    DEVINL Key* own() const { return base + (size_t)lane_id() * jcap; }
    DEVINL Key ld(uint32_t j) const { return own()[j]; }
    DEVINL void st(uint32_t j, const Key& k) { own()[j] = k; }
    DEVINL void reset() { cnt = 0; hcnt = 0; set_root(key_inf()); top = key_inf(); top_lane = 0; }
    DEVINL bool empty() const { return (top.hi & top.lo) == ~0ull; }
    // sift the pending entries of this lane into its heap (every lane runs its own loop: at most two parents above an entry)
    DEVINL void integrate() {
        while (hcnt < cnt) {
            const Key k = ld(hcnt);
            uint32_t j = hcnt;
            while (j > 0) {
                const uint32_t pj = (j - 1) >> 6;
                const Key pk = ld(pj);
                if (key_less(k, pk)) { st(j, pk); j = pj; } else break;
            }
            if (j != hcnt) st(j, k);
            if (j == 0) set_root(k);
            hcnt += 1;
        }
    }
    DEVINL void push(const Key& k) {
        const uint32_t tgt = (uint32_t)key_idx(k) & 63u;
        if (lane_id() == tgt) {
            if (cnt >= jcap) ovf = 1;
            else { st(cnt, k); cnt += 1; }
        }
        if (key_less(k, top)) { top = k; top_lane = tgt; }
    }
    // up to three keys of consecutive node indices (key_inf() = absent): their owner lanes differ
    DEVINL void push3(const Key& ka, const Key& kb, const Key& kc) {
        const uint32_t lane = lane_id();
        Key mine = key_inf();
        const bool pa = (ka.hi & ka.lo) != ~0ull, pb = (kb.hi & kb.lo) != ~0ull, pc = (kc.hi & kc.lo) != ~0ull;
        if (pa && lane == ((uint32_t)key_idx(ka) & 63u)) mine = ka;
        if (pb && lane == ((uint32_t)key_idx(kb) & 63u)) mine = kb;
        if (pc && lane == ((uint32_t)key_idx(kc) & 63u)) mine = kc;
        if ((mine.hi & mine.lo) != ~0ull) {
            if (cnt >= jcap) ovf = 1;
            else { st(cnt, mine); cnt += 1; }
        }
        if (pa && key_less(ka, top)) { top = ka; top_lane = (uint32_t)key_idx(ka) & 63u; }
        if (pb && key_less(kb, top)) { top = kb; top_lane = (uint32_t)key_idx(kb) & 63u; }
        if (pc && key_less(kc, top)) { top = kc; top_lane = (uint32_t)key_idx(kc) & 63u; }
    }
    // astar_phaser.rs:576-581: every queued node with depth < min_progress gets cost 0. Clearing is a decrease-key,
    // so each lane scans its heap front to back (independent loads) and sifts UP only the entries that
    // are newly cleared — the pop order is a total order on the keys, so the heap's internal layout is free.
    DEVINL void clear_below(uint32_t min_progress) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        integrate();
        for (uint32_t j0 = 0; j0 < cnt; j0 += 4) {
            Key k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = (j0 + u < cnt) ? ld(j0 + u) : key_inf();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if ((uint32_t)(k[u].lo & 0xFFFFFFu) < min_progress && (k[u].hi >> 24) != 0) {
                    Key c = k[u];
                    c.hi &= 0xFFFFFFull;
                    uint32_t j = j0 + u;
                    while (j > 0) {
                        const uint32_t pj = (j - 1) >> 6;
                        const Key pk = ld(pj);
                        if (key_less(c, pk)) { st(j, pk); j = pj; } else break;
                    }
                    st(j, c);
                    if (j == 0) set_root(c);
                }
            }
        }
        recompute_top();
    }
    DEVINL void recompute_top() {   // every pending entry has been integrated: the minimum is the least of the lanes' roots
        top_lane = wave_argmin_key(root(), top);   // (an empty lane's root is the infinite key; all empty: lane 0, as before)
    }
    DEVINL void pop() {
        integrate();   // the lane minima below are the heap roots (own-lane loads and stores only)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // what follows reads lane T's entries from every lane
        // the heap of lane `top_lane` loses its root; all 64 lanes sift its last entry down from there
        const uint32_t T = bcast32(top_lane), lane = lane_id();
        Key* const heap = base + (size_t)T * jcap;
        const uint32_t n = bcast32(__builtin_amdgcn_readlane(cnt, T)) - 1u;   // what is left of it
        Key nroot = key_inf();
        if (n > 0) {
            Key k = key_inf();
            if (lane == 0) k = heap[n];
            uint32_t i = 0;
            bool have_k = false;
            for (;;) {
                const uint32_t c0 = 64u * i + 1u;
                if (c0 >= n) break;
                const Key ck = (c0 + lane < n) ? heap[c0 + lane] : key_inf();
                if (!have_k) { k = Key{bcast64(k.hi), bcast64(k.lo)}; have_k = true; }
                Key m;   // (uniform values in scalar registers: the loop's control flow is scalar)
                const uint32_t u = wave_argmin_key(ck, m);
                if (!key_less(m, k)) break;
                if (lane == T) heap[i] = m;
                if (i == 0) nroot = m;
                i = c0 + u;
            }
            if (!have_k) k = Key{bcast64(k.hi), bcast64(k.lo)};
            if (lane == T) heap[i] = k;
            if (i == 0) nroot = k;
        }
        if (lane == T) { cnt = n; hcnt = n; set_root(nroot); }
        recompute_top();
    }
};

// ---- search node held in (uniform) registers ---------------------------------------------------------------
struct Cur {
    uint64_t frozen, total, idx;
    uint32_t depth, hets, anc1, anc2;
    Win w0, w1;
};
DEVINL Win fresh_win() { return Win{0u, 0u, 0xFFFFFFFFu}; }
DEVINL Cur root_node(uint64_t heur) {
    Cur n;
    n.frozen = 0; n.total = heur; n.idx = 0; n.depth = 0; n.hets = 0; n.anc1 = NONE32; n.anc2 = NONE32;
    n.w0 = fresh_win(); n.w1 = fresh_win();
    return n;
}
// rebuilds a queued child from its family record and creation rank (slow path: the node was popped from the heap)
DEVINL Cur cur_from_fam(const FamRec& f, uint32_t rank, uint64_t total, uint64_t idx, uint32_t off) {
    const bool bad = (f.depth_flags >> 30) & 1u, has1 = (f.depth_flags >> 31) & 1u;
    const uint32_t pdepth = f.depth_flags & 0xFFFFFFu;
    const uint32_t slot = bad ? 0u : (has1 ? rank : (rank == 0 ? 0u : rank + 1u));
    const uint32_t bit = 1u << ((off + pdepth) & 31u);
    Cur n;
    n.frozen = f.frozen + (slot == 0 ? f.sumF[0] : slot == 1 ? f.sumF[1] : slot == 2 ? f.sumF[2] : f.sumF[3]);
    n.total = total; n.idx = idx; n.depth = pdepth + 1;
    n.hets = f.hets + ((slot <= 1 && !bad) ? 1u : 0u);
    n.anc1 = f.anc1; n.anc2 = f.anc2;
    n.w0 = f.base;
    if (!bad) {
        n.w0.nv &= ~bit;
        if (slot == 1 || slot == 3) n.w0.h1 |= bit;
        if (slot == 0 || slot == 3) n.w0.h2 |= bit;
    }
    n.w1 = f.w1;
    return n;
}

struct WaveCounters {
    uint64_t sub_pops, main_pops, nodes;
    uint64_t seg[6];   // HP_SEG_PROFILE: shader-clock cycles per sub-solver segment (prm.seg_profile != 0)
    uint64_t tlast;
};
// segment profiling (tuning aid; PROF is a compile-time switch so the production kernel carries none of its state):
// drain outstanding memory ops, read s_memtime
template <bool PROF>
DEVINL void seg_stamp(WaveCounters& wc, int which) {
    if (PROF) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const uint64_t t = __builtin_amdgcn_s_memtime();
        if (which >= 0) wc.seg[which] += t - wc.tlast;
        wc.tlast = t;
    }
}

struct Pools {   // per-wave scratch of one search (sub-solver or main)
    FamRec* fam;        // [cap] indexed by the node_index of an expansion's first child
    ChunkRec* chunk;    // [cap_chunk] appended at chunk transitions
    uint32_t cap_chunk;
    uint32_t n_chunk;   // uniform append cursor
    uint32_t ovf;       // uniform: chunk pool exhausted
    uint4* vec;         // sub-solver only: [cap][64 lanes] saved prefix scores of every expansion (FastState)
};

struct Ctx {
    const uint32_t *rstart, *rend, *rword;
    const uint32_t* words;
    uint32_t n_rows;
    uint32_t tiles;         // 64-entry tiles per variant in the cell table (1 or 2)
    const uint32_t* ctab;   // per-position cell table of the block (hp_astar_dev.h CELL_*), nullptr with HP_NO_CTAB
    uint32_t cring_off;     // CR (below): byte offset in LDS of the staged window of the cell table, 64 variants x 64 entries x u32
    uint32_t N;
    uint64_t evals, cells;  // per-lane work counters
    uint32_t ev32, cl32;    // ... their 32-bit front end (one sub-solve / one main pop), folded in by flush()
    DEVINL void flush() { evals += ev32; cells += cl32; ev32 = 0; cl32 = 0; }
};

// The children of one expansion, fixed slots in the reference's hap_order (astar_phaser.rs:367-372):
//   slot 0 = (0,1)   slot 1 = (1,0)   slot 2 = (0,0)   slot 3 = (1,1);   an ignored variant: slot 0 = (2,2) only.
struct Kids {
    bool bad, has1;        // has1: the (1,0) child exists (parent haplotypes differ <=> hets != 0)
    uint32_t n;            // number of children created
    uint64_t pfrozen;      // parent's frozen cost:  child frozen = pfrozen + sumF<slot>
    uint64_t tbase;        // pfrozen + H[child depth]: child total = tbase + sumT<slot>
    uint32_t depth, anc1, anc2, hets_het, hets_hom;
    uint32_t sumF0, sumF1, sumF2, sumF3;   // frozen increments (what the family record keeps)
    uint32_t sumT0, sumT1, sumT2, sumT3;   // frozen + fluid increments
    uint32_t tvec;         // PER LANE: lane l holds the frozen + fluid increment of slot ((l >> 1) & 1) * 2 + ((l >> 3) & 1)
    uint32_t gvec;         // PER LANE: wave_sum8's result (even lanes: the frozen increment of the lane's slot)
    Win base, w1;          // base = parent's window in the child's chunk (fresh when a new chunk opens)
    uint32_t bit;          // 1 << (p & 31)
};
template <int S> DEVINL uint32_t kid_sumT(const Kids& k) { return S == 0 ? k.sumT0 : S == 1 ? k.sumT1 : S == 2 ? k.sumT2 : k.sumT3; }
template <int S> DEVINL uint32_t kid_sumF(const Kids& k) { return S == 0 ? k.sumF0 : S == 1 ? k.sumF1 : S == 2 ? k.sumF2 : k.sumF3; }
template <int S> DEVINL uint64_t kid_total(const Kids& k) { return k.tbase + kid_sumT<S>(k); }
template <int S> DEVINL uint64_t kid_frozen(const Kids& k) { return k.pfrozen + kid_sumF<S>(k); }
template <int S> DEVINL bool kid_valid(const Kids& k) { return S == 0 ? true : k.bad ? false : (S == 1 ? k.has1 : true); }
template <int S> DEVINL uint32_t kid_hets(const Kids& k) { return (S <= 1 && !k.bad) ? k.hets_het : k.hets_hom; }
// creation rank of slot S among the children of this expansion (node_index = next_idx + rank)
template <int S> DEVINL uint32_t kid_rank(const Kids& k) { return S == 0 ? 0u : S == 1 ? 1u : (S == 2 ? (k.has1 ? 2u : 1u) : (k.has1 ? 3u : 2u)); }
template <int S> DEVINL Win kid_win(const Kids& k) {
    Win w = k.base;
    if (!k.bad) {
        w.nv &= ~k.bit;
        if (S == 1 || S == 3) w.h1 |= k.bit;   // a1 == 1
        if (S == 0 || S == 3) w.h2 |= k.bit;   // a2 == 1
    }
    return w;
}
template <int S> DEVINL Cur kid_as_cur(const Kids& k, uint64_t next_idx) {
    Cur n;
    n.frozen = kid_frozen<S>(k); n.total = kid_total<S>(k); n.idx = next_idx + kid_rank<S>(k);
    n.depth = k.depth; n.hets = kid_hets<S>(k); n.anc1 = k.anc1; n.anc2 = k.anc2;
    n.w0 = kid_win<S>(k); n.w1 = k.w1;
    return n;
}
// ... the same minus the two cost fields (the sub-solver takes those from the key / the lane vector)
template <int S> DEVINL void kid_into_cur(const Kids& k, uint64_t next_idx, Cur& n) {
    n.idx = next_idx + kid_rank<S>(k);
    n.depth = k.depth; n.hets = kid_hets<S>(k); n.anc1 = k.anc1; n.anc2 = k.anc2;
    n.w0 = kid_win<S>(k); n.w1 = k.w1;
}
// Sub-solver form of fam_store: lane 0 writes the 48 bytes the siblings share, the four even lanes that hold the
// slots' sums (wave_sum8 layout) write sumF[slot] / tot[slot] themselves - no trip through scalar registers.
DEVINL void fam_store_lanes(FamRec* fam, const Kids& k, const Cur& parent, uint32_t next_idx) {
    const uint32_t lane = lane_id();
    FamRec* rec = fam + next_idx;
    if (lane == 0) {
        uint4* d = reinterpret_cast<uint4*>(rec);
        d[0] = make_uint4((uint32_t)parent.frozen, (uint32_t)(parent.frozen >> 32),
                          parent.depth | (k.bad ? (1u << 30) : 0u) | (k.has1 ? (1u << 31) : 0u), parent.hets);
        d[1] = make_uint4(k.anc1, k.anc2, k.base.h1, k.base.h2);
        d[2] = make_uint4(k.base.nv, k.w1.h1, k.w1.h2, k.w1.nv);
    }
    if ((lane & ~10u) == 0u) {   // lanes 0, 2, 8, 10
        const uint32_t slot = ((lane >> 1) & 1u) * 2u + ((lane >> 3) & 1u);
        uint32_t* w = reinterpret_cast<uint32_t*>(rec);
        w[12 + slot] = k.gvec;   // FamRec::sumF[slot]
        w[16 + slot] = k.tvec;   // FamRec::tot[slot]
    }
}
// one 80-byte family record per expansion, written by lane 0 at fam[node_index of the first child]
DEVINL void fam_store(FamRec* fam, const Kids& k, const Cur& parent, uint32_t next_idx) {
    if (lane_id() == 0) {
        uint4* d = reinterpret_cast<uint4*>(fam + next_idx);
        d[0] = make_uint4((uint32_t)parent.frozen, (uint32_t)(parent.frozen >> 32),
                          parent.depth | (k.bad ? (1u << 30) : 0u) | (k.has1 ? (1u << 31) : 0u), parent.hets);
        d[1] = make_uint4(k.anc1, k.anc2, k.base.h1, k.base.h2);
        d[2] = make_uint4(k.base.nv, k.w1.h1, k.w1.h2, k.w1.nv);
        d[3] = make_uint4(k.sumF0, k.sumF1, k.sumF2, k.sumF3);
        d[4] = make_uint4(k.sumT0, k.sumT1, k.sumT2, k.sumT3);
    }
}

// weighted popcount: sum_b popc(M & Q_b) << b   (Horner over the 8 quality bit-planes)
DEVINL uint32_t wpop(uint32_t M, uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, uint32_t q4, uint32_t q5,
                     uint32_t q6, uint32_t q7) {
    uint32_t s = (uint32_t)__popc(M & q7);
    s = (s << 1) + (uint32_t)__popc(M & q6);
    s = (s << 1) + (uint32_t)__popc(M & q5);
    s = (s << 1) + (uint32_t)__popc(M & q4);
    s = (s << 1) + (uint32_t)__popc(M & q3);
    s = (s << 1) + (uint32_t)__popc(M & q2);
    s = (s << 1) + (uint32_t)__popc(M & q1);
    s = (s << 1) + (uint32_t)__popc(M & q0);
    return s;
}

// new_extended_node (astar_phaser.rs:69-119) for all children of `cur` at once.
// Row r covering p contributes min(score(h1'), score(h2')) where h' = parent prefix + child allele:
//   score(h') = S(parent prefix over [max(start_r, off), p)) + (allele_r[p] != a ? qual_r[p] : 0)
// so the O(overlap) part S is shared by the children. Two ways to get it:
//   * expand():      from scratch, bit-parallel per 32-variant plane word (any node, any coverage);
//   * expand_fast(): carried in two registers per lane from the parent's expansion when the node being expanded
//                    is the child the previous expansion kept (86 % of the sub-solver's pops on HiFi-like data):
//                    S(child) = S(parent) + the child's own cell, read from the per-position cell table.
// Rows are dealt to lanes by (row index mod 64) in both, so a row keeps its lane from one position to the next.
// Per-lane incremental state: prefix scores of the lane's row in each of the (up to two) 64-row tiles of the cell
// table, and the cost of the row's cell at the variant being expanded for haplotype allele 0 / 1.
struct FastState { uint32_t s1a, s2a, s1b, s2b; };
struct CellCost { uint32_t x0a, x1a, x0b, x1b; };
template <int TILES>
DEVINL void fast_apply(FastState& f, const CellCost& c, bool a1, bool a2) {   // the kept child carries alleles (a1, a2)
    f.s1a += a1 ? c.x1a : c.x0a; f.s2a += a2 ? c.x1a : c.x0a;
    if (TILES == 2) { f.s1b += a1 ? c.x1b : c.x0b; f.s2b += a2 ? c.x1b : c.x0b; }   // second tile: dead state otherwise
}

struct ExpPre {
    bool trans;
    uint32_t new_chunk, nkids, bp;
    Win W0, W1, W2;
};
DEVINL ExpPre expand_begin(const Cur& cur, uint32_t off, uint32_t p, bool bad, Pools& pl) {
    ExpPre e;
    const uint32_t kp = p >> 5;
    e.bp = p & 31u;
    const uint32_t ck = cur.depth ? ((off + cur.depth - 1) >> 5) : (off >> 5);
    e.trans = (ck != kp);  // the child opens a new 32-variant chunk
    e.new_chunk = NONE32;
    if (e.trans) {  // cur's window is a complete chunk that its descendants link to: persist it as a ChunkRec
        if (pl.n_chunk >= pl.cap_chunk) pl.ovf = 1;
        else {
            e.new_chunk = pl.n_chunk++;
            store_chunk(pl.chunk + e.new_chunk, cur.w0, cur.w1, cur.anc2);
        }
    }
    e.W0 = e.trans ? fresh_win() : cur.w0;
    e.W1 = e.trans ? cur.w0 : cur.w1;
    e.W2 = cur.w1;  // only meaningful when trans
    e.nkids = bad ? 1u : (cur.hets != 0 ? 4u : 3u);
    return e;
}
DEVINL void expand_finish(const ExpPre& e, const Cur& cur, bool bad, uint64_t h_next, uint32_t g, Kids& kd) {
    kd.bad = bad;
    kd.has1 = !bad && cur.hets != 0;
    kd.n = e.nkids;
    kd.depth = cur.depth + 1;
    kd.anc1 = e.trans ? e.new_chunk : cur.anc1;
    kd.anc2 = e.trans ? cur.anc1 : cur.anc2;
    kd.w1 = e.W1;
    kd.base = e.W0;
    kd.bit = 1u << e.bp;
    kd.hets_het = cur.hets + 1;
    kd.hets_hom = cur.hets;
    // g (wave_sum8): even lanes hold the frozen sum of their slot, the odd neighbour its frozen + fluid sum
    uint32_t o = dpp<0xB1>(g);
    asm volatile("" : "+v"(o));   // keep the lane exchange out of the select below: a DPP read under a partial EXEC
                                  // mask sees the masked-off neighbours as zero
    kd.tvec = (lane_id() & 1u) ? g : o;
    kd.gvec = g;
    kd.sumF0 = rdlane(g, SUM8_LANE(0)); kd.sumF1 = rdlane(g, SUM8_LANE(1)); kd.sumF2 = rdlane(g, SUM8_LANE(2)); kd.sumF3 = rdlane(g, SUM8_LANE(3));
    kd.sumT0 = rdlane(g, SUM8_LANE(4)); kd.sumT1 = rdlane(g, SUM8_LANE(5)); kd.sumT2 = rdlane(g, SUM8_LANE(6)); kd.sumT3 = rdlane(g, SUM8_LANE(7));
    kd.pfrozen = cur.frozen;
    kd.tbase = cur.frozen + h_next;
}
// the per-row part shared by both paths: child costs from (S1, S2) and the cell at p, split into frozen / fluid
DEVINL void row_costs(uint32_t s1, uint32_t s2, uint32_t x0, uint32_t x1, bool frozen, uint32_t (&acc)[8]) {
    const uint32_t c0 = min(s1 + x0, s2 + x1);  // (0,1)  [== min(s1, s2) for the (2,2) child]
    const uint32_t c1 = min(s1 + x1, s2 + x0);  // (1,0)
    const uint32_t c2 = min(s1 + x0, s2 + x0);  // (0,0)
    const uint32_t c3 = min(s1 + x1, s2 + x1);  // (1,1)
    // [0..3] frozen part per slot (rows whose last cell is this one), [4..7] frozen + fluid per slot
    acc[0] += frozen ? c0 : 0u; acc[4] += c0;
    acc[1] += frozen ? c1 : 0u; acc[5] += c1;
    acc[2] += frozen ? c2 : 0u; acc[6] += c2;
    acc[3] += frozen ? c3 : 0u; acc[7] += c3;
}

// lo = first candidate row of variant p (rows are start-sorted; the candidates end with the last row whose start
// is <= p), bad = variant ignored. Leaves, per lane, the scores of the
// parent prefix and the cell costs of its row(s) (zeros where it has none).
template <bool PROF, int TILES>
DEVINL void expand(Ctx& cx, const Cur& cur, uint32_t off, uint32_t p, uint32_t lo, bool bad,
                   uint64_t h_next, Pools& pl, Kids& kd, WaveCounters& wc, FastState& fs, CellCost& cc) {
    const uint32_t lane = lane_id();
    const ExpPre e = expand_begin(cur, off, p, bad, pl);
    const uint32_t kp = p >> 5, bp = e.bp;
    const bool trans = e.trans;
    const Win W0 = e.W0, W1 = e.W1, W2 = e.W2;
    const uint32_t nkids = e.nkids;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [0..3] frozen per slot, [4..7] frozen + fluid per slot
    fs = FastState{0, 0, 0, 0};
    cc = CellCost{0, 0, 0, 0};

    for (uint32_t base = lo;; base += 64) {
        const uint32_t r = base + ((lane - base) & 63u);   // lane == r mod 64
        const bool inb = r < cx.n_rows;
        uint32_t rs = 0xFFFFFFFFu, re = 0, rw = 0;
        if (inb) {
            rs = cx.rstart[r];
            re = cx.rend[r];
            rw = cx.rword[r];
        }
        const bool started = inb && rs <= p;   // rows are sorted by start: the candidates are a prefix of the tile
        const bool valid = started && re > p;
        seg_stamp<PROF>(wc, 1);   // [1] row metadata loads
        const uint32_t kr = rs >> 5;
        const uint32_t myj = kp - kr;  // words before the one holding p (garbage when !valid)
        uint32_t s1 = 0, s2 = 0, ap = 3, qp = 0;
        const uint32_t* wbase = cx.words + (size_t)(rw + (kp - kr)) * WORD_DWORDS;  // plane word holding p
        // ---- j = 0: the word that holds p (every valid lane) ------------------------------------------------
        if (valid) {
            const uint4* pw = reinterpret_cast<const uint4*>(wbase);
            const uint4 x0 = pw[0], x1 = pw[1], x2 = pw[2];
            const uint32_t aLo = x0.x, aHi = x0.y;
            const uint32_t M1 = ~W0.nv & ((aLo ^ W0.h1) | aHi);
            const uint32_t M2 = ~W0.nv & ((aLo ^ W0.h2) | aHi);
            s1 = wpop(M1, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
            s2 = wpop(M2, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
            ap = ((aLo >> bp) & 1u) | (((aHi >> bp) & 1u) << 1);
            qp = ((x0.z >> bp) & 1u) | (((x0.w >> bp) & 1u) << 1) | (((x1.x >> bp) & 1u) << 2) |
                 (((x1.y >> bp) & 1u) << 3) | (((x1.z >> bp) & 1u) << 4) | (((x1.w >> bp) & 1u) << 5) |
                 (((x2.x >> bp) & 1u) << 6) | (((x2.y >> bp) & 1u) << 7);
        }
        // ---- j = 1: the previous chunk (rows that started before this chunk) -------------------------------
        const bool need1 = valid && myj >= 1;
        if (need1) {
            const uint4* pw = reinterpret_cast<const uint4*>(wbase - WORD_DWORDS);
            const uint4 x0 = pw[0], x1 = pw[1], x2 = pw[2];
            const uint32_t M1 = ~W1.nv & ((x0.x ^ W1.h1) | x0.y);
            const uint32_t M2 = ~W1.nv & ((x0.x ^ W1.h2) | x0.y);
            s1 += wpop(M1, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
            s2 += wpop(M2, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
        }
        // ---- j >= 2: rows spanning more than two chunks walk the haplotype-window chain (rare) --------------
        if (__any(valid && myj >= 2)) {
            uint32_t chain_slot = cur.anc2, chain_phase = 0, chain_next = NONE32, guard = 0;
            Win cw1 = fresh_win();
            for (uint32_t j = 2;; ++j) {
                const bool need = valid && (j <= myj);
                if (!__any(need)) break;
                Win w;
                if (trans && j == 2) w = W2;
                else {
                    if (chain_phase == 0) {
                        if (chain_slot == NONE32 || ++guard > (cx.N >> 6) + 4) break;  // nothing older: zero cost
                        const ChunkRec a = load_chunk(pl.chunk + chain_slot);
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
                    const uint4* pw = reinterpret_cast<const uint4*>(wbase - (size_t)j * WORD_DWORDS);
                    const uint4 x0 = pw[0], x1 = pw[1], x2 = pw[2];
                    const uint32_t M1 = ~w.nv & ((x0.x ^ w.h1) | x0.y);
                    const uint32_t M2 = ~w.nv & ((x0.x ^ w.h2) | x0.y);
                    s1 += wpop(M1, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
                    s2 += wpop(M2, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
                }
            }
        }
        if (valid) {
            const uint32_t x0 = (!bad && ap != 0u) ? qp : 0u;  // cost of giving a haplotype allele 0 here
            const uint32_t x1 = (!bad && ap != 1u) ? qp : 0u;  // ... allele 1
            row_costs(s1, s2, x0, x1, re == p + 1 /* rs.region().end <= hap_len (astar_phaser.rs:101) */, acc);
            cx.ev32 += nkids;
            cx.cl32 += nkids * (p + 1 - max(rs, off));
            // seed the incremental state (a lane has one live row per tile unless the variant is VAR_NOFAST)
            if (TILES == 2 && cx.tiles == 2u && (r & 64u)) { fs.s1b = s1; fs.s2b = s2; cc.x0b = x0; cc.x1b = x1; }
            else { fs.s1a = s1; fs.s2a = s2; cc.x0a = x0; cc.x1a = x1; }
        }
        if (base != lo) cx.flush();   // > 64 candidate rows (rare): keep the 32-bit counters far from wrapping
        if (!__all(started)) break;   // a lane ran past the last candidate: no further tile
    }
    seg_stamp<PROF>(wc, 2);       // [2] plane-word loads + bit-sliced scoring
    const uint32_t g = wave_sum8(acc);
    seg_stamp<PROF>(wc, 3);       // [3] wave reduction
    expand_finish(e, cur, bad, h_next, g, kd);
}

// The same expansion when `fs` already holds, per lane, the scores of cur's haplotype prefix against the lane's row(s)
// (see above): one coalesced 256-byte read per tile of the cell table row of p replaces the row metadata and plane
// words. Blocks with up to 64 candidate rows per variant use one tile, others two (row index mod 128).
DEVINL void fast_tile(Ctx& cx, uint32_t cell, uint32_t p, uint32_t off, uint32_t nkids, uint32_t& fs1, uint32_t& fs2,
                      uint32_t& px0, uint32_t& px1, uint32_t (&acc)[8]) {
    const bool valid = (cell & CELL_VALID) != 0;
    const uint32_t t = cell >> CELL_T_SHIFT;          // p - row start (saturated): 0 = the row starts here
    if (t == 0) { fs1 = 0; fs2 = 0; }
    const uint32_t x0 = cell & 0xFFu, x1 = (cell >> 8) & 0xFFu;   // both 0 for an empty entry or an ignored variant
    const uint32_t s1 = valid ? fs1 : 0u, s2 = valid ? fs2 : 0u;
    row_costs(s1, s2, x0, x1, (cell & CELL_ENDS) != 0, acc);
    if (valid) {
        cx.ev32 += nkids;
        cx.cl32 += nkids * (min(t, p - off) + 1u);    // == p + 1 - max(row start, off)
    }
    px0 = x0; px1 = x1;
}
// CR (round 6; BASELINE.json north_star: "the allele matrix packed 2-bit and staged into LDS"): the sub-solver only ever looks at the
// <= 40 variants of its window (astar_phaser.rs:311-405: problem_size <= max_segment_size), and the heuristic chain moves that window
// down one variant per step - so the window's rows of the cell table (per variant 64 entries: a row's 2-bit allele + its u8 quality +
// the ends-here / valid flags, hp_astar_dev.h CELL_*) live in a 64-variant LDS ring that the chain feeds with ONE coalesced 256-byte
// read per step (solve_segment), and every expansion of the sub-solver - a hundred pops per step, each of which used to read its
// variant's row from L1 / L2 - takes its row from LDS. 16 KB per wavefront: 8 single-wave workgroups per CU instead of 24, which
// is what the segment kernel needs anyway (two per SIMD run at the speed of one, profiles/round2/issue_ceiling.txt). The full matrix
// of a C2 block (188 KB) would not fit; the main search, whose look-back is unbounded, keeps reading the table where it lies.
template <bool PROF, int TILES, bool CR = false>
DEVINL void expand_fast(Ctx& cx, const Cur& cur, uint32_t off, uint32_t p, bool bad, uint64_t h_next, Pools& pl,
                        Kids& kd, WaveCounters& wc, FastState& fs, CellCost& cc) {
    const bool two = TILES == 2 && cx.tiles == 2u;   // TILES: what the launch supports, cx.tiles: this block's table
    const uint32_t* row = cx.ctab + ((size_t)p << (two ? 7 : 6)) + lane_id();
    const uint32_t cell_a = (CR && TILES == 1) ? reinterpret_cast<const uint32_t*>(hp_smem + cx.cring_off)[((p & 63u) << 6) + lane_id()] : row[0];
    const uint32_t cell_b = two ? row[64] : 0u;
    const ExpPre e = expand_begin(cur, off, p, bad, pl);
    seg_stamp<PROF>(wc, 1);
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    fast_tile(cx, cell_a, p, off, e.nkids, fs.s1a, fs.s2a, cc.x0a, cc.x1a, acc);
    if (two) fast_tile(cx, cell_b, p, off, e.nkids, fs.s1b, fs.s2b, cc.x0b, cc.x1b, acc);
    else if (TILES == 2) { cc.x0b = 0; cc.x1b = 0; }
    seg_stamp<PROF>(wc, 2);
    const uint32_t g = wave_sum8(acc);
    seg_stamp<PROF>(wc, 3);
    expand_finish(e, cur, bad, h_next, g, kd);
}

// LDS rings, indexed by (variant & 63): H[x] and the per-variant (lo, hi, flags) triple. Written by lane 0, read by
// all lanes at one address (LDS operations of a wave execute in order), then made scalar with v_readfirstlane.
DEVINL uint64_t ringH_get(uint32_t x) {  // every lane reads the same address: one broadcast LDS access, no exec masking
    return bcast64(reinterpret_cast<const uint64_t*>(hp_smem + LDS_HRING_OFF)[x & 63u]);
}
DEVINL void ringH_set(uint32_t x, uint64_t v) {
    if (lane_id() == 0) reinterpret_cast<uint64_t*>(hp_smem + LDS_HRING_OFF)[x & 63u] = v;
}
DEVINL void ringV_set(uint32_t x, uint32_t lo, uint32_t flags) {  // rows per block < 2^28 (host check)
    if (lane_id() == 0) reinterpret_cast<uint32_t*>(hp_smem + LDS_VRING_OFF)[x & 63u] = lo | (flags << 28);
}
// astar_subsolver (astar_phaser.rs:311-405). Returns status; outputs (max_cost_so_far, farthest).
template <bool SUB_LDS, bool PROF, int TILES, bool CR = false>
DEVINL int32_t subsolve(Ctx& cx, const SolveParams& prm, uint32_t off, uint32_t ps, SubHeap<SUB_LDS>& heap,
                        Pools& pl, WaveCounters& wc, uint64_t& est, uint32_t& solved) {
    heap.reset();
    pl.n_chunk = 0;
    seg_stamp<PROF>(wc, -1);
    uint32_t next_idx = 1;
    Cur cur = root_node(ringH_get(off + 1));  // initial_estimate = H[off+1] (astar_phaser.rs:322)
    uint32_t next_expected = 0, visited = 0;
    uint64_t max_cost = opaque(0);   // wave-uniform, updated on the vector ALU
    const uint32_t max_visits = prm.minq_sub + prm.qinc * ps;
    int32_t st = ST_OK;
    // (fs1, fs2): per-lane scores of cur's prefix against the lane's row, valid while cur is the child the previous
    // expansion kept in registers; the root has scored nothing yet, so it starts valid (all zero)
    const bool fast_ok = cx.ctab != nullptr;
    bool fast_valid = fast_ok;
    FastState fs{0, 0, 0, 0};
    while (cur.depth < ps && visited < max_visits) {
        visited += 1;
        if (cur.depth == next_expected) {
            max_cost = max(max_cost, cur.total);
            next_expected += 1;
        }
        const uint32_t p = off + cur.depth;
        // both ring reads are issued before either result is consumed
        const uint32_t rv = reinterpret_cast<const uint32_t*>(hp_smem + LDS_VRING_OFF)[p & 63u];
        const uint64_t rh = reinterpret_cast<const uint64_t*>(hp_smem + LDS_HRING_OFF)[(p + 1) & 63u];
        const uint32_t low = bcast32(rv), lo = low & 0x0FFFFFFFu, flags = low >> 28;
        Kids kd;
        seg_stamp<PROF>(wc, 0);       // [0] loop head + LDS ring reads
        CellCost cc;
        const bool collide = (flags & VAR_NOFAST) != 0;   // two rows of this variant on one lane: plane-word path only
        if (fast_valid && !collide) expand_fast<PROF, TILES, CR>(cx, cur, off, p, (flags & HP_VAR_IGNORED) != 0, bcast64(rh), pl, kd, wc, fs, cc);
        else expand<PROF, TILES>(cx, cur, off, p, lo, (flags & HP_VAR_IGNORED) != 0, bcast64(rh), pl, kd, wc, fs, cc);
        // (no capacity check: next_idx <= 4 * visited + 1 <= 4 * max_visits + 1 == cap_sub, see hp_batch_create)
        if (kd.bad && kd.tbase + rdlane(kd.tvec, 0) != cur.total) { st = ST_INVARIANT; break; }  // astar_phaser.rs:360
        // High-coverage launches (TILES == 2) keep the incremental state of every expansion (prefix scores of cur, rows
        // starting here already reset) with the family: a sibling popped from the queue later resumes from it instead
        // of re-scoring two tiles of plane words. Measured: at coverage 60 that is +20 %, at coverage 30 the extra
        // 512 B written per pop cost more than the rebuilds they save (58 -> 51 M hets/s), hence the compile-time gate.
        const bool SAVE_STATE = TILES == 2 && prm.save_state != 0;   // the host sized the pool for it
        if (SAVE_STATE && fast_ok && !collide)
            pl.vec[(size_t)next_idx * 64 + lane_id()] = make_uint4(fs.s1a, fs.s2a, fs.s1b, fs.s2b);
        // Keys of the (up to 4) children, one slot per lane group, on the vector ALU: the scalar unit is the busier of
        // the two pipes in this loop (measured: an extra scalar instruction per pop costs four times an extra vector
        // one), and lane l already holds the cost sum of slot s(l) = ((l >> 1) & 1) * 2 + ((l >> 3) & 1).
        const uint32_t lslot = ((lane_id() >> 1) & 1u) * 2u + ((lane_id() >> 3) & 1u);
        const uint32_t KB = prm.sub_idx_bits;
        const uint64_t kl = lane_subkey(lslot, kd.bad, kd.has1, kd.tbase, kd.tvec, kd.hets_hom, next_idx, kd.depth, KB);
        const uint64_t kbest = bcast64(lane_min4(kl));
        // the family's next key should kbest leave it: its smallest other child (~0: none)
        const uint64_t ksecond = lane_min4(kl == kbest ? ~0ull : kl);   // stays in vector registers (uniform value)
        const uint32_t best_lane = (uint32_t)__builtin_ctzll(__ballot(kl == kbest));   // lowest lane of the best slot
        const uint32_t best = ((best_lane >> 1) & 1u) * 2u + ((best_lane >> 3) & 1u);
        // If the best child beats everything queued it is the next pop: keep it in registers (push + pop elided;
        // the priority is a total order, so this is exactly what the reference's queue would return).
        const bool take_child = __any(kbest < heap.top);   // heap.top lives in vector registers (uniform value)
        seg_stamp<PROF>(wc, 4);   // [4] child totals + keys
        fam_store_lanes(pl.fam, kd, cur, next_idx);  // one record for all siblings
        if (take_child) {
            if (__any(ksecond != ~0ull)) heap.deal(ksecond);
            // slots (a1,a2): 0 = (0,1), 1 = (1,0), 2 = (0,0), 3 = (1,1); the kept child's own cell joins the prefix scores
            cur.frozen = kd.pfrozen + (uint32_t)__builtin_amdgcn_readlane((int)kd.gvec, (int)best_lane);   // best_lane is even
            cur.total = subkey_total(kbest, KB);
            if (best == 0u) { kid_into_cur<0>(kd, next_idx, cur); fast_apply<TILES>(fs, cc, false, true); }
            else if (best == 1u) { kid_into_cur<1>(kd, next_idx, cur); fast_apply<TILES>(fs, cc, true, false); }
            else if (best == 2u) { kid_into_cur<2>(kd, next_idx, cur); fast_apply<TILES>(fs, cc, false, false); }
            else { kid_into_cur<3>(kd, next_idx, cur); fast_apply<TILES>(fs, cc, true, true); }
            fast_valid = fast_ok && !collide;
        } else {
            // the queue's minimum t is a child of an earlier expansion: rebuild it from its family record, and put
            // that family's next child (the smallest sibling key above t) back in the queue together with kbest
            const uint64_t t = bcast64(heap.top);
            const uint32_t fbase = subkey_idx(t, KB) - subkey_rank(t);
            const FamRec fr = load_fam(pl.fam + fbase);
            const uint64_t hn_t = ringH_get(off + subkey_depth(t));
            const bool fbad = (fr.depth_flags >> 30) & 1u, fhas1 = (fr.depth_flags >> 31) & 1u;
            const uint32_t fdepth = (fr.depth_flags & 0xFFFFFFu) + 1u;
            // the siblings' keys, again one slot per lane group: each lane fetches its slot's cost sum from the record
            const uint32_t ftot = reinterpret_cast<const uint32_t*>(pl.fam + fbase)[16 + lslot];   // FamRec::tot[lslot]
            const uint64_t sk = lane_subkey(lslot, fbad, fhas1, fr.frozen + hn_t, ftot, fr.hets, fbase, fdepth, KB);
            const uint64_t knext = lane_min4(sk > t ? sk : ~0ull);
            heap.replace_push(kbest, knext);
            cur = cur_from_fam(fr, subkey_rank(t), subkey_total(t, KB), subkey_idx(t, KB), off);
            // resume the incremental state: the family's saved prefix scores + the popped child's own cell at the
            // parent's variant (unless that variant has colliding rows: then the plane words rebuild it)
            const uint32_t pF = off + fdepth - 1u;
            const uint32_t fflags = bcast32(reinterpret_cast<const uint32_t*>(hp_smem + LDS_VRING_OFF)[pF & 63u]) >> 28;
            fast_valid = SAVE_STATE && fast_ok && !(fflags & VAR_NOFAST);
            if (fast_valid) {
                const bool two = cx.tiles == 2u;
                const uint32_t* row = cx.ctab + ((size_t)pF << (two ? 7 : 6)) + lane_id();
                const uint32_t cell_a = row[0], cell_b = two ? row[64] : 0u;
                const uint4 q = pl.vec[(size_t)fbase * 64 + lane_id()];
                fs = FastState{q.x, q.y, q.z, q.w};
                const uint32_t slot = fbad ? 0u : (fhas1 ? subkey_rank(t) : (subkey_rank(t) == 0u ? 0u : subkey_rank(t) + 1u));
                const bool a1 = (slot & 1u) != 0, a2 = (slot == 0u) || (slot == 3u);
                CellCost pc;
                pc.x0a = cell_a & 0xFFu; pc.x1a = (cell_a >> 8) & 0xFFu;
                pc.x0b = cell_b & 0xFFu; pc.x1b = (cell_b >> 8) & 0xFFu;
                if (a1) { if (a2) fast_apply<TILES>(fs, pc, true, true); else fast_apply<TILES>(fs, pc, true, false); }
                else { if (a2) fast_apply<TILES>(fs, pc, false, true); else fast_apply<TILES>(fs, pc, false, false); }
            }
        }
        next_idx += kd.n;
        seg_stamp<PROF>(wc, 5);   // [5] record store + heap pushes (+ pop on the slow path)
    }
    // the heap and the chunk pool never write past their capacity (they drop the item and raise a sticky flag), so
    // the flags are looked at once per sub-solve; the host's sizing makes them unreachable anyway
    if (st == ST_OK && (__any(heap.ovf) || pl.ovf)) st = ST_OVERFLOW;
    if (cur.depth == ps) {  // astar_phaser.rs:395-399 (peek, not pop)
        max_cost = max(max_cost, cur.total);
        next_expected += 1;
    }
    est = bcast64(max_cost);
    solved = next_expected - 1;
    wc.sub_pops += visited;        // the work counters advance once per sub-solve, not once per pop
    wc.nodes += next_idx - 1;      // children created (node_index 0 is the root)
    cx.flush();
    return st;
}

// What the main search needs from the heuristic phase of the same block (all wave-uniform).
struct HeurResult {
    int32_t st;
    uint64_t sub_pops, nodes, evals, cells;   // work of the heuristic chain
    uint64_t t_start, t_heur;
};

// calculate_astar_heuristic (astar_phaser.rs:246-292): the chain of N sub-solves, i.e. ~97 % of a block's work.
template <bool SUB_LDS, bool PROF, int TILES>
DEVINL HeurResult heuristic_phase(const BatchDev& B, uint32_t blk, uint32_t slot, WaveCounters& wc) {
    const SolveParams& prm = B.prm;
    const BlockDesc d = B.desc[blk];
    const uint32_t N = d.n_vars;
    const uint32_t lane = lane_id();
    Ctx cx;
    const uint32_t* vlo = B.vlo + d.var_off;
    const uint32_t* vhi = B.vhi + d.var_off;
    const uint8_t* vflags = B.vflags + d.var_off;
    cx.rstart = B.rstart + d.read_off; cx.rend = B.rend + d.read_off; cx.rword = B.rword + d.read_off;
    cx.words = B.words + d.word_off * WORD_DWORDS;
    cx.ctab = (d.cell_off != ~0ull) ? B.ctab + d.cell_off : nullptr;
    cx.n_rows = d.n_reads;
    cx.tiles = d.ctab_shift == 7u ? 2u : 1u;
    cx.N = N; cx.evals = 0; cx.cells = 0; cx.ev32 = 0; cx.cl32 = 0; cx.cring_off = 0;
    uint64_t* H = B.H + d.h_off;
    Pools subp;
    {
        unsigned char* sb = B.sub_pool + (size_t)slot * sub_pool_bytes_per_slot(prm);
        subp.vec = reinterpret_cast<uint4*>(sb + sub_pool_vec_off(prm));
        subp.fam = reinterpret_cast<FamRec*>(sb);
        subp.chunk = reinterpret_cast<ChunkRec*>(sb + (size_t)prm.cap_sub * sizeof(FamRec));
        subp.cap_chunk = prm.cap_chunk_sub; subp.n_chunk = 0; subp.ovf = 0;
    }
    int32_t st = ST_OK;
    const uint64_t t_start = __builtin_readcyclecounter();
    const bool resume = bcast32(lane == 0 ? (uint32_t)B.status[blk] : 0u) == (uint32_t)ST_OVERFLOW_MAIN;

    if (!resume) {
        // ---- calculate_astar_heuristic (astar_phaser.rs:246-292) ---------------------------------------
        SubHeap<SUB_LDS> sub;
        sub.gbase = SUB_LDS ? nullptr : reinterpret_cast<uint64_t*>(B.sub_heap_g) + (size_t)slot * prm.jcap_sub * 64;
        sub.jcap = prm.jcap_sub;
        sub.ovf = 0;
        if (lane == 0) H[N] = 0;
        ringH_set(N, 0);
        uint32_t clip = 1;
        // A block the segment-parallel heuristic ran for and could not accept (a seam stayed open after the long warm-up) is not
        // walked again from end to end: this chain IS the truth, so at the top of every segment it holds what that segment's
        // warm-up had to reproduce - the 40-value look-ahead state and the clip. Where they are identical the segment's owned
        // values are this chain's own (same function of the same inputs) up to the offset: they are taken over, offset applied,
        // and the chain goes on below the segment; only a segment whose seam is open against the TRUE state is computed here.
        // (Round 6. Until then one open seam of a hundred cost the block its whole sequential chain: 123 us a variant at 60x.)
        const uint32_t seg_n = B.blk_seg_n ? B.blk_seg_n[blk] : 0u;
        const uint32_t seg_0 = seg_n ? B.blk_seg_first[blk] : 0u;
        uint32_t seg_k = seg_n;   // segments [0, seg_k) lie below the chain's position
        for (uint32_t v = N; v-- > 0;) {
            if (seg_k > 0 && v + 1 == B.segs[seg_0 + seg_k - 1].b) {
                seg_k -= 1;
                const SegDesc sd = B.segs[seg_0 + seg_k];
                const SegOut& o = B.seg_out[seg_0 + seg_k];
                const uint64_t hb = ringH_get(sd.b), s00 = o.seam[0];
                bool cj = true;
                if (lane >= 1 && lane < SEG_STATE && sd.b + lane <= N)
                    cj = (o.seam[lane] - s00) == (reinterpret_cast<const uint64_t*>(hp_smem + LDS_HRING_OFF)[(sd.b + lane) & 63u] - hb);
                if (o.status == ST_OK && o.clip_at_b == clip && __all(cj) && sd.a < sd.b) {
                    const uint64_t off = hb - s00;
                    for (uint32_t x = sd.a + lane; x < sd.b; x += 64) {
                        const uint64_t hx = H[x] + off;
                        H[x] = hx;
                        if (x - sd.a < 64u) {   // the chain's rings: the 64 variants above its new position
                            reinterpret_cast<uint64_t*>(hp_smem + LDS_HRING_OFF)[x & 63u] = hx;
                            reinterpret_cast<uint32_t*>(hp_smem + LDS_VRING_OFF)[x & 63u] = vlo[x] | ((uint32_t)vflags[x] << 28);
                        }
                    }
                    clip = o.clip_out;
                    wc.sub_pops += o.ctr.sub_pops;
                    wc.nodes += o.ctr.nodes_created;
                    if (lane == 0) { cx.evals += o.ctr.evals; cx.cells += o.ctr.cells; }
                    v = sd.a;   // (the loop's own v-- moves on to sd.a - 1)
                    continue;
                }
            }
            ringH_set(v, 0);  // heuristic_costs[problem_offset] is still 0 (astar_phaser.rs:320)
            uint32_t fl = 0, l = 0, h = 0;
            if (lane == 0) { fl = vflags[v]; l = vlo[v]; }
            fl = bcast32(fl);
            ringV_set(v, l, fl);
            uint64_t est = 0;
            uint32_t solved = 0;
            st = subsolve<SUB_LDS, PROF, TILES>(cx, prm, v, clip, sub, subp, wc, est, solved);
            if (st != ST_OK) break;
            if (solved < min(clip, 2u)) { st = ST_INVARIANT; break; }  // astar_phaser.rs:268
            const bool bad = (fl & HP_VAR_IGNORED) != 0;
            const uint64_t hnext = ringH_get(v + 1);
            uint64_t hv;
            if (bad) hv = hnext;
            else {
                if (est < hnext) { st = ST_INVARIANT; break; }  // astar_phaser.rs:284
                hv = est;
            }
            ringH_set(v, hv);
            if (lane == 0) H[v] = hv;
            clip = min(solved + 1, prm.max_seg);
        }
    } else {
        // the heuristic of this block was completed by an earlier launch whose main-search scratch overflowed
        uint64_t a = 0, b2 = 0, c2 = 0, d2 = 0;
        if (lane == 0) {
            const hp_work_counters c = B.counters[blk];
            a = c.sub_pops; b2 = c.nodes_created; c2 = c.evals; d2 = c.cells;
            cx.evals = c2; cx.cells = d2;
        }
        wc.sub_pops = bcast64(a);
        wc.nodes = bcast64(b2);
    }
    const uint64_t t_heur = __builtin_readcyclecounter();
    // counters of the heuristic phase, kept in case the main search has to be re-run with more scratch
    const uint64_t h_evals = wave_sum_u64(cx.evals), h_cells = wave_sum_u64(cx.cells);
    HeurResult hr;
    hr.st = st; hr.sub_pops = wc.sub_pops; hr.nodes = wc.nodes; hr.evals = h_evals; hr.cells = h_cells;
    hr.t_start = t_start; hr.t_heur = t_heur;
    return hr;
}


// astar_solver's main pruned search + emission (astar_phaser.rs:451-633) for a block whose H[] is complete.
// (Tried as a real, non-inlined call to keep its state out of the sub-solver loop's register allocation: the
// heuristic chain got 3 % faster, the main search 2.5x slower through its stack frame - a net loss.)
template <int TILES>
DEVINL void main_phase(const BatchDev& B, uint32_t blk, uint32_t slot, HeurResult hr) {
    const SolveParams& prm = B.prm;
    const BlockDesc d = B.desc[blk];
    const uint32_t N = d.n_vars;
    const uint32_t lane = lane_id();
    Ctx cx;
    const uint32_t* vlo = B.vlo + d.var_off;
    const uint32_t* vhi = B.vhi + d.var_off;
    const uint8_t* vflags = B.vflags + d.var_off;
    cx.rstart = B.rstart + d.read_off; cx.rend = B.rend + d.read_off; cx.rword = B.rword + d.read_off;
    cx.words = B.words + d.word_off * WORD_DWORDS;
    cx.ctab = (d.cell_off != ~0ull) ? B.ctab + d.cell_off : nullptr;
    cx.n_rows = d.n_reads;
    cx.tiles = d.ctab_shift == 7u ? 2u : 1u;
    cx.N = N; cx.evals = 0; cx.cells = 0; cx.ev32 = 0; cx.cl32 = 0; cx.cring_off = 0;
    uint64_t* H = B.H + d.h_off;
    Pools mainp;
    {
        unsigned char* mb = B.main_pool + (size_t)slot * ((size_t)prm.cap_main * sizeof(FamRec) + (size_t)prm.cap_chunk_main * sizeof(ChunkRec));
        mainp.fam = reinterpret_cast<FamRec*>(mb);
        mainp.chunk = reinterpret_cast<ChunkRec*>(mb + (size_t)prm.cap_main * sizeof(FamRec));
        mainp.cap_chunk = prm.cap_chunk_main; mainp.n_chunk = 0; mainp.ovf = 0;
    }
    uint32_t* tracker = B.tracker + (size_t)slot * ((size_t)prm.max_n_vars + 1);
    WaveCounters wc{hr.sub_pops, 0, hr.nodes, {0, 0, 0, 0, 0, 0}, 0};
    int32_t st = hr.st;
    const uint64_t h_evals = hr.evals, h_cells = hr.cells, h_nodes = hr.nodes, t_start = hr.t_start, t_heur = hr.t_heur;

    // ---- main pruned search (astar_phaser.rs:451-633) ---------------------------------------------------
    hp_phase_stats stats{};
    uint32_t ovf_progress = 0;   // how far the search had come when its scratch ran out: the host sizes the next attempt with it
#if HP_MAIN_PROF
    uint64_t mp_prof0 = 0, mp_prof1 = 0, mp_prof2 = 0;
#if HP_MAIN_PROF == 2
    uint64_t mp_dv[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    const uint64_t mp_rt0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
#endif
    if (st == ST_OK) {
        MainHeap hq;
        hq.base = B.main_heap + (size_t)slot * prm.jcap_main * 64;
        hq.jcap = prm.jcap_main;
        hq.ovf = 0;
        hq.reset();
        // PQueueHapTracker (astar_phaser.rs:171-231): per-length counts live in global scratch and are only
        // ever touched by lane 0 (plain same-thread read-modify-write); the running total is a register.
        for (uint32_t i = lane; i <= N; i += 64) tracker[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        uint32_t trk_total = 0, trk_thr = 0;
        // The histogram lives in HBM (N + 1 counters), but a dive touches it in a fixed pattern: add at a NEW largest
        // length, then remove from that same length at the next pop. One entry is therefore kept in (uniform)
        // registers and written back when another length is touched; a length above everything touched so far is known
        // to be zero and is not loaded. Only a jump back in the search loads a counter (one round trip) - the two
        // read-modify-writes per pop used to stall the wave twice per pop.
        uint32_t tc_len = NONE32, tc_val = 0, tc_top = 0;   // cached length, its count, largest length touched + 1
        auto trk_at = [&](uint32_t len) {
            if (len == tc_len) return;
            if (tc_len != NONE32 && lane == 0) tracker[tc_len] = tc_val;
            if (len >= tc_top) { tc_val = 0; tc_top = len + 1; }
            else {
                uint32_t v = 0;
                if (lane == 0) v = tracker[len];
                tc_val = bcast32(v);
            }
            tc_len = len;
        };
        auto trk_add = [&](uint32_t len, uint32_t n) {
            trk_at(len);
            tc_val += n;
            if (len >= trk_thr) trk_total += n;
        };
        auto trk_remove = [&](uint32_t len) {
            trk_at(len);
            tc_val -= 1u;
            if (len >= trk_thr) trk_total -= 1;
        };

        uint64_t thr = prm.minq_main;                  // curr_queue_size_threshold
        const uint64_t max_q = 10ull * prm.minq_main;  // max_queue_size
        uint32_t min_progress = 0, next_expected = 0;
        uint64_t pruned = 0, next_idx = 1, qlen = 1;
        const uint64_t h0 = bcast64(lane == 0 ? H[0] : 0);
        Cur cur = root_node(h0);
        trk_add(0, 1);
        // incremental scoring as in the sub-solver (see expand_fast): the root has scored nothing yet
        const bool fast_ok = cx.ctab != nullptr;
        bool fast_valid = fast_ok;
        FastState fs{0, 0, 0, 0};
        uint32_t ring_chunk = NONE32;   // which 64-variant chunk the LDS rings hold
#if HP_MAIN_PROF
        uint64_t mp_exp = 0, mp_store = 0, mp_pop = 0, mp_fam = 0, mp_jumps = 0, mp_t = 0;
#define MP_T0() mp_t = __builtin_readcyclecounter()
#define MP_ACC(x) { const uint64_t n_ = __builtin_readcyclecounter(); x += n_ - mp_t; mp_t = n_; }
#else
#define MP_T0()
#define MP_ACC(x)
#endif
#if HP_MAIN_PROF == 2   // a dive step by part (the counters' fields carry raw ticks: head, expand, keys, store, push, prune, cur)
        uint64_t dv[7] = {0, 0, 0, 0, 0, 0, 0}, dv_t = __builtin_readcyclecounter();
#define DV(i) { const uint64_t n_ = __builtin_readcyclecounter(); dv[i] += n_ - dv_t; dv_t = n_; }
#else
#define DV(i)
#endif

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
                if (hq.empty()) { st = ST_INVARIANT; break; }
                const Key t = hq.top;
                MP_T0();
                hq.pop();
                MP_ACC(mp_pop);
                cur = cur_from_fam(load_fam(mainp.fam + (key_idx(t) - key_rank(t))), key_rank(t), t.hi >> 24, key_idx(t), 0);
#if HP_MAIN_PROF
                if (cur.depth == 0xFFFFFFFFu) break;
                MP_ACC(mp_fam);
                mp_jumps += 1;
#endif
                fast_valid = false;
                continue;
            }
            const uint32_t p = cur.depth;
            // per-variant inputs (flags, first candidate row, H[p + 1]) come from the LDS rings of the heuristic phase,
            // refilled 64 variants at a time with coalesced loads: one HBM round trip per 64 pops of a dive instead of
            // one per pop at the head of the loop
            if ((p >> 6) != ring_chunk) {
                const uint32_t cb = p & ~63u;
                const uint32_t x = cb + lane, y = cb + 1u + lane;
                if (x < N) reinterpret_cast<uint32_t*>(hp_smem + LDS_VRING_OFF)[x & 63u] = vlo[x] | ((uint32_t)vflags[x] << 28);
                if (y <= N) reinterpret_cast<uint64_t*>(hp_smem + LDS_HRING_OFF)[y & 63u] = H[y];
                ring_chunk = p >> 6;
            }
            const uint32_t rv = bcast32(reinterpret_cast<const uint32_t*>(hp_smem + LDS_VRING_OFF)[p & 63u]);
            const uint64_t hn = bcast64(reinterpret_cast<const uint64_t*>(hp_smem + LDS_HRING_OFF)[(p + 1u) & 63u]);
            const uint32_t fl = rv >> 28, l = rv & 0x0FFFFFFFu;
            Kids kd;
            CellCost cc;
            const bool collide = (fl & VAR_NOFAST) != 0;
            DV(0);
            MP_T0();
            if (fast_valid && !collide) expand_fast<false, TILES>(cx, cur, 0, p, (fl & HP_VAR_IGNORED) != 0, hn, mainp, kd, wc, fs, cc);
            else expand<false, TILES>(cx, cur, 0, p, l, (fl & HP_VAR_IGNORED) != 0, hn, mainp, kd, wc, fs, cc);
            cx.flush();
            MP_ACC(mp_exp);
            DV(1);
            wc.nodes += kd.n;
            if (next_idx + kd.n > prm.cap_main) { st = ST_OVERFLOW_MAIN; ovf_progress = next_expected; break; }
            if (kd.bad && kid_total<0>(kd) != cur.total) { st = ST_INVARIANT; break; }  // astar_phaser.rs:529
            const Key k0 = make_key(kid_total<0>(kd), kid_hets<0>(kd), next_idx + kid_rank<0>(kd), kid_rank<0>(kd), kd.depth);
            const Key k1 = kid_valid<1>(kd) ? make_key(kid_total<1>(kd), kid_hets<1>(kd), next_idx + kid_rank<1>(kd), kid_rank<1>(kd), kd.depth) : key_inf();
            const Key k2 = kid_valid<2>(kd) ? make_key(kid_total<2>(kd), kid_hets<2>(kd), next_idx + kid_rank<2>(kd), kid_rank<2>(kd), kd.depth) : key_inf();
            const Key k3 = kid_valid<3>(kd) ? make_key(kid_total<3>(kd), kid_hets<3>(kd), next_idx + kid_rank<3>(kd), kid_rank<3>(kd), kd.depth) : key_inf();
            int best = 0;
            Key kbest = k0;
            if (key_less(k1, kbest)) { kbest = k1; best = 1; }
            if (key_less(k2, kbest)) { kbest = k2; best = 2; }
            if (key_less(k3, kbest)) { kbest = k3; best = 3; }
            // push every child except the best one, which is held in registers (it is logically queued)
#if HP_MAIN_PROF == 2
            if (kbest.hi == 0x123456789ull) break;   // (never: the keys are a dependency of the stamp)
#endif
            DV(2);
            trk_add(kd.depth, kd.n);
            fam_store(mainp.fam, kd, cur, (uint32_t)next_idx);
            DV(3);
            {   // the (at most three) children other than `best`
                const Key q0 = best == 0 ? k1 : k0;
                const Key q1 = best <= 1 ? k2 : k1;
                const Key q2 = best <= 2 ? k3 : k2;
                hq.push3(q0 /* never the infinite key unless a lone child */, q1, q2);
            }
            MP_ACC(mp_store);
            DV(4);
            qlen += kd.n;
            // astar_phaser.rs:564-585
            while (trk_total > thr && min_progress < next_expected) {
                min_progress += 1;
                {   // increase_threshold(min_progress)
                    uint32_t c = 0;
                    if (min_progress - 1 == tc_len) c = tc_val;
                    else { if (lane == 0) c = tracker[min_progress - 1]; c = bcast32(c); }
                    trk_total -= c;
                    trk_thr = min_progress;
                }
                if (qlen > max_q) {
                    // full prune: every queued node shorter than min_progress gets the cleared priority
                    // (cost 0, same hets, same index)
                    hq.clear_below(min_progress);
                    if (kd.depth < min_progress) kbest.hi &= 0xFFFFFFull;
                }
            }
            DV(5);
            if (key_less(kbest, hq.top)) {
                if (best == 0) { cur = kid_as_cur<0>(kd, next_idx); fast_apply<TILES>(fs, cc, false, true); }
                else if (best == 1) { cur = kid_as_cur<1>(kd, next_idx); fast_apply<TILES>(fs, cc, true, false); }
                else if (best == 2) { cur = kid_as_cur<2>(kd, next_idx); fast_apply<TILES>(fs, cc, false, false); }
                else { cur = kid_as_cur<3>(kd, next_idx); fast_apply<TILES>(fs, cc, true, true); }
                cur.total = kbest.hi >> 24;
                fast_valid = fast_ok && !collide;
            } else {
                fast_valid = false;
                MP_T0();
                hq.push(kbest);
                const Key t = hq.top;
                hq.pop();
                MP_ACC(mp_pop);
                cur = cur_from_fam(load_fam(mainp.fam + (key_idx(t) - key_rank(t))), key_rank(t), t.hi >> 24, key_idx(t), 0);
#if HP_MAIN_PROF
                if (cur.depth == 0xFFFFFFFFu) break;   // (never: makes the record's words a dependency of the stamp below)
                MP_ACC(mp_fam);
                mp_jumps += 1;
#endif
            }
            next_idx += kd.n;
#if HP_MAIN_PROF == 2
            if (cur.frozen == 0x123456789abcull && cur.anc1 == 77u) break;   // (never)
#endif
            DV(6);
            if (__any(hq.ovf) || mainp.ovf) { st = ST_OVERFLOW_MAIN; ovf_progress = next_expected; break; }
        }

#if HP_MAIN_PROF
        mp_prof0 = (mp_exp >> 10) | ((mp_store >> 10) << 32);   // kilo-ticks: expansion | record store + pushes
        mp_prof1 = mp_jumps;
        mp_prof2 = (mp_pop >> 10) | ((mp_fam >> 10) << 32);     // kilo-ticks: push + pop of a jump | the popped node's family record
#endif
#if HP_MAIN_PROF == 2
        for (int i = 0; i < 7; ++i) mp_dv[i] = dv[i];
#endif
        if (st == ST_OK) {
            // ---- emit the solution (astar_phaser.rs:588-628): walk the window chain from the last chunk down
            uint8_t* o1 = B.h1 + d.var_off;
            uint8_t* o2 = B.h2 + d.var_off;
            uint64_t phased = 0, snvs = 0, skipped = 0;
            uint32_t chunk = (N - 1) >> 5;
            Win w = cur.w0, wn = cur.w1;
            uint32_t slot_next = cur.anc2;
            bool have_wn = true;
            for (;;) {
                const uint32_t pos = chunk * 32 + (lane & 31u);
                const bool inb = lane < 32 && pos < N;
                const uint32_t bit = lane & 31u;
                const uint32_t nvb = (w.nv >> bit) & 1u, b1 = (w.h1 >> bit) & 1u, b2 = (w.h2 >> bit) & 1u;
                if (lane == 0) B.hapw[d.chunk_off + chunk] = w;
                if (inb) {
                    o1[pos] = nvb ? 2 : (uint8_t)b1;
                    o2[pos] = nvb ? 2 : (uint8_t)b2;
                }
                const uint64_t m_het = __ballot(inb && !nvb && b1 != b2);
                const uint64_t m_skip = __ballot(inb && nvb);
                const uint64_t m_snv = __ballot(inb && (vflags[inb ? pos : 0] & HP_VAR_SNV));
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
                    const ChunkRec a = load_chunk(mainp.chunk + slot_next);
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

    const uint64_t m_evals = wave_sum_u64(cx.evals), m_cells = wave_sum_u64(cx.cells);
    if (lane == 0) {
        B.stats[blk] = stats;
        hp_work_counters c{};
        c.sub_pops = wc.sub_pops;
        if (st == ST_OVERFLOW_MAIN) {  // keep only the heuristic phase; the main search will be redone
            c.evals = h_evals; c.cells = h_cells; c.nodes_created = h_nodes;
            c.reserved[2] = ovf_progress;   // (the attempt that succeeds writes its own counters: reserved[] is zero there)
        } else {
            c.main_pops = wc.main_pops; c.evals = h_evals + m_evals; c.cells = h_cells + m_cells; c.nodes_created = wc.nodes;
        }
        c.reserved[0] = t_heur - t_start;                      // shader-clock cycles spent in the heuristic chain
        c.reserved[1] = __builtin_readcyclecounter() - t_heur;  // ... in the main search + emit
#if HP_MAIN_PROF
        c.reserved[2] = mp_prof2; c.sub_pops = mp_prof1; c.reserved[0] = mp_prof0;
        c.cells = __builtin_amdgcn_s_memrealtime() - mp_rt0;
#endif
#if HP_MAIN_PROF == 2
        c.sub_pops = mp_dv[0]; c.evals = mp_dv[1]; c.cells = mp_dv[2]; c.nodes_created = mp_dv[3]; c.reserved[0] = mp_dv[4]; c.reserved[1] = mp_dv[5]; c.reserved[2] = mp_dv[6];
#endif
        B.counters[blk] = c;
        B.status[blk] = st;
    }
}

template <bool SUB_LDS, bool PROF, int TILES>
DEVINL void solve_block(const BatchDev& B, uint32_t blk, uint32_t slot) {
    WaveCounters wc{0, 0, 0, {0, 0, 0, 0, 0, 0}, 0};
    const HeurResult hr = heuristic_phase<SUB_LDS, PROF, TILES>(B, blk, slot, wc);
    main_phase<TILES>(B, blk, slot, hr);
    if (PROF && lane_id() == 0) {  // segment profile: pack 6 x 32-bit kilo-cycle counters over the cycle fields
        hp_work_counters* c = B.counters + blk;
        c->reserved[0] = (wc.seg[0] >> 10) | ((wc.seg[1] >> 10) << 32);
        c->reserved[1] = (wc.seg[2] >> 10) | ((wc.seg[3] >> 10) << 32);
        c->reserved[2] = (wc.seg[4] >> 10) | ((wc.seg[5] >> 10) << 32);
    }
}


// ---- segment-parallel heuristic: one wavefront per segment ------------------------------------------------------
struct SegBatchDev {
    BatchDev B;
    const SegDesc* segs;
    const uint32_t* seg_order;
    uint32_t n_segs;
    SegOut* out;
    const uint8_t* run_flag;   // second round: only the segments below a seam the first round's warm-up did not close
    uint32_t warm;             // second round: warm-up length that replaces SegDesc::v0 (0 = first round)
    uint32_t cring_off;        // CR: where the staged cell-table window lies in LDS (behind the sub-solver's heap)
};

template <bool SUB_LDS, int TILES, bool CR>
DEVINL void solve_segment(const SegBatchDev& S, uint32_t seg, uint32_t slot) {
    if (S.run_flag && !S.run_flag[seg]) return;
    const BatchDev& B = S.B;
    const SolveParams& prm = B.prm;
    SegDesc sd = S.segs[seg];
    if (S.warm) sd.v0 = min(B.desc[sd.blk].n_vars, sd.b + S.warm);   // (the top segment ends at N: it has no warm-up)
    const BlockDesc d = B.desc[sd.blk];
    const uint32_t N = d.n_vars;
    const uint32_t lane = lane_id();
    Ctx cx;
    const uint32_t* vlo = B.vlo + d.var_off;
    const uint32_t* vhi = B.vhi + d.var_off;
    const uint8_t* vflags = B.vflags + d.var_off;
    cx.rstart = B.rstart + d.read_off; cx.rend = B.rend + d.read_off; cx.rword = B.rword + d.read_off;
    cx.words = B.words + d.word_off * WORD_DWORDS;
    cx.ctab = (d.cell_off != ~0ull) ? B.ctab + d.cell_off : nullptr;
    cx.n_rows = d.n_reads;
    cx.tiles = d.ctab_shift == 7u ? 2u : 1u;
    cx.N = N; cx.evals = 0; cx.cells = 0; cx.ev32 = 0; cx.cl32 = 0; cx.cring_off = S.cring_off;
    uint64_t* H = B.H + d.h_off;
    Pools subp;
    {
        unsigned char* sb = B.sub_pool + (size_t)slot * sub_pool_bytes_per_slot(prm);
        subp.vec = reinterpret_cast<uint4*>(sb + sub_pool_vec_off(prm));
        subp.fam = reinterpret_cast<FamRec*>(sb);
        subp.chunk = reinterpret_cast<ChunkRec*>(sb + (size_t)prm.cap_sub * sizeof(FamRec));
        subp.cap_chunk = prm.cap_chunk_sub; subp.n_chunk = 0; subp.ovf = 0;
    }
    WaveCounters wc{0, 0, 0, {0, 0, 0, 0, 0, 0}, 0};
    SubHeap<SUB_LDS> sub;
    sub.gbase = SUB_LDS ? nullptr : reinterpret_cast<uint64_t*>(B.sub_heap_g) + (size_t)slot * prm.jcap_sub * 64;
    sub.jcap = prm.jcap_sub;
    sub.ovf = 0;
    SegOut* out = S.out + seg;
    int32_t st = ST_OK;
    if (sd.v0 == N && lane == 0) H[N] = 0;
    ringH_set(sd.v0, 0);   // cold start (for the top segment this IS heuristic_costs[N] = 0)
    if (lane == 0 && sd.v0 - sd.b < SEG_STATE) out->seam[sd.v0 - sd.b] = 0;   // (a warm-up that starts at the block's end, fewer than 40 variants up)
    uint32_t clip = 1, clip_at_b = 1;
    for (uint32_t v = sd.v0; v-- > sd.a;) {
        if (v + 1 == sd.b) {   // entering the owned range: only owned work is counted (== the sequential chain's work)
            wc.sub_pops = 0; wc.nodes = 0; cx.evals = 0; cx.cells = 0;
            clip_at_b = clip;
        }
        ringH_set(v, 0);
        uint32_t fl = 0, l = 0, h = 0;
        if (lane == 0) { fl = vflags[v]; l = vlo[v]; }
        fl = bcast32(fl);
        ringV_set(v, l, fl);
        // (CR: the row of the variant that enters the window - one coalesced read per step of the chain; rows leave the ring by
        // being overwritten 64 steps later, the window is 40 deep)
        if (CR && TILES == 1 && cx.ctab != nullptr) reinterpret_cast<uint32_t*>(hp_smem + cx.cring_off)[((v & 63u) << 6) + lane] = cx.ctab[((size_t)v << 6) + lane];
        uint64_t est = 0;
        uint32_t solved = 0;
        st = subsolve<SUB_LDS, false, TILES, CR>(cx, prm, v, clip, sub, subp, wc, est, solved);
        if (st != ST_OK) break;
        if (solved < min(clip, 2u)) { st = ST_INVARIANT; break; }
        const bool bad = (fl & HP_VAR_IGNORED) != 0;
        const uint64_t hnext = ringH_get(v + 1);
        uint64_t hv;
        if (bad) hv = hnext;
        else {
            if (est < hnext) { st = ST_INVARIANT; break; }
            hv = est;
        }
        ringH_set(v, hv);
        if (lane == 0) {
            if (v < sd.b) H[v] = hv;                                    // owned: relative to this segment's cold start
            else if (v - sd.b < SEG_STATE) out->seam[v - sd.b] = hv;    // warm-up values the seam check compares
        }
        clip = min(solved + 1, prm.max_seg);
    }
    // a warm-up failure (e.g. an invariant that only holds with true look-ahead) is not an error of the block:
    // the seam check fails and the block takes the sequential path, which reports real invariant violations.
    const uint64_t evals = wave_sum_u64(cx.evals), cells = wave_sum_u64(cx.cells);
    if (lane == 0) {
        out->clip_at_b = clip_at_b;
        out->clip_out = clip;
        out->status = st;
        hp_work_counters c{};
        c.sub_pops = wc.sub_pops; c.evals = evals; c.cells = cells; c.nodes_created = wc.nodes;
        out->ctr = c;
    }
}

// OCC = waves per SIMD the register allocation targets. 6 (80 VGPRs, a few spills) measured equal or faster than
// the spill-free 4 on every workload tried (throughput batches, a single block, the heavy-tailed mix).
template <bool SUB_LDS, int OCC, int TILES, bool CR>
__global__ void __launch_bounds__(64, OCC) hp_heur_seg_kernel(SegBatchDev S) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t slot = blockIdx.x, G = gridDim.x;
    for (uint32_t round = 0;; ++round) {
        const uint32_t base = round * G;
        if (base >= S.n_segs) break;
        const uint32_t i = base + ((round & 1u) ? (G - 1u - slot) : slot);
        if (i < S.n_segs) solve_segment<SUB_LDS, TILES, CR>(S, S.seg_order[i], slot);
    }
}
template __global__ void hp_heur_seg_kernel<true, 6, 1, false>(SegBatchDev);
template __global__ void hp_heur_seg_kernel<true, 6, 1, true>(SegBatchDev);
template __global__ void hp_heur_seg_kernel<true, 6, 2, false>(SegBatchDev);

// Seam verification + offsets, one thread per segmented block (segments of a block are consecutive, bottom first).
struct StitchDev {
    const BlockDesc* desc;
    const SegDesc* segs;
    SegOut* out;
    const uint32_t* blk_first_seg;   // per segmented block
    const uint32_t* blk_n_seg;
    const uint32_t* blk_id;
    uint32_t n_seg_blocks;
    uint64_t* H;
    uint64_t* seg_offset;            // per segment: what to add to its owned H values
    int32_t* status;
    hp_work_counters* counters;
    uint8_t* retry;                  // first of two rounds: flags the segment below every seam that did not close
    uint32_t final_round;            // 1: a seam that does not close sends the block down the sequential path
};
// One wavefront per segmented block: the seams are walked from the top (each offset builds on the one above), the 40
// look-ahead values of a seam are compared by 40 lanes at once.
__global__ void __launch_bounds__(64) hp_heur_stitch_kernel(StitchDev T) {
    const uint32_t t = blockIdx.x, lane = threadIdx.x;
    if (t >= T.n_seg_blocks) return;
    const uint32_t blk = T.blk_id[t], s0 = T.blk_first_seg[t], ns = T.blk_n_seg[t];
    if (T.status[blk] == ST_H_READY) return;   // accepted by the first round
    const BlockDesc d = T.desc[blk];
    const uint64_t* H = T.H + d.h_off;
    bool ok = T.out[s0 + ns - 1].status == ST_OK;
    hp_work_counters tot = T.out[s0 + ns - 1].ctr;
    if (lane == 0) T.seg_offset[s0 + ns - 1] = 0;
    // offsets of the segment above the seam and of the one above that: a segment is at least 32 variants long, so the 40
    // look-ahead values of a seam lie in those two. In the first of two rounds every segment below an open seam is solved
    // again with the long warm-up (its seam cannot be judged against values that are themselves in doubt).
    uint64_t off0 = 0, off1 = 0;
    for (uint32_t k = ns - 1; (ok || !T.final_round) && k-- > 0;) {
        const uint32_t seg = s0 + k, above = seg + 1;
        if (!ok) { if (lane == 0) T.retry[seg] = 1; continue; }
        const SegOut& o = T.out[seg];
        const uint32_t b = T.segs[seg].b, b_above = T.segs[above].b;
        const uint64_t hb = H[b], s00 = o.seam[0];
        bool cj = true;
        if (lane >= 1 && lane < SEG_STATE && b + lane <= d.n_vars) {
            const uint32_t x = b + lane;
            cj = (o.seam[lane] - s00) == ((H[x] + (x < b_above ? off0 : off1)) - (hb + off0));   // identical look-ahead state (differences)
        }
        const bool closed = o.status == ST_OK && o.clip_at_b == T.out[above].clip_out && __all(cj);
        if (!closed) {
            ok = false;
            if (!T.final_round && lane == 0) T.retry[seg] = 1;
            continue;
        }
        const uint64_t off = off0 + hb - s00;
        if (lane == 0) T.seg_offset[seg] = off;
        off1 = off0; off0 = off;
        tot.sub_pops += o.ctr.sub_pops; tot.evals += o.ctr.evals; tot.cells += o.ctr.cells; tot.nodes_created += o.ctr.nodes_created;
    }
    if (ok) {
        if (lane == 0) { T.counters[blk] = tot; T.status[blk] = ST_H_READY; }
    } else {
        for (uint32_t k = lane; k < ns; k += 64) T.seg_offset[s0 + k] = 0;   // final round: the block falls back to the sequential chain
    }
}
struct ApplyDev {
    const SegDesc* segs;
    const uint64_t* seg_offset;
    const BlockDesc* desc;
    uint32_t n_segs;
    uint64_t* H;
};
__global__ void __launch_bounds__(256) hp_heur_apply_kernel(ApplyDev A) {
    const uint32_t seg = blockIdx.x;
    if (seg >= A.n_segs) return;
    const uint64_t off = A.seg_offset[seg];
    if (off == 0) return;
    const SegDesc sd = A.segs[seg];
    uint64_t* H = A.H + A.desc[sd.blk].h_off;
    for (uint32_t v = sd.a + threadIdx.x; v < sd.b; v += blockDim.x) H[v] += off;
}

// ---- post-processing on the resident matrix (reference src/phaser.rs:350-388 and :714-750) ---------------------
// Kernel A, one thread per packed row: scores the row against both solved haplotypes (haplotag_reads), finds the
// first het it resolves, and the juncture range [js, je) it supports after trimming solution-homozygous ends.
struct PostDev {
    const BlockDesc* desc;
    const uint32_t* row_block;   // packed row -> block
    const uint32_t *rstart, *rend, *rword;
    const uint32_t* words;
    const uint32_t *vlo, *vhi;
    const Win* hapw;
    uint32_t n_rows_total;
    uint64_t n_junctures_total;
    uint8_t* haplotag;     // per packed row: 0 / 1 / 2 (untagged)
    uint32_t* first_het;   // per packed row: block-local index of the first resolved het (NONE32 when untagged)
    uint32_t *js, *je;     // per packed row
    const uint32_t* junc_block;  // juncture -> block
    const uint64_t* junc_off;    // per block: first juncture index
    uint64_t* span_counts;       // per juncture
};

__global__ void __launch_bounds__(256) hp_post_rows_kernel(PostDev P) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P.n_rows_total) return;
    const BlockDesc d = P.desc[P.row_block[r]];
    const uint32_t rs = P.rstart[r], re = P.rend[r];
    const uint32_t* wbase = P.words + (d.word_off + P.rword[r]) * (size_t)WORD_DWORDS;
    const Win* hw = P.hapw + d.chunk_off;
    uint32_t s1 = 0, s2 = 0, fh = NONE32, firstD = NONE32, lastD = NONE32;
    const uint32_t k0 = rs >> 5, k1 = (re - 1) >> 5;
    for (uint32_t k = k0; k <= k1; ++k) {
        const uint4* pw = reinterpret_cast<const uint4*>(wbase + (size_t)(k - k0) * WORD_DWORDS);
        const uint4 x0 = pw[0], x1 = pw[1], x2 = pw[2];
        const Win w = hw[k];
        uint32_t rmask = 0xFFFFFFFFu;
        if (k == k0) rmask &= 0xFFFFFFFFu << (rs & 31u);
        if (k == k1 && (re & 31u)) rmask &= 0xFFFFFFFFu >> (32u - (re & 31u));
        const uint32_t M1 = ~w.nv & ((x0.x ^ w.h1) | x0.y) & rmask;
        const uint32_t M2 = ~w.nv & ((x0.x ^ w.h2) | x0.y) & rmask;
        s1 += wpop(M1, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
        s2 += wpop(M2, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y);
        const uint32_t D = ~w.nv & (w.h1 ^ w.h2) & rmask;   // solution is heterozygous here
        if (D) {
            if (firstD == NONE32) firstD = k * 32 + (uint32_t)__builtin_ctz(D);
            lastD = k * 32 + 31u - (uint32_t)__builtin_clz(D);
            const uint32_t R = D & ~x0.y;                    // ... and the row's allele is set (0/1)
            if (R && fh == NONE32) fh = k * 32 + (uint32_t)__builtin_ctz(R);
        }
    }
    const uint8_t tag = s1 < s2 ? 0 : (s1 > s2 ? 1 : 2);
    P.haplotag[r] = tag;
    P.first_het[r] = tag == 2 ? NONE32 : fh;
    // get_solution_span_counts: junctures [rs, re-1); skip homozygous positions at both ends
    uint32_t js = (firstD != NONE32 && firstD < re - 1) ? firstD : re - 1;
    uint32_t je = (lastD != NONE32 && lastD > js) ? lastD : js;
    P.js[r] = js;
    P.je[r] = je;
}

// Kernel B, one thread per juncture: number of rows whose trimmed range covers it (no atomics).
__global__ void __launch_bounds__(256) hp_post_spans_kernel(PostDev P) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P.n_junctures_total) return;
    const uint32_t blk = P.junc_block[g];
    const BlockDesc d = P.desc[blk];
    const uint32_t j = (uint32_t)(g - P.junc_off[blk]);
    const uint32_t lo = P.vlo[d.var_off + j], hi = P.vhi[d.var_off + j];
    uint64_t c = 0;
    for (uint32_t r = lo; r < hi; ++r) {
        const uint64_t gr = d.read_off + r;
        c += (P.js[gr] <= j && j < P.je[gr]) ? 1u : 0u;
    }
    P.span_counts[g] = c;
}

// Builds the bit-sliced plane words (hp_astar_dev.h) of every row from the caller's cells (2-bit alleles, u8
// qualities, row-major): one thread per row, one 48-byte word per 32 variants. Cells outside the row's region are
// NoOverlap (allele 3), quality 0.
struct PackDev {
    const BlockDesc* desc;
    const PackRaw* raw;
    const uint32_t* row_block;
    const uint32_t *rstart, *rend, *rword;
    const uint64_t* rcell;
    const uint8_t *alleles, *quals;
    uint32_t* words;
    uint64_t n_rows;
};
__global__ void __launch_bounds__(256) hp_pack_words_kernel(PackDev P) {
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= P.n_rows) return;
    const uint32_t blk = P.row_block[row];
    const BlockDesc d = P.desc[blk];
    const PackRaw raw = P.raw[blk];
    const uint32_t rs = P.rstart[row], re = P.rend[row];
    const uint64_t cell0 = P.rcell[row];
    const uint8_t* al = P.alleles + raw.allele_off;
    const uint8_t* ql = P.quals + raw.qual_off;
    uint32_t* w = P.words + ((size_t)d.word_off + P.rword[row]) * WORD_DWORDS;
    for (uint32_t k = rs >> 5; k <= (re - 1) >> 5; ++k, w += WORD_DWORDS) {
        uint32_t a0 = 0xFFFFFFFFu, a1 = 0xFFFFFFFFu, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t p0 = max(rs, k << 5), p1 = min(re, (k + 1) << 5);
        for (uint32_t p = p0; p < p1; ++p) {
            const uint64_t cell = cell0 + (p - rs);
            const uint32_t a = (al[cell >> 2] >> (2u * ((uint32_t)cell & 3u))) & 3u, qv = ql[cell];
            const uint32_t b = p & 31u;
            a0 &= ~((~a & 1u) << b);
            a1 &= ~(((~a >> 1) & 1u) << b);
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] |= ((qv >> j) & 1u) << b;
        }
        reinterpret_cast<uint4*>(w)[0] = make_uint4(a0, a1, q[0], q[1]);
        reinterpret_cast<uint4*>(w)[1] = make_uint4(q[2], q[3], q[4], q[5]);
        reinterpret_cast<uint4*>(w)[2] = make_uint4(q[6], q[7], 0u, 0u);
    }
}

// Fills the per-position cell tables (hp_astar_dev.h CELL_*) from the packed rows: one wavefront per row, lanes over
// the row's cells; the table was zeroed (= no valid entry) beforehand.
struct CtabDev {
    const BlockDesc* desc;
    const uint32_t* row_block;            // packed row -> block
    const uint32_t *rstart, *rend, *rword;
    const uint32_t* words;
    const uint8_t* vflags;
    uint32_t* ctab;
    uint64_t n_rows;
};
__global__ void __launch_bounds__(256) hp_build_ctab_kernel(CtabDev T) {
    const uint64_t row = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= T.n_rows) return;
    const BlockDesc d = T.desc[T.row_block[row]];
    if (d.cell_off == ~0ull) return;
    const uint32_t rs = T.rstart[row], re = T.rend[row];
    const uint32_t sh = d.ctab_shift;   // entries per variant: 64 or 128
    const uint32_t entry = (uint32_t)(row - d.read_off) & ((1u << sh) - 1u);
    const uint32_t* w0 = T.words + ((size_t)d.word_off + T.rword[row]) * WORD_DWORDS;
    uint32_t* tab = T.ctab + d.cell_off;
    for (uint32_t p = rs + (threadIdx.x & 63u); p < re; p += 64) {
        const uint32_t* w = w0 + (size_t)((p >> 5) - (rs >> 5)) * WORD_DWORDS;
        const uint32_t b = p & 31u;
        const uint32_t a = ((w[0] >> b) & 1u) | (((w[1] >> b) & 1u) << 1);
        uint32_t q = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) q |= ((w[2 + k] >> b) & 1u) << k;
        if (T.vflags[d.var_off + p] & HP_VAR_IGNORED) q = 0;
        const uint32_t x0 = a != 0u ? q : 0u, x1 = a != 1u ? q : 0u;
        tab[((size_t)p << sh) + entry] = x0 | (x1 << 8) | (p + 1 == re ? CELL_ENDS : 0u) | CELL_VALID |
                                         (min(p - rs, CELL_T_MAX) << CELL_T_SHIFT);
    }
}

// TILES = 64-row tiles of the per-position cell table the launch supports (2 costs two more VGPRs per lane and is
// only used when some block of the launch has more than 64 candidate rows per variant)
template <bool SUB_LDS, int OCC, bool PROF, int TILES>
__global__ void __launch_bounds__(64, OCC) hp_astar_kernel(BatchDev B) {
    // A search wavefront is one long dependent chain; when it shares a SIMD with throughput kernels of another stream
    // (the next chunk's graph-WFA, hp_block.hip) it should win the issue arbitration: those have slack, it has none.
    __builtin_amdgcn_s_setprio(3);
    const uint32_t slot = blockIdx.x;
    const uint32_t G = gridDim.x;
    // Static "snake" assignment over the LPT-sorted work list: workgroup w takes ranks w, 2G-1-w, 2G+w, ...
    // (uniform control flow, no atomics; every workgroup gets a similar mix of large and small blocks).
    for (uint32_t round = 0;; ++round) {
        const uint32_t base = round * G;
        if (base >= B.n_items) break;
        const uint32_t i = base + ((round & 1u) ? (G - 1u - slot) : slot);
        if (i < B.n_items) solve_block<SUB_LDS, PROF, TILES>(B, B.order[i], slot);
    }
}

template __global__ void hp_astar_kernel<true, 6, false, 1>(BatchDev);
template __global__ void hp_astar_kernel<true, 6, false, 2>(BatchDev);
template __global__ void hp_astar_kernel<true, 6, true, 1>(BatchDev);    // HP_SEG_PROFILE=1
template __global__ void hp_astar_kernel<false, 4, false, 2>(BatchDev);

}  // namespace hp
