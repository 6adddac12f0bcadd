// hp_wfa2_dev.h — device-resident graph-WFA stage, second generation (round 2).
//
// Data layout in HBM for one batch of BAM records ("jobs", reference src/read_parsing.rs:738-780):
//   seq[]      one byte buffer: [merged reference ranges][allele pool][read bases][pad]
//   vars[]     W2Variant: every distinct hp_wfa_variant of the batch ONCE (the reads of a block pass slices of the
//              block's variant vectors; the host merges their address ranges like it merges the reference windows)
//   jobs[]     W2Job: window + variant index ranges + read + where the job's graph lives
//   gnodes[]   W2Node (16 B): the job's graph, written by w2_build (one thread per job, hp_wfa2_build_kernel)
//   gedges[]   u16 children lists (creation order is irrelevant to the results: injections are set unions)
//   gtags[]    node -> (het index, allele) table (wfa_graph.rs:19 NodeAlleleMap), in node order
// w2_build restates WFAGraph::from_reference_variants_with_hom (wfa_graph.rs:119-284) over sequence SPANS; it is plain
// C++ that compiles for the device (hipcc) and for the host (the CPU model in tests/cpp/wfa2_model.cpp, which pins
// this builder and the compact wavefront formulation of hp_wfa2_kernel.hip against the oracle without a GPU).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HP_HD __host__ __device__ inline
#else
#define HP_HD inline
#endif

namespace hp {

struct W2Variant {          // 32 B
    int64_t  position;      // Variant::position()
    uint32_t ref_len;       // get_ref_len()
    uint32_t flags;         // bit0 ignored, bit1 allele0 is itself an ALT (index_allele0 != 0)
    uint32_t a0_off, a0_len;   // truncated allele0 in the allele pool (only read when flags & 2)
    uint32_t a1_off, a1_len;   // truncated allele1
};

struct W2Job {              // 80 B
    uint64_t ref_off;       // byte offset in seq[] of reference base `ref_start`
    uint64_t read_off;      // byte offset in seq[] of read[0]
    int64_t  ref_start;     // chromosome coordinate of the window start (min_position)
    uint32_t ref_len;       // ref_end - ref_start
    uint32_t read_len;
    uint32_t het_first, n_hets;   // into vars[]
    uint32_t hom_first, n_homs;
    uint32_t node_off, node_cap;  // into gnodes / per-node scratch
    uint32_t edge_off, edge_cap;  // into gedges (u16 units: two per overflow entry)
    uint32_t tag_off, tag_cap;    // into gtags
    uint32_t allele_off;          // into the output allele pool (n_hets bytes)
    uint32_t group;               // caller-defined (block-level path: qname group); unused by the WFA stage
};

struct W2Node {             // 16 B: one aligned 16-byte load per node visit
    uint32_t seq_off;       // reference node: offset inside the job's window; allele node: offset in the allele pool
    uint32_t len_ref;       // length | is_reference << 31
    uint32_t child;         // n_children | first overflow entry << 16
    uint32_t c01;           // the first two children (child 0 | child 1 << 16): most nodes have at most two. A third and later
                            // child is an entry (parent, child) of the job's overflow list edges[], in creation order = ascending
                            // child id; a node's entries are found by scanning on from its first one (w2_next_child)
};
// child j >= 2 of node n: the next overflow entry of n at or after `scan` (which moves past it)
HP_HD uint32_t w2_next_child(const uint16_t* edges, uint32_t n, uint32_t& scan) {
    while ((uint32_t)edges[2u * scan] != n && scan < 32767u) ++scan;
    const uint32_t c = edges[2u * scan + 1u];
    ++scan;
    return c;
}
constexpr uint32_t W2_IS_REF = 0x80000000u;

struct W2Info {             // builder output per job
    uint32_t n_nodes, n_edges, n_tags;
    int32_t  status;        // W2B_*
};
constexpr int32_t W2B_OK = 0;
constexpr int32_t W2B_NEED_HOST = 1;   // outside the device builder's small fixed queues / 16-bit ids: the host path builds it
constexpr int32_t W2B_INVARIANT = -3;  // an assert! of wfa_graph.rs:170,257,276,281 would have fired

// tag word: node | het index << 16 | allele << 31
HP_HD uint32_t w2_tag(uint32_t node, uint32_t vi, uint32_t allele) { return node | (vi << 16) | (allele << 31); }

constexpr int W2B_MAXQ = 24;    // alt nodes waiting to reconnect
constexpr int W2B_MAXRR = 32;   // reference_reconnect
constexpr int W2B_MAXRA = 32;   // reference alleles waiting for the next reference node

// WFAGraph::from_reference_variants_with_hom (wfa_graph.rs:119-284) for one job, in one pass: a new node is entered into
// its parents' child lists as it is created (the parents are the handful of nodes of reference_reconnect). Node ids are
// creation order (wfa_graph.rs:298-331).
HP_HD void w2_build(const W2Job& J, const W2Variant* vars, W2Node* nodes, uint16_t* edges, uint32_t* tags, W2Info* info) {
    uint32_t nn = 0, ne = 0, nt = 0, no = 0;   // nodes, edges, tags, overflow entries
    int32_t status = W2B_OK;
    const int64_t ref_start = J.ref_start, ref_end = J.ref_start + (int64_t)J.ref_len;
    int64_t previous_end = ref_start;
    uint32_t rr[W2B_MAXRR]; int nrr = 0;      // reference_reconnect
    uint32_t ra[W2B_MAXRA]; int nra = 0;      // pending (het index, 0) tags
    int64_t qpos[W2B_MAXQ]; uint32_t qalt[W2B_MAXQ]; int nq = 0;   // reconnect queue, ascending position

    // add_node (wfa_graph.rs:298-331); parents = the current reference_reconnect
    auto add_node = [&](uint32_t seq_off, uint32_t len, bool is_ref) -> int {
        if (nn == 0) { if (nrr != 0) { status = W2B_INVARIANT; return -1; } }
        else if (nrr == 0) { status = W2B_INVARIANT; return -1; }
        if (nn >= J.node_cap || nn >= 65535u || ne + (uint32_t)nrr > 65535u || len >= W2_IS_REF) {
            status = W2B_NEED_HOST; return -1;
        }
        nodes[nn].seq_off = seq_off;
        nodes[nn].len_ref = len | (is_ref ? W2_IS_REF : 0u);
        nodes[nn].child = 0;
        nodes[nn].c01 = 0;
        for (int k = 0; k < nrr; ++k) {
            const uint32_t p = rr[k];
            uint32_t ch = nodes[p].child;
            const uint32_t c = ch & 0xFFFFu;
            if (c == 0) nodes[p].c01 = nn;
            else if (c == 1) nodes[p].c01 |= nn << 16;
            else {
                if (c >= 65535u || 2u * (no + 1u) > J.edge_cap || no >= 32767u) { status = W2B_NEED_HOST; return -1; }
                if (c == 2) ch |= no << 16;
                edges[2u * no] = (uint16_t)p; edges[2u * no + 1u] = (uint16_t)nn;
                ++no;
            }
            nodes[p].child = ch + 1u;
        }
        ne += (uint32_t)nrr;
        return (int)nn++;
    };
    auto flush_ref_alleles = [&](uint32_t node) {
        for (int k = 0; k < nra; ++k) {
            if (nt >= J.tag_cap) { status = W2B_NEED_HOST; return; }
            tags[nt++] = w2_tag(node, ra[k], 0u);
        }
        nra = 0;
    };
    // wfa_graph.rs:168-189 and :256-272
    auto drain_one = [&]() -> bool {
        const int64_t alt_reconnect = qpos[0];
        const uint32_t alt_index = qalt[0];
        for (int k = 1; k < nq; ++k) { qpos[k - 1] = qpos[k]; qalt[k - 1] = qalt[k]; }
        --nq;
        if (!(alt_reconnect > previous_end)) { status = W2B_INVARIANT; return false; }
        const int ri = add_node((uint32_t)(previous_end - ref_start), (uint32_t)(alt_reconnect - previous_end), true);
        if (ri < 0) return false;
        flush_ref_alleles((uint32_t)ri);
        if (status != W2B_OK) return false;
        previous_end = alt_reconnect;
        nrr = 0;
        rr[nrr++] = (uint32_t)ri;
        rr[nrr++] = alt_index;
        while (nq > 0 && qpos[0] == alt_reconnect) {
            if (nrr >= W2B_MAXRR) { status = W2B_NEED_HOST; return false; }
            rr[nrr++] = qalt[0];
            for (int k = 1; k < nq; ++k) { qpos[k - 1] = qpos[k]; qalt[k - 1] = qalt[k]; }
            --nq;
        }
        return true;
    };
    auto queue_insert = [&](int64_t pos, uint32_t alt) -> bool {
        if (nq >= W2B_MAXQ) { status = W2B_NEED_HOST; return false; }
        int k = nq;
        while (k > 0 && qpos[k - 1] > pos) { qpos[k] = qpos[k - 1]; qalt[k] = qalt[k - 1]; --k; }
        qpos[k] = pos; qalt[k] = alt;
        ++nq;
        return true;
    };

    // stable merge of hets and homs by position, hets first on ties (wfa_graph.rs:137-144); both lists must arrive
    // sorted (they are slices of position-sorted vectors) - otherwise the host path sorts
    uint32_t ih = 0, im = 0;
    int64_t last_het = INT64_MIN, last_hom = INT64_MIN;
    while (status == W2B_OK && (ih < J.n_hets || im < J.n_homs)) {
        bool take_het;
        if (im >= J.n_homs) take_het = true;
        else if (ih >= J.n_hets) take_het = false;
        else take_het = vars[J.het_first + ih].position <= vars[J.hom_first + im].position;
        const W2Variant v = take_het ? vars[J.het_first + ih] : vars[J.hom_first + im];
        const int64_t vi = take_het ? (int64_t)ih : -1;
        if (take_het) { if (v.position < last_het) { status = W2B_NEED_HOST; break; } last_het = v.position; ++ih; }
        else { if (v.position < last_hom) { status = W2B_NEED_HOST; break; } last_hom = v.position; ++im; }
        if (vi >= 32768) { status = W2B_NEED_HOST; break; }
        if (v.flags & 1u) continue;                                   // is_ignored (wfa_graph.rs:147-150)
        if (v.position < ref_start) continue;                         // :155-159
        const int64_t pos = v.position;
        if (pos + (int64_t)v.ref_len > ref_end) continue;             // :160-164
        bool ok = true;
        while (ok && nq > 0 && qpos[0] <= pos) ok = drain_one();
        if (!ok) break;
        if (previous_end < pos || nn == 0) {                          // :196-209
            const int ri = add_node((uint32_t)(previous_end - ref_start), (uint32_t)(pos - previous_end), true);
            if (ri < 0) break;
            flush_ref_alleles((uint32_t)ri);
            if (status != W2B_OK) break;
            nrr = 0;
            rr[nrr++] = (uint32_t)ri;
            previous_end = pos;
        } else if (previous_end != pos) { status = W2B_INVARIANT; break; }
        if (v.flags & 2u) {                                           // allele0 is itself an ALT (:217-231)
            const int ai = add_node(v.a0_off, v.a0_len, false);
            if (ai < 0) break;
            if (vi >= 0) { if (nt >= J.tag_cap) { status = W2B_NEED_HOST; break; } tags[nt++] = w2_tag((uint32_t)ai, (uint32_t)vi, 0u); }
            if (!queue_insert(pos + (int64_t)v.ref_len, (uint32_t)ai)) break;
        } else if (vi >= 0) {
            if (nra >= W2B_MAXRA) { status = W2B_NEED_HOST; break; }
            ra[nra++] = (uint32_t)vi;                                 // tags the NEXT reference node (:233-237)
        }
        const int ai = add_node(v.a1_off, v.a1_len, false);           // :240-251
        if (ai < 0) break;
        if (vi >= 0) { if (nt >= J.tag_cap) { status = W2B_NEED_HOST; break; } tags[nt++] = w2_tag((uint32_t)ai, (uint32_t)vi, 1u); }
        if (!queue_insert(pos + (int64_t)v.ref_len, (uint32_t)ai)) break;
    }
    while (status == W2B_OK && nq > 0) if (!drain_one()) break;
    if (status == W2B_OK && !(previous_end <= ref_end)) status = W2B_INVARIANT;
    if (status == W2B_OK) {
        const int ri = add_node((uint32_t)(previous_end - ref_start), (uint32_t)(ref_end - previous_end), true);
        if (ri >= 0 && nra != 0) status = W2B_INVARIANT;   // assert!(reference_alleles.is_empty()) (:281)
    }
    info->n_nodes = nn;
    info->n_edges = ne;
    info->n_tags = nt;
    info->status = status;
}

// read_parsing.rs:790-800: traversed nodes in ascending id; first assignment wins, a different one -> Ambiguous
HP_HD void w2_map_alleles(const uint32_t* tags, uint32_t n_tags, const uint32_t* set, bool ok, uint8_t* alleles, uint32_t n_hets) {
    for (uint32_t k = 0; k < n_hets; ++k) alleles[k] = 3;   // NoOverlap
    if (!ok) return;
    for (uint32_t t = 0; t < n_tags; ++t) {
        const uint32_t w = tags[t], node = w & 0xFFFFu, vi = (w >> 16) & 0x7FFFu, a = w >> 31;
        if (!((set[node >> 5] >> (node & 31u)) & 1u)) continue;
        if (alleles[vi] == 3) alleles[vi] = (uint8_t)a;
        else if (alleles[vi] != (uint8_t)a) alleles[vi] = 2;   // Ambiguous
    }
}

// ---- compact wavefront state of hp_wfa2_kernel (per read, in LDS) ---------------------------------------------------
// A round's waves are kept per LIVE node as a dense array over that node's hull of diagonals (slots of the round's
// arena); the next round pulls from the previous round's arena (d+1: offset+1, d: offset+1, d-1: offset) and from
// the same-round injection list. The only state that outlives two rounds is the set of (node, diagonal) pairs whose
// wave reached its cap min(node length, read length - diagonal): see "capped diagonals" in hp_wfa2_kernel.hip.
constexpr int32_t W2_ST_OK = 0;
constexpr int32_t W2_ST_MAX_ED = 1;
constexpr int32_t W2_ST_NEED_BIG = 2;     // outgrew the compact state: the job is re-run by the dense-band kernel
constexpr int32_t W2_ST_PENDING = 7;
constexpr int32_t W2_ST_INTERNAL = -3;

constexpr uint32_t W2_KIND_NONE = 0;
constexpr uint32_t W2_KIND_FINISHED = 2;       // max_offset == node_length, not the last node: the node's children take it up THIS round
constexpr uint32_t W2_KIND_INTERIOR = 1;       // max_offset < node_length: the -1 diagonal gets a wave
constexpr uint32_t W2_KIND_INTERIOR_READ = 3;  // ... and the read has bases left: 0 / +1 diagonals too
constexpr uint32_t W2_KIND_END_LAST = 4;       // end of the LAST node with read left: only the +1 diagonal

constexpr uint32_t W2_MAX_STEPS = 1u << 24;    // watchdog on the tiles of one job
constexpr uint32_t W2_FMT_SUSPECT = 0x80u;       // bit of a job's format byte (block mode): routed past the compact kernels by the host (hp_wfa2.hip, layout_blocks)
constexpr int W2_SET_STRIDE = 16;              // out_sets: 16 words per job (graphs of up to 512 nodes: W2Cfg<16, true>)
constexpr int32_t W2_DIAG_LIM = 1 << 17;       // |diagonal| representable in a capped-set key
constexpr uint32_t W2_LDS_LEN_LIM = 1u << 18;  // node length / 1024 edges / 7 children: what the packed LDS node descriptor holds

// LDS layout of one read (group). Entries ("clusters": a run of diagonals of one node that holds a wave) are uint4
// headers: x = node | first slot << 16, y = first diagonal, z = live hull lo | hi << 8 | finished hull lo << 16 | hi << 24
// (relative to y; lo > hi = empty), w = node length. Entries with a live wave go to the round's L list (two rounds
// deep: the next round pulls from them); entries that only hold FINISHED waves are needed by the node's children in
// the same round only and go to the F list. A parent entry is named by a code: L index, or 128 + F index.
#ifndef W2_SLOTS_WIDE
#define W2_SLOTS_WIDE 384
#endif
#ifndef W2_MAXPREV_SMALL
#define W2_MAXPREV_SMALL 3
#endif
template <int W, bool WIDE = false> struct W2Cfg {
    static constexpr int MAXN = 32 * W;          // nodes
    static constexpr int MAXE = 2 * MAXN + 32;   // edges
    // Table sizes of the two smaller classes follow what jobs actually use (W2_STATS build on the default bench, jobs per
    // 10 000 that need more: live entries > 15: 2, finished-only entries > 19: 8, slots > 79: 17); the jobs that outgrow
    // them (about 0.3 %) are handed to the largest class on the device (W2Batch::esc)
    static constexpr int MAXL = W <= 4 ? 16 : 56;     // live entries per round
    static constexpr int MAXF = W <= 4 ? 20 : 32;     // finished-only entries per round
    // (node, diagonal) slots per round (86: what is left of 1 600 bytes). WIDE: the tables of the second launch (hp_wfa2.hip
    // late()) over reads that outgrew the slots above: at 2 % edit noise the waves off the best path live ten rounds before the
    // pruning floor reaches them (500 bases at ~50 a round), a node holds twenty-odd diagonals, and 80 % of the reads of such a
    // set need more than 86 slots. (Not the largest class's own tables: a read past the edit cap then crawls on to the cap with
    // hundreds of diagonals instead of leaving early for the reference-window test - the tail of every launch set grew by 10 ms.)
    static constexpr int SLOTS = WIDE ? W2_SLOTS_WIDE : (W <= 4 ? 86 : 144);
    static constexpr int MAXQ = 8;               // nodes waiting for their turn with waves handed over by parents
    static constexpr int MAXPAR = 8;             // finished parent entries of one node in one round
    static constexpr int MAXS = W <= 4 ? 8 : 12; // source intervals of one node (slow path scratch)
    static constexpr int MAXW = 250;             // diagonals per entry (8-bit relative hulls)
    static constexpr int MAXPREV = W <= 4 ? W2_MAXPREV_SMALL : 4;   // live entries of ONE node the next round can pull from (kept in registers)
    static constexpr int a16(int x) { return (x + 15) & ~15; }
    // The node table is read from HBM (L2-resident: one 16-byte descriptor per node visit, with the first two children
    // inline) - with the table sizes above that takes a read's LDS to 1 600 bytes, i.e. 12 workgroups of 8 reads per CU =
    // three wavefronts per SIMD for the two smaller classes. The kernel waits on memory half of the time; its throughput
    // follows the resident wavefronts almost linearly (4 / 6 / 8 workgroups per CU: 2.6 / 3.9 / 5.1 M reads/s).
    static constexpr int WAVES_PER_SIMD = W <= 4 ? 3 : 2;   // the register budget the kernel is compiled for (512 / waves)
    static constexpr int O_LIVE = 0;                                  // uint4[2][MAXL]
    static constexpr int O_FIN = O_LIVE + 2 * 16 * MAXL;              // uint4[MAXF]
    static constexpr int O_EK = O_FIN + 16 * MAXF;                    // u32[2][SLOTS]: offset << 3 | kind
    static constexpr int O_MISC = a16(O_EK + 2 * 4 * SLOTS);          // outset[W]
    static constexpr int O_SRC = a16(O_MISC + 4 * W);                 // int2[MAXS] item intervals (far-apart sources only)
    static constexpr int BYTES = a16(O_SRC + 8 * MAXS);
    static constexpr int SET_DWORDS = 2 * SLOTS * W;                  // per group in HBM: the slots' traversed-node sets
    // ... followed by one 16-byte capped-diagonal record per node (tag, anchor diagonal, 64 bits: diagonal anchor - 32 + k
    // has reached its cap); diagonals outside a node's window go to the group's hash set
    static constexpr int REC_DWORDS = 4 * MAXN;
    static constexpr int GROUP_DWORDS = SET_DWORDS + REC_DWORDS;
};

// ---- third generation (round 4): a round's waves as ONE FLAT SORTED LIST of (node, diagonal) slots --------------------------------
// hp_wfa3_kernel.hip. A lockstep step of the second generation is one NODE's diagonals per group; a HiFi read's round holds a
// dozen live slots spread over three or four nodes, so a round costs three or four steps with a third of the lanes busy. Here
//   * a round's live waves are one list sorted by key = node << 19 | (diagonal + 2^18), 8 bytes a slot (key, offset << 3 | kind);
//   * the next round's TARGET list is built from it in one pass (every live slot (n, d) emits (n, d - 1), (n, d), (n, d + 1) unless
//     its predecessor already did): a target's candidates (d + 1: offset + 1, d: offset + 1, d - 1: offset) are the three list
//     entries from its emitter on;
//   * a step is a TILE of the next G targets whatever nodes they belong to: node descriptor, sequence pointer, capped-diagonal
//     record per LANE;
//   * waves that finish a node THIS round (wfa_graph.rs:527-553) become targets (child, diagonal + length) inserted into the
//     sorted target list, or merged into the target that is already there. A tile that holds a finishing wave of node q only
//     COMMITS its slots of nodes < min child(q) (node ids are topological: nothing before q's first child can be reached from q
//     this round); the others stay targets and are computed again - with what the finished waves hand them - in the next tile.
// Same results as the second generation by construction (same candidates, same decisions, per (node, diagonal)); tests/cpp/
// wfa2_model.cpp holds a CPU model of this formulation too (w3m_wfa_assign) and tests/test_wfa2_model.py pins it to the oracle.
constexpr uint32_t W3_DIAG_BIAS = 1u << 18;
constexpr uint32_t W3_DIAG_MASK = (1u << 19) - 1u;
HP_HD uint32_t w3_key(uint32_t node, int32_t diag) { return (node << 19) | (uint32_t)(diag + (int32_t)W3_DIAG_BIAS); }
HP_HD uint32_t w3_key_node(uint32_t key) { return key >> 19; }
HP_HD int32_t w3_key_diag(uint32_t key) { return (int32_t)(key & W3_DIAG_MASK) - (int32_t)W3_DIAG_BIAS; }
// target aux word: back | src0 << 10 | src1 << 20 | start wave << 30. back = index (previous round's live list) of the target's
// emitter; src = set-arena index of a wave that finished a parent this round and lands on this target; 0x3FF = none
constexpr uint32_t W3_NONE = 0x3FFu;
constexpr uint32_t W3_START = 1u << 30;
HP_HD uint32_t w3_aux(uint32_t back, uint32_t src0, uint32_t src1) { return back | (src0 << 10) | (src1 << 20); }

template <int W, bool WIDE = false> struct W3Cfg {
    static constexpr int MAXN = 32 * W;
    static constexpr int MAXE = 2 * MAXN + 32;
    // slots per round: targets (consumed from the front) and live slots (appended behind them) share one array per round parity;
    // the sets of waves that finished a node this round take the round's set arena from the top
#ifndef W3_OCC
#define W3_OCC 3   // wavefronts per SIMD the two smaller classes are built for (experiment: 4 = 70 slots, 128 registers)
#endif
    static constexpr int SLOTS = WIDE ? W2_SLOTS_WIDE : (W <= 4 ? (W3_OCC >= 4 ? 70 : 88) : 144);
    static constexpr int WAVES_PER_SIMD = W <= 4 ? W3_OCC : 2;
    static constexpr int a16(int x) { return (x + 15) & ~15; }
    static constexpr int O_A = 0;                                   // uint2[2][SLOTS]
    static constexpr int QN = W <= 4 ? 16 : 32;                      // child targets of the waves that finished in one tile (two per lane)
    static constexpr int O_Q = a16(O_A + 2 * 8 * SLOTS);             // uint2[QN]: key, set-arena index of the finished wave
    static constexpr int O_MISC = O_Q + 8 * QN;                      // outset[W]
    static constexpr int BYTES = a16(O_MISC + 4 * W);
#ifndef W3_SETIDS
#define W3_SETIDS 0   // hp_wfa3_kernel.hip: 1 = slots carry the index of an immutable set entry instead of a copy of the set (round 6: built,
                      // bit-identical, HBM traffic 12.2 -> 7.6 GB a set - and 3 % slower, 4 % at 1 % noise: measured, left off; profiles/DIARY.md)
#endif
#ifndef W3_ARENA
#define W3_ARENA 1023   // entries of a job's set arena (W3_SETIDS; indices are 10 bits, 0x3FF = none)
#endif
    static constexpr int SET_ENTRIES = W3_SETIDS ? W3_ARENA : 2 * SLOTS;
    static constexpr int SET_DWORDS = (SET_ENTRIES * W + 3) & ~3;    // per group in HBM: the traversed-node sets
    static constexpr int REC_DWORDS = 4 * MAXN;                      // + one capped-diagonal record per node
    static constexpr int GROUP_DWORDS = SET_DWORDS + REC_DWORDS;
};
static_assert(W2_SLOTS_WIDE < 0x3FF && W3_ARENA <= 0x3FF, "set-arena indices are 10 bits");

struct W2Batch {
    const W2Job* jobs;
    const W2Info* info;
    const uint32_t* order;     // job ids of this launch's class, longest read first
    uint32_t* next;            // work queue head of this launch (zeroed before the launch): groups pull jobs from `order`
    uint32_t n_items;        // jobs of this class (= *n_items_dev)
    const uint32_t* n_items_dev;   // written by hp_wfa2_scatter_kernel
    uint32_t tag_base;         // capped-set keys carry tag_base + job + 1 (never 0 = empty)
    const W2Node* nodes;
    const uint16_t* edges;
    const uint8_t* seq;
    uint64_t alt_off;          // byte offset of the allele pool inside seq[]
    uint32_t* out_sets;        // [n_jobs][W2_SET_STRIDE]
    uint64_t* out_score;
    uint32_t* out_work;        // [n_jobs][2]: (node, diagonal) wave updates (the kernel), bytes of the traversed nodes (the map kernel)
    int32_t* status;
    uint64_t* htab;            // [groups][1 << hcap_log2] capped-diagonal hash sets, never cleared (tagged)
    uint32_t* gsets;           // [groups][set_stride] traversed-node sets of the arena slots (consumed off the critical path)
    uint32_t set_stride;
    uint32_t hcap_log2;
    uint64_t prune_distance;   // UINT64_MAX disables pruning
    uint64_t max_ed;
    // Escalation: a job that outgrows the tables of its class is handed to the largest class WHILE that class's kernel is
    // still running (its groups keep a ticket for the next list position and wait for it to be published), instead of
    // waiting for a pass of its own after all three. esc[0] = list positions reserved, esc[1] = positions published
    // (in order), esc[2] = producer workgroups that have exited, esc[3] = scratch, esc[4] = producer workgroups that have started. Every access is an atomic RMW: the
    // L2s of different XCDs are not coherent for plain loads and stores inside a kernel.
    uint32_t* esc;
    uint32_t* esc_order;       // the largest class's job list (capacity: the whole batch)
    uint32_t esc_role;         // 0 none, 1 producer, 2 consumer (the largest class)
    uint32_t esc_producers;    // consumer: producer workgroups to wait for
    uint32_t esc_limit;        // producer: no hand-over once this many list positions are taken (the class's own jobs + a budget)
    uint8_t* handed;           // [n_jobs]: set by the producer that hands a job over (zeroed before the launch)
    uint32_t hopeless_pct;     // third generation: a job that outgrows its lists is NOT handed over when its projected edit distance exceeds this share of max_ed (it goes to the host's pass)
    uint32_t group_jobs;       // > 0: a group leaves after this many jobs (the grid then has a workgroup for every NG x group_jobs jobs:
                               // workgroups retire all through the launch and other streams' kernels get their slots); 0: persistent
};

}  // namespace hp
