// hp_astar_dev.h — device-side data layout of the A* MEC solver (shared by host packer and kernel).
//
// HBM layout of one batch (all blocks concatenated, see DESIGN.md "Data layout in HBM"):
//   desc[n_blocks]                      BlockDesc
//   vlo/vhi[sum N]    u32               per variant p: candidate rows are [vlo[p], vhi[p]) in start-sorted
//                                       order (the IntervalTree::find(p..p+1) replacement, astar_phaser.rs:92)
//   vflags[sum N]     u8                HP_VAR_*
//   rstart/rend/rword[sum R] u32        rows sorted by start; rword = index of the row's first plane word
//   words[sum W][12]  u32               bit-sliced matrix: one "plane word" covers 32 consecutive variants
//                                       (aligned to absolute variant index / 32) of one row:
//                                       [0] allele bit0, [1] allele bit1, [2..9] qual bit-planes 0..7, [10..11] pad.
//                                       Cells outside the row's region are allele 3 (NoOverlap), qual 0.
//   ctab[sum N'][64]  u32               per-position cell table (blocks with max_cov <= 64 only), see CELL_* below
//   H[sum (N+1)]      u64               heuristic array (output, astar_phaser.rs:252)
//   h1/h2[sum N]      u8                haplotypes (output)
//   stats/counters/status per block
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hiphase_gpu.h"

namespace hp {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr int WORD_DWORDS = 12;  // 48 B per plane word -> three 16-B loads

struct BlockDesc {
    uint32_t n_vars;
    uint32_t n_reads;
    uint64_t var_off;   // into vlo/vhi/vflags/h1/h2
    uint64_t read_off;  // into rstart/rend/rword
    uint64_t word_off;  // into words (units of plane words)
    uint64_t h_off;     // into H
    uint32_t max_cov;   // max over p of vhi-vlo
    uint32_t n_words;
    uint64_t chunk_off; // into hapw (one Win per 32-variant chunk of the solution)
    uint64_t cell_off;  // into ctab (entries); ~0 when the block has no cell table (HP_NO_CTAB)
    uint32_t ctab_shift; // log2(entries per variant) of the block's cell table: 6 or 7
    uint32_t pad;
};
// device-only bit of vflags[p] (the caller's bits are HP_VAR_IGNORED / HP_VAR_SNV): two rows covering p share a cell
// table entry (same row index mod 64), so the incremental path must not be used at p
constexpr uint8_t VAR_NOFAST = 0x4;

// Per-position cell table (built on the device by hp_build_ctab_kernel): E u32 entries per variant p, E = 64 for
// blocks with at most 64 candidate rows per variant (max_cov), else 128; entry (row index mod E) describes that
// row's cell at p (lane = entry & 63, tile = entry >> 6). Variants where two covering rows would share an entry
// carry VAR_NOFAST and are handled by the plane-word path.
//   bits 0-7 x0 = cost of giving a haplotype allele 0 at p against this row ((allele != 0) ? qual : 0) | 8-15 x1 =
//   the same for allele 1 (both 0 for an ignored variant: astar_phaser.rs:348-364 scores nothing there) |
//   16 row ends here (end == p+1) | 17 valid (row covers p) | 18-31 min(p - row start, 2^14-1)
constexpr uint32_t CELL_ENDS = 1u << 16, CELL_VALID = 1u << 17, CELL_T_SHIFT = 18, CELL_T_MAX = (1u << 14) - 1;

// where a block's cells start in the concatenated caller arrays uploaded for the device-side packing
struct PackRaw { uint64_t allele_off, qual_off; };   // bytes into the 2-bit allele / u8 quality uploads

// Priority key (astar_phaser.rs:131-133): min total cost, then MORE hets, then OLDER node.
//   hi = cost << 24 | (0xFFFFFF - num_hets)      cost < 2^40 (sum of all quals of a block < 2^40)
//   lo = node_index << 26 | rank << 24 | depth   node_index < 2^38, creation rank among siblings (0..3),
//                                                depth <= N < 2^24  (rank never decides: node_index is unique)
// Lexicographic (hi, lo) order == the reference's pop order; node_index is unique so it is total.
// depth rides in the low bits so the full prune (astar_phaser.rs:576-581) needs no node lookup.
struct Key {
    uint64_t hi, lo;
};

struct Win {  // haplotype window over one 32-variant chunk
    uint32_t h1, h2;  // allele bit of each haplotype (valid where nv == 0)
    uint32_t nv;      // 1 = not assigned yet, before the sub-problem offset, or (2,2)
};

// Search-tree state in HBM. O(1) per node instead of the reference's O(len) Vec copies (astar_phaser.rs:79-82):
//  * one FamRec per EXPANSION (not per child): everything the siblings share + the four per-slot frozen
//    increments and cost sums; a child is rebuilt from (family, creation rank) only if it is ever popped from the
//    queue — on HiFi-like data 97 % of the children never are. Stored at fam[node_index of the first child].
//    The sub-solver queues ONE key per family (its best not-yet-popped child) and derives the next one from the
//    record when that child is popped: the pop sequence is the reference's, the heap is a third of the size.
//  * one ChunkRec per completed 32-variant haplotype chunk on a path: w0 = that chunk, w1 = the chunk before,
//    anc2 = the ChunkRec two chunks back. A node carries (w0, w1, anc1, anc2), so any look-back costs one hop
//    per two chunks and rows shorter than 64 variants never touch memory.
struct FamRec {  // 80 B
    uint64_t frozen;       // parent's frozen cost
    uint32_t depth_flags;  // parent depth | bad << 30 | has_10_child << 31
    uint32_t hets;         // parent's num_hets
    uint32_t anc1, anc2;   // the children's chunk links
    Win base;              // parent's window in the children's chunk (fresh when they open a new chunk)
    Win w1;                // the chunk before it
    uint32_t sumF[4];      // frozen increment of slot 0..3 = (0,1) (1,0) (0,0) (1,1)
    uint32_t tot[4];       // frozen + fluid increment of the slot: total = frozen + tot[slot] + H[child depth]
};
static_assert(sizeof(FamRec) == 80, "FamRec must be 80 bytes");
struct ChunkRec {  // 32 B
    Win w0, w1;
    uint32_t anc2, pad;
};
static_assert(sizeof(ChunkRec) == 32, "ChunkRec must be 32 bytes");

struct SolveParams {
    uint32_t minq_main;   // min_queue_size
    uint32_t minq_sub;    // min_queue_size / 10 (astar_phaser.rs:266)
    uint32_t qinc;        // queue_increment
    uint32_t max_seg;     // 40 (astar_phaser.rs:466)
    uint32_t cap_sub;     // node capacity of the sub-solver pool/heap
    uint32_t jcap_sub;    // per-lane heap capacity (sub)
    uint32_t cap_main;    // node capacity of the main pool/heap
    uint32_t jcap_main;   // per-lane heap capacity (main)
    uint32_t sub_heap_in_lds;
    uint32_t max_n_vars;  // largest N among the blocks of this launch (tracker stride)
    uint32_t seg_profile; // 1: launch the s_memtime-instrumented kernel variant (HP_SEG_PROFILE, tuning aid)
    uint32_t sub_idx_bits; // node-index bits of the sub-solver's packed key: 14, or 20 (wide index: large --phase-min-queue-size)
    uint32_t cap_chunk_sub, cap_chunk_main;  // ChunkRec capacities
    uint32_t save_state;  // 1: the sub-solver pool has room for the per-expansion prefix scores (TILES == 2 launches)
    uint32_t pad3;
};

// Per-slot scratch of the sub-solver: [cap_sub x FamRec | cap_chunk_sub x ChunkRec | cap_sub x 64 lanes x 16 B of
// saved prefix scores (the incremental-scoring state of every expansion, so that a node popped from the queue
// resumes on the cell-table path)]
inline __host__ __device__ size_t sub_pool_vec_off(const SolveParams& prm) {
    return ((size_t)prm.cap_sub * sizeof(FamRec) + (size_t)prm.cap_chunk_sub * sizeof(ChunkRec) + 15) & ~(size_t)15;
}
inline __host__ __device__ size_t sub_pool_bytes_per_slot(const SolveParams& prm) {
    return sub_pool_vec_off(prm) + (prm.save_state ? (size_t)prm.cap_sub * 64 * 16 : 0);
}

// ---- segment-parallel heuristic (DESIGN.md §3.1 "speculative segments") --------------------------------------
// The heuristic chain H[v] = f(H[v+1..v+40], clip) is sequential, but it forgets its start: a chain started cold
// (H = 0, clip = 1) ~100 variants above a seam reproduces the exact differences H[v] - H[v+1] below it. A large
// block is therefore cut into segments that run concurrently, each warming up over `warm` extra variants; a seam
// is accepted only if the 40-value state (relative H + clip) entering the owned range is IDENTICAL to what the
// segment above produced — then everything below is the same function of the same inputs, i.e. exact. A failed
// seam sends the block down the sequential path.
constexpr uint32_t SEG_STATE = 40;   // >= max_segment_size look-ahead (host enforces max_seg <= 40 when segmenting)
struct SegDesc {
    uint32_t blk;
    uint32_t a, b;      // owned variants [a, b)
    uint32_t v0;        // cold start: chain runs v0-1 .. a   (v0 == N for the top segment: the true block end)
};
struct SegOut {
    uint64_t seam[SEG_STATE];  // warm-up values H_w[b + j], j = 0..39 (relative to this segment's cold start)
    uint32_t clip_at_b;        // clip entering v = b-1 as the warm-up computed it
    uint32_t clip_out;         // clip entering v = a-1 (what the segment below must have warmed up to)
    int32_t status;
    uint32_t pad;
    hp_work_counters ctr;      // owned steps only
};

// block status written by the kernel
constexpr int32_t ST_OK = 0;
constexpr int32_t ST_OVERFLOW = 1;        // sub-solver pool/heap capacity exceeded (cannot happen with the host's sizing)
constexpr int32_t ST_OVERFLOW_MAIN = 2;   // main-search scratch exceeded: H[] is complete, host re-launches with 4x scratch
constexpr int32_t ST_INVARIANT = -3;      // a reference assert!/panic! would have fired
constexpr int32_t ST_PENDING = 7;
constexpr int32_t ST_H_READY = ST_OVERFLOW_MAIN;  // same meaning for the solver: heuristic complete, run the main search

struct BatchDev {
    const BlockDesc* desc;
    const uint32_t* order;   // LPT work list: indices of the blocks this launch solves
    uint32_t n_items;
    const uint32_t *vlo, *vhi;
    const uint8_t* vflags;
    const uint32_t *rstart, *rend, *rword;
    const uint32_t* words;
    const uint32_t* ctab;    // per-position cell tables (see CELL_*)
    uint64_t* H;
    uint8_t *h1, *h2;
    Win* hapw;            // packed solution windows, written at emission (input of the post-processing kernels)
    hp_phase_stats* stats;
    hp_work_counters* counters;
    int32_t* status;
    // per-workgroup-slot scratch
    unsigned char* sub_pool;   // [slots][cap_sub x FamRec | cap_chunk_sub x ChunkRec]
    unsigned char* main_pool;  // [slots][cap_main x FamRec | cap_chunk_main x ChunkRec]
    uint64_t* sub_heap_g; // [slots][jcap_sub*64] packed sub keys (only when the sub heap does not fit LDS)
    Key* main_heap;       // [slots][jcap_main*64]
    uint32_t* tracker;    // [slots][max_n_vars+1]
    // what the segment-parallel heuristic left for this launch's blocks (null / 0 segments: none): a block it could not accept
    // takes over every segment whose seam closes against the true chain (heuristic_phase)
    const SegDesc* segs;
    const SegOut* seg_out;
    const uint32_t* blk_seg_first;   // per block: its first segment (segments of a block are consecutive, bottom first)
    const uint32_t* blk_seg_n;       // per block: how many
    SolveParams prm;
};

}  // namespace hp
