// hp_wfa_dev.h — device-side layout of the graph-WFA allele-assignment kernel (host packer + kernel).
//
// One job = one BAM record (reference src/read_parsing.rs:738-780): a small DAG of sequence nodes
// (reference src/wfa_graph.rs:24-68) plus the read bases. One wavefront aligns one job.
//
// Wavefront state is kept per (node, diagonal) in a dense band (diag = other_start of wfa_graph.rs:439-442):
// for node n every reachable diagonal lies in [emin_n - ED, emax_n + ED] where emin/emax are the shortest /
// longest graph path lengths from the root to the node's first base, so the band is addressed directly
// (no hash maps): di = diag - (emin_n - band - 1), width_n = emax_n - emin_n + 2*band + 3.
// Entry layout of one (node, diagonal) slot, in dwords (W = ceil(n_nodes/32) set words):
//   [0]              maxfront (max_wavefronts, wfa_graph.rs:360,464-470)
//   [1 .. 2+W]       result of even rounds:  max_offset, kind, traversed-node set
//   [3+W .. 4+2W]    result of odd rounds
//   [5+2W + k*W ..]  k-th parent's same-round injection (set only; its offset is always 0, wfa_graph.rs:552)
// The next round PULLS its in-node candidates from the previous round's results of diagonals d+1, d, d-1
// (what the reference pushes at wfa_graph.rs:555-573 and :516-523); parents push injections.
#pragma once
#include <stdint.h>

#include "../../include/hiphase_gpu.h"

namespace hp {

constexpr uint32_t WFA_NODE_IS_REF = 0x8000u;   // bit of WfaNode::n_parents: the node is a span of the reference slice
struct WfaNode {          // 32 B
    uint32_t seq_off;     // into the job's reference slice (WFA_NODE_IS_REF) or into its private bytes
    uint32_t seq_len;
    uint32_t child_off;   // into the job's edge list
    uint16_t n_children;
    uint16_t n_parents;   // number of injection slots (node 0 has one virtual parent: the start wave) | WFA_NODE_IS_REF
    int32_t  dbase;       // diagonal of di == 0
    uint32_t width;       // number of diagonals in the band
    uint32_t entry_off;   // dword offset of this node's band inside the job scratch
    uint32_t entry_stride;// dwords per diagonal slot
};

struct WfaEdge {
    uint32_t child;
    uint32_t ordinal;     // which injection slot of the child this parent owns
};

struct WfaJobDesc {       // 72 B
    uint64_t node_off;    // into nodes[]
    uint64_t edge_off;    // into edges[]
    uint64_t seq_off;     // into seq[] (bytes): the job's private bytes = [alt allele bytes][read][pad]
    uint32_t n_nodes;
    uint32_t set_words;   // W
    uint32_t read_off;    // read position inside the job's private bytes
    uint32_t read_len;
    uint32_t band;        // edit-distance capacity of this layout
    uint32_t scratch_dwords;
    uint32_t n_edges;
    uint32_t pad;
    uint64_t out_set_off; // into out_sets[] (dwords)
    uint64_t ref_off;     // into seq[] (bytes): first base of the job's reference window inside the shared upload
};

constexpr int32_t WFA_ST_OK = 0;
constexpr int32_t WFA_ST_MAX_ED = 1;     // Err(MaxEditDistance) (wfa_graph.rs:645-648)
constexpr int32_t WFA_ST_NEED_BAND = 2;  // edit distance exceeded this launch's band: host re-runs with a wider one
constexpr int32_t WFA_ST_PENDING = 7;
constexpr int32_t WFA_ST_UNSUPPORTED = 9;   // host-side verdict: outside the kernels' limits (soft: HP_WFA_UNSUPPORTED for this job only)
constexpr int32_t WFA_ST_INTERNAL = -3;

constexpr uint32_t WFA_KIND_NONE = 0;
constexpr uint32_t WFA_KIND_INTERIOR = 1;       // max_offset < node_length: -1 diagonal always gets a wave
constexpr uint32_t WFA_KIND_INTERIOR_READ = 3;  // ... and the read has bases left: 0 / +1 diagonals too
constexpr uint32_t WFA_KIND_END_LAST = 4;       // end of the LAST node with read left: only the +1 diagonal

constexpr uint32_t WFA_MAX_NODES = 1024;        // LDS budget: 24 B of state + a 32 B copy of the node table entry per node; larger graphs: hp_wfa_big_kernel
constexpr uint32_t WFA_MAX_PARENTS = 64;        // injection slots of one node (a 64-bit mask per diagonal)
constexpr uint32_t WFA_NODE_STATE_BYTES = 24;
constexpr uint32_t WFA_NODE_LDS_BYTES = WFA_NODE_STATE_BYTES + 32;

struct WfaBatchDev {
    const WfaJobDesc* jobs;
    const uint32_t* order;
    uint32_t n_items;
    const WfaNode* nodes;
    const WfaEdge* edges;
    const uint8_t* seq;
    uint32_t* out_sets;      // traversed-node bitsets
    uint64_t* out_score;
    int32_t* status;
    uint32_t* scratch;       // [slots][scratch_stride] dwords, zero-initialised, left zeroed by every job
    uint64_t scratch_stride;
    uint64_t prune_distance; // UINT64_MAX disables pruning
    uint64_t max_ed;
    uint32_t lds_nodes_off;  // byte offset of the node-table copy in LDS (after the NodeState array, 16-byte aligned)
    uint32_t lds_edges_off;  // byte offset of the edge-list copy (after the node table)
    unsigned char* big_state;  // hp_wfa_big_kernel: [slots][big_stride] per-node state in HBM (graphs beyond WFA_MAX_NODES)
    uint64_t big_stride;
};

}  // namespace hp
