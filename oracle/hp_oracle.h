/*
 * hp_oracle.h — C API of liboracle.so.
 *
 * TEST INFRASTRUCTURE ONLY. This is a line-faithful CPU restatement of the reference algorithms
 * (PacificBiosciences/HiPhase v1.5.0). Only tests/, __graft_entry__.smoke() and the `cpu_baseline`
 * leg of bench.py may load it. The product (libhiphase_gpu.so) never links or calls it.
 *
 * Parity status: the reference is Rust and cannot be built in this image (no cargo/rustc, crates not
 * vendored), so there is no oracle/_ref. The restatement is pinned against every known-answer test
 * the reference holds for this path (transcribed to tests/golden/ (JSON)):
 *   read_segments.rs:214-308, astar_phaser.rs:663-798, wfa_graph.rs:677-1208,
 *   sequence_alignment.rs:45-76, variants.rs:838-845, phaser.rs:757-804.
 * PARITY UNPINNED by any reference test (no read-bearing fixture exists upstream): astar_subsolver,
 * calculate_astar_heuristic, astar_solver end-to-end. For those the evidence is the line-by-line
 * restatement + brute-force MEC equality on small blocks + the reference's own asserts.
 */
#ifndef HP_ORACLE_H
#define HP_ORACLE_H
#include "../include/hiphase_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- read_segments.rs ------------------------------------------------------------------------ */
/* ReadSegment::new (read_segments.rs:40-62): clips [first set, last set+1). Writes region into
 * *start,*end; clipped alleles/quals are alleles[*start..*end). */
void hpo_read_segment_new(const uint8_t* alleles, size_t len, size_t* start, size_t* end);
/* ReadSegment::collapse (read_segments.rs:71-121) over k full-length (block-length `len`) rows given
 * unclipped (NoOverlap/0 outside their regions). Output: full-length out_alleles/out_quals + region.
 * Returns 0, or HP_ERR_INVARIANT if the `assert!(quals[i] > 0)` would fire. */
int hpo_read_segment_collapse(const uint8_t* alleles, const uint8_t* quals, size_t k, size_t len,
                              uint8_t* out_alleles, uint8_t* out_quals, size_t* start, size_t* end);
/* score_partial_haplotype (read_segments.rs:177-206) for a clipped row. */
uint64_t hpo_score_partial_haplotype(const uint8_t* row_alleles, const uint8_t* row_quals,
                                     size_t start, size_t end,
                                     const uint8_t* haplotype, size_t hap_len, size_t offset);

/* ---- astar_phaser.rs -------------------------------------------------------------------------- */
/* Walks AstarNode::new / new_extended_node (astar_phaser.rs:47-119) along a given allele path:
 * path1/path2[len], heuristic_costs[len+1], hap_offset. Outputs per step: frozen, total cost, num_hets. */
int hpo_astar_node_walk(const hp_block_view* blk, const uint8_t* path1, const uint8_t* path2, size_t len,
                        const uint64_t* heuristic_costs, size_t hap_offset,
                        uint64_t* frozen, uint64_t* total, uint64_t* num_hets);
/* PQueueHapTracker script (astar_phaser.rs:171-231): ops[i] = 0 add, 1 remove, 2 increase_threshold;
 * out_len[i] = len() after op i. */
int hpo_hap_tracker_script(size_t max_hap_length, const int32_t* ops, const uint64_t* values, size_t n,
                           uint64_t* out_len);
/* calculate_astar_heuristic (astar_phaser.rs:246-292): heuristics[N+1]. */
int hpo_astar_heuristic(const hp_block_view* blk, const hp_astar_params* p, uint64_t* heuristics);
/* astar_solver (astar_phaser.rs:426-633). counters/heuristics may be NULL. */
int hpo_astar_solve(const hp_block_view* blk, const hp_astar_params* p, uint8_t* h1, uint8_t* h2,
                    hp_phase_stats* out, hp_work_counters* counters, uint64_t* heuristics);
/* Exhaustive MEC over {0,1}^N x {0,1}^N (N <= 10): minimal sum_r min(score(h1), score(h2)). */
uint64_t hpo_bruteforce_mec(const hp_block_view* blk);

/* ---- phaser.rs post-processing ---------------------------------------------------------------- */
/* get_solution_span_counts (phaser.rs:350-388): out[N-1]. */
int hpo_solution_span_counts(const hp_block_view* blk, const uint8_t* h1, const uint8_t* h2, uint64_t* out);
/* haplotag_reads (phaser.rs:714-750): per row haplotag (0,1, or 2 = untagged) and block tag. */
int hpo_haplotag_reads(const hp_block_view* blk, const uint8_t* h1, const uint8_t* h2,
                       const uint64_t* block_tags, uint8_t* haplotag, uint64_t* phase_block);

/* ---- sequence_alignment.rs -------------------------------------------------------------------- */
uint64_t hpo_edit_distance(const uint8_t* v1, size_t l1, const uint8_t* v2, size_t l2);

/* ---- read_parsing.rs: local re-alignment (hp_oracle_local.cpp; PARITY UNPINNED upstream) -------- */
/* Variant::match_allele (variants.rs:598-606) -> 0 / 1 / 2 */
int hpo_match_allele(const hp_local_variant* variant, const uint8_t* allele, size_t len);
/* Variant::closest_allele_clip (variants.rs:624-641) -> AlleleType 0/1/2 (+ min / other distance), -1 on a failed assert */
int hpo_closest_allele_clip(const hp_local_variant* variant, const uint8_t* allele, size_t len, size_t head_clip,
                            size_t tail_clip, uint64_t* dmin, uint64_t* dother);
/* local_realignment (read_parsing.rs:121-503) for ONE record: alleles/quals hold num_variants bytes; -3 = panic */
int hpo_local_realignment(const hp_local_read* read, const hp_local_variant* variant_calls, size_t num_variants,
                          uint8_t* alleles, uint8_t* quals, hp_read_stats* stats);

/* ---- wfa_graph.rs ------------------------------------------------------------------------------ */
typedef struct hpo_graph hpo_graph;
hpo_graph* hpo_graph_new(uint64_t max_edit_distance);
void       hpo_graph_free(hpo_graph* g);
/* WFAGraph::add_node (wfa_graph.rs:298-331): returns node index or <0 on the reference's bail!(). */
int64_t    hpo_graph_add_node(hpo_graph* g, const uint8_t* seq, size_t len, const uint64_t* parents, size_t n_parents);
/* WFAGraph::from_reference_variants_with_hom (wfa_graph.rs:119-284) using the job's reference,
 * variants and [ref_start, ref_end). */
hpo_graph* hpo_graph_from_job(const hp_wfa_job* job, uint64_t max_edit_distance, int* status);
uint64_t   hpo_graph_num_nodes(const hpo_graph* g);
/* node_to_alleles lookup: writes up to cap (variant_index, allele) pairs, returns the count. */
size_t     hpo_graph_node_alleles(const hpo_graph* g, uint64_t node, uint64_t* var_idx, uint8_t* allele, size_t cap);
/* node sequence / parents / edges accessors for structure tests */
size_t     hpo_graph_node_seq(const hpo_graph* g, uint64_t node, uint8_t* out, size_t cap);
size_t     hpo_graph_node_parents(const hpo_graph* g, uint64_t node, uint64_t* out, size_t cap);
size_t     hpo_graph_node_edges(const hpo_graph* g, uint64_t node, uint64_t* out, size_t cap);
/* edit_distance_with_pruning (wfa_graph.rs:350-650). Returns HP_OK or HP_WFA_MAX_ED.
 * traversed: sorted node ids, *n_traversed in: capacity, out: count. shuffle_seed != 0 randomises the
 * iteration order of every hash-map the reference iterates (results must not depend on it). */
int        hpo_graph_edit_distance(const hpo_graph* g, const uint8_t* other, size_t other_len,
                                   uint64_t prune_distance, uint64_t shuffle_seed,
                                   uint64_t* score, uint64_t* traversed, size_t* n_traversed);
/* Independent check of the above (hp_oracle_brute.cpp): every root -> last-node path spelled out, plain Levenshtein against each.
 * out[0] min distance, [1] paths, [2] optimal paths, [3] union of their nodes (bit per node), [4] some optimal path lies inside wfa_mask. */
int        hpo_graph_bruteforce(const hpo_graph* g, const uint8_t* other, size_t other_len, uint64_t wfa_mask, uint64_t out[5]);
/* Full per-job path: graph build + WFA + allele mapping of read_parsing.rs:790-800. */
void       hpo_wfa_hull_stats(uint64_t out[8], int reset);   /* measurement aid: hulls of live diagonals per (round, node) visit, hp_oracle_wfa.cpp */
int        hpo_wfa_assign(const hp_wfa_job* job, uint64_t prune_distance, uint64_t max_ed,
                          hp_wfa_result* out, uint8_t* alleles);

/* ---- the whole path for one block (hp_oracle_block.cpp): phaser::solve_block from the decoded records on ------------
 * (phaser.rs:513-630 = load_full_read_segments / load_read_segments, astar_solver, get_solution_span_counts, haplotag_reads).
 * Same structs in and out as the product's hp_solve_blocks; segments in first-seen read-name order. PARITY UNPINNED upstream
 * (no read-bearing fixture exists): assembled from the pinned pieces above in the reference's statement order. */
int hpo_solve_block(const hp_block_input* block, const hp_block_params* params, hp_block_output* out);

#ifdef __cplusplus
}
#endif
#endif
