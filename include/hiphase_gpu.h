/*
 * hiphase_gpu.h — C ABI of libhiphase_gpu.so: the MI355X (gfx950) phasing core that drops in
 * behind HiPhase's per-block entry point (reference src/phaser.rs:406 `solve_block`).
 *
 * The reference has no FFI for this path; the seams this ABI replaces are the Rust call sites
 *   - src/phaser.rs:541-543            astar_phaser::astar_solver(...)            -> hp_astar_solve*
 *   - src/read_parsing.rs:769-780      WFAGraph::from_reference_variants_with_hom + edit_distance_with_pruning
 *                                                                                  -> hp_wfa_assign_batch
 *   - src/data_types/variants.rs:627   sequence_alignment::edit_distance           -> hp_edit_distance_batch
 *   - src/read_parsing.rs:75,570       local_realignment(&read, variant_calls)     -> hp_local_realign_batch
 *   - src/phaser.rs:546,614-623        get_solution_span_counts / haplotag_reads   -> hp_batch_postprocess
 * INTEGRATION.md shows the `extern "C"` block + call-site patch a HiPhase maintainer would add.
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, no pointer outlives a call
 * (except opaque hp_batch handles). All entry points are re-entrant (called from HiPhase's
 * `--threads` pool, src/main.rs:332,385). Return value: 0 = OK, >0 = per-item soft status,
 * <0 = fatal (HIP error / OOM / violated invariant) — the Rust side maps <0 onto its existing
 * `error!` + `exit(SOFTWARE)` path (src/main.rs:401-405).
 */
#ifndef HIPHASE_GPU_H
#define HIPHASE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
#define HP_OK                 0
#define HP_WFA_MAX_ED         1   /* per-job: WFAGraphError::MaxEditDistance (wfa_graph.rs:13-17,645-648) */
#define HP_WFA_UNSUPPORTED    3   /* per-job: the graph-WFA job lies outside what the device kernels hold (max_edit_distance > 60 000, a node
                                     with more than 64 parents, a diagonal band wider than 65 535, more than 16 GiB of wavefront state for one
                                     read): nothing is computed for it; the other jobs of the call are. Through hp_solve_blocks a block with such
                                     a record comes back as HP_BLOCK_UNSUPPORTED */
#define HP_BLOCK_UNSUPPORTED  2   /* per-block (hp_block_output.status): outside the device solver's packed-key limits;
                                     the other blocks of the call are solved, the caller runs its own solve_block for this one */
#define HP_ERR_HIP           -1   /* HIP runtime error / no device / kernel image missing */
#define HP_ERR_OOM           -2   /* host or device allocation failed */
#define HP_ERR_INVARIANT     -3   /* an `assert!`/`panic!` of the reference would have fired */
#define HP_ERR_ARG           -4   /* malformed view (row_off not monotone, N==0, ...) */
#define HP_ERR_UNSUPPORTED   -5   /* outside the packed-key limits documented in DESIGN.md */

/* ---- read sequence encodings ---------------------------------------------------------------------------------------
 * HP_SEQ_ASCII: one byte per base, what `read.seq().as_bytes()` returns (read_parsing.rs:738).
 * HP_SEQ_BAM4:  the BAM record's own 4-bit encoding, what rust-htslib's `read.seq().encoded` points at: base k sits in byte
 *               k / 2, the HIGH nibble for even k, code -> base "=ACMGRSVTWYHKDBN". The caller hands the record's bytes over as
 *               they are (no decode on the host, half the bytes across PCIe); the library expands them on the device with exactly
 *               that table, so every comparison sees the bytes `as_bytes()` would have produced. */
#define HP_SEQ_ASCII 0
#define HP_SEQ_BAM4  1

/* ---- AlleleType (src/data_types/read_segments.rs:5-16) ----------------------------------- */
#define HP_ALLELE_REFERENCE 0
#define HP_ALLELE_ALTERNATE 1
#define HP_ALLELE_AMBIGUOUS 2
#define HP_ALLELE_NOOVERLAP 3

/* ---- per-variant flags (what astar_solver reads from `Variant`, astar_phaser.rs:438,446,606) */
#define HP_VAR_IGNORED 0x1   /* Variant::is_ignored() */
#define HP_VAR_SNV     0x2   /* Variant::get_type() == VariantType::Snv */

/*
 * One phase block's read x variant allele matrix = the IntervalTree<usize, ReadSegment> that
 * phaser.rs:514-533 hands to astar_solver, flattened to CSR. Row r is ReadSegment r:
 *   region()  = [read_start[r], read_end[r])            (read_segments.rs:29,52)
 *   allele(i) = 2-bit code at cell  row_off[r] + (i - read_start[r])
 *   qual(i)   = quals[ row_off[r] + (i - read_start[r]) ]
 * Cell c of alleles_2bit lives in byte c/4, bits [2*(c%4), 2*(c%4)+2).
 * Rows may come in any order (the library sorts by start). A row with start==end is legal and inert.
 */
typedef struct hp_block_view {
    uint32_t        n_variants;    /* N  = variants.len()  (astar_phaser.rs:431) */
    uint32_t        n_reads;       /* R  = rows in the solver tree */
    const uint32_t* read_start;    /* [R] */
    const uint32_t* read_end;      /* [R] exclusive */
    const uint64_t* row_off;       /* [R+1] cell offsets, row_off[r+1]-row_off[r] == end-start */
    const uint8_t*  alleles_2bit;  /* ceil(row_off[R]/4) bytes */
    const uint8_t*  quals;         /* row_off[R] bytes */
    const uint8_t*  var_flags;     /* [N] HP_VAR_* */
} hp_block_view;

/* Solver parameters (cli.rs:217,224; astar_phaser.rs:466). */
typedef struct hp_astar_params {
    uint64_t min_queue_size;    /* --phase-min-queue-size, default 1000 */
    uint64_t queue_increment;   /* --phase-queue-increment, default 3   */
    uint64_t max_segment_size;  /* hard-coded 40 in the reference (astar_phaser.rs:466); 0 => 40 */
    uint64_t block_index;       /* PhaseBlock::get_block_index(), diagnostics only */
} hp_astar_params;

/* PhaseStats as filled by astar_solver (astar_phaser.rs:599-621, phase_stats.rs:131-173). */
typedef struct hp_phase_stats {
    uint64_t pruned_solutions;
    uint64_t estimated_cost;
    uint64_t actual_cost;
    uint64_t phased_variants;
    uint64_t phased_snvs;
    uint64_t homozygous_variants;
    uint64_t skipped_variants;
} hp_phase_stats;

/* Work counters: the roofline numerator of SURVEY.md §8(d). Deterministic properties of the input
 * (the search trajectory is a total order), so oracle and device must agree exactly. */
typedef struct hp_work_counters {
    uint64_t sub_pops;      /* nodes popped by astar_subsolver over the whole heuristic chain */
    uint64_t main_pops;     /* nodes popped by the main search (incl. pruned ones) */
    uint64_t evals;         /* (child node, read) pairs scored in new_extended_node */
    uint64_t cells;         /* sum over evals of the overlap length l (cells touched per haplotype) */
    uint64_t nodes_created; /* children created (sub + main) */
    uint64_t reserved[3];
} hp_work_counters;

/* ---- A* MEC solver ------------------------------------------------------------------------ */

/* Replaces astar_phaser::astar_solver (astar_phaser.rs:426-633) for one block.
 * h1/h2: N bytes each, AlleleType codes 0/1/2. Runs on device `hp_default_device()`. */
int hp_astar_solve(const hp_block_view* blk, const hp_astar_params* p,
                   uint8_t* h1, uint8_t* h2, hp_phase_stats* out);

/* Batch form used by the multi-GPU block queue: n_blocks independent blocks, one params struct
 * shared by all (HiPhase passes the same CLI values to every block, main.rs:385-399).
 * device_id >= 0: that device; device_id == -1: shard over every visible device with a host-side
 * LPT work queue (one worker thread per device, no collective). h1[i]/h2[i] -> N_i bytes. */
int hp_astar_solve_batch(size_t n_blocks, const hp_block_view* blks, const hp_astar_params* p,
                         uint8_t* const* h1, uint8_t* const* h2, hp_phase_stats* out, int device_id);

/* Resident form: pack + upload once, solve many times (what bench.py times: inputs already in HBM). */
typedef struct hp_batch hp_batch;
hp_batch* hp_batch_create(size_t n_blocks, const hp_block_view* blks, const hp_astar_params* p,
                          int device_id, int* status);
/* Launches the solve on `stream` (a hipStream_t passed as void*; NULL = the batch's own stream) and
 * waits for it. kernel_ms (may be NULL) receives the HIP-event time of the solve kernel(s). */
int  hp_batch_solve(hp_batch* b, void* stream, float* kernel_ms);
/* Copies results of the last solve to host. Any pointer may be NULL. h1/h2: concatenated over blocks
 * (sum N_i bytes); stats/counters: [n_blocks]. */
int  hp_batch_results(hp_batch* b, uint8_t* h1, uint8_t* h2, hp_phase_stats* stats,
                      hp_work_counters* counters, uint64_t* heuristics /* sum (N_i+1) or NULL */);
/* Post-processing on the resident matrix after a successful hp_batch_solve (reference src/phaser.rs:350-388
 * get_solution_span_counts and :714-750 haplotag_reads). Any pointer may be NULL.
 *   span_counts: concatenated over blocks, N_i - 1 junctures each;
 *   haplotag / first_het: concatenated over blocks in the CALLER's row order (R_i each): haplotag 0 / 1, or 2 =
 *   untagged (tie); first_het = block-local index of the first het the row resolves (UINT32_MAX when untagged).
 *   The caller looks block_tags[first_het] up itself (phaser.rs:740) — tags need variant positions. */
int  hp_batch_postprocess(hp_batch* b, uint64_t* span_counts, uint8_t* haplotag, uint32_t* first_het);
void hp_batch_destroy(hp_batch* b);

/* ---- graph-WFA allele assignment ---------------------------------------------------------- */

/* One variant as WFAGraph::from_reference_variants_with_hom reads it (wfa_graph.rs:146-251):
 * position(), get_ref_len(), get_truncated_allele0/1(), convert_index(Reference) != 0, is_ignored(). */
typedef struct hp_wfa_variant {
    int64_t        position;        /* 0-based, chromosome coordinates */
    uint32_t       ref_len;
    uint32_t       flags;           /* bit0: ignored; bit1: allele0 is itself an ALT (index_allele0 != 0) */
    const uint8_t* allele0;         /* truncated allele0 (variants.rs:581-585); unused unless flags&2 */
    uint32_t       allele0_len;
    uint32_t       allele1_len;
    const uint8_t* allele1;         /* truncated allele1 (variants.rs:587-591) */
} hp_wfa_variant;

/* One BAM record's re-alignment job (read_parsing.rs:738-780). `reference` points at the chromosome
 * sequence such that reference[x - ref_base] is base x for x in [ref_start, ref_end). */
typedef struct hp_wfa_job {
    const uint8_t*        reference;
    uint64_t              ref_base;      /* chromosome coordinate of reference[0] */
    uint64_t              ref_start;     /* min_position (inclusive) */
    uint64_t              ref_end;       /* max_position + 1 (exclusive) */
    const hp_wfa_variant* hets;          /* variant_calls[first_overlap..last_overlap) */
    uint32_t              n_hets;
    uint32_t              n_homs;
    const hp_wfa_variant* homs;          /* hom_calls[first_hom_overlap..last_hom_overlap) */
    const uint8_t*        read;          /* read_align = seq[read_start..=read_end] */
    uint32_t              read_len;
    uint32_t              reserved;
} hp_wfa_job;

typedef struct hp_wfa_result {
    int32_t  status;       /* HP_OK, HP_WFA_MAX_ED or HP_WFA_UNSUPPORTED */
    uint32_t n_nodes;      /* WFAGraph::get_num_nodes() */
    uint64_t score;        /* WFAResult::score(); max_edit_distance when status == HP_WFA_MAX_ED */
} hp_wfa_result;

/* Replaces graph build + edit_distance_with_pruning + the node->allele mapping of
 * read_parsing.rs:790-800 for n jobs. alleles[i] must hold jobs[i].n_hets bytes and receives one
 * AlleleType per het of the job (NoOverlap / Reference / Alternate / Ambiguous).
 * prune_distance: GlobalRealignmentConfig::wfa_prune_distance (UINT64_MAX disables pruning).
 * max_ed: GlobalRealignmentConfig::max_edit_distance.
 * The jobs of a block normally point at ONE reference buffer (the chromosome) with their own windows: the
 * library uploads the union of the windows' address ranges once. Every pointer stays owned by the caller and
 * must remain valid (and unmodified) until the call returns; nothing is retained afterwards. */
int hp_wfa_assign_batch(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed,
                        hp_wfa_result* out, uint8_t* const* alleles, int device_id);

/* Caller-built graphs: WFAGraph::new + add_node (src/wfa_graph.rs:100-117, 298-331: node k is added with its sequence and
 * the indices of its parents, all of them earlier nodes; node 0 has none) and edit_distance_with_pruning (:350-650) ->
 * WFAResult { score, traversed_nodes } (:654-670). This is the layer under from_reference_variants_with_hom that the
 * reference's own tests drive with hand-built topologies (:677-839). traversed[i], when not NULL, receives the
 * traversed-node bitset of job i, (n_nodes + 31) / 32 words (bit k = node k lies on a best alignment); status is HP_OK or
 * HP_WFA_MAX_ED (score = max_ed, nothing traversed). An add_node assert maps to HP_ERR_INVARIANT. */
typedef struct hp_graph_node { const uint8_t* seq; uint32_t seq_len; uint32_t n_parents; const uint32_t* parents; } hp_graph_node;
typedef struct hp_graph_job { const hp_graph_node* nodes; uint32_t n_nodes; uint32_t read_len; const uint8_t* read; } hp_graph_job;
typedef struct hp_graph_result { int32_t status; uint32_t n_traversed; uint64_t score; } hp_graph_result;
int hp_wfa_align_graphs(const hp_graph_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed, hp_graph_result* out,
                        uint32_t* const* traversed, int device_id);

/* ---- Levenshtein (sequence_alignment.rs:7-38) --------------------------------------------- */
typedef struct hp_ed_pair { const uint8_t* a; const uint8_t* b; uint32_t a_len; uint32_t b_len; } hp_ed_pair;
int hp_edit_distance_batch(const hp_ed_pair* pairs, size_t n, uint64_t* out, int device_id);

/* ---- local re-alignment (read_parsing.rs:121-503) ------------------------------------------ */
/* Replaces `local_realignment(read, variant_calls)` for a batch of BAM records of one block: the per-variant
 * coordinate logic, exact matching and quality scaling run on the host exactly as the reference runs them; every
 * inexact allele (Variant::closest_allele_clip, variants.rs:624-641) of the whole batch goes to the device in ONE
 * Levenshtein launch. Used by `load_read_segments` (read_parsing.rs:47-113, --disable-global-realignment) and
 * by the fallback branch of `load_full_read_segments` (read_parsing.rs:564-575). */
typedef struct hp_local_variant {        /* what local_realignment reads from a `Variant` (variants.rs:67-94) */
    int64_t        position;             /* Variant::position() */
    uint32_t       ref_len;              /* get_ref_len() */
    uint32_t       variant_type;         /* VariantType repr, variants.rs:10-33 (Snv 0 ... Unknown 10) */
    uint32_t       prefix_len;           /* get_prefix_len() */
    uint32_t       postfix_len;          /* get_postfix_len() */
    const uint8_t* allele0;              /* get_allele0(): prefix + allele + postfix */
    const uint8_t* allele1;              /* get_allele1() */
    uint32_t       allele0_len;
    uint32_t       allele1_len;
    uint32_t       flags;                /* HP_VAR_IGNORED */
    uint32_t       reserved;
} hp_local_variant;

typedef struct hp_local_read {           /* what local_realignment reads from a bam::Record */
    int64_t         pos;                 /* read.pos() */
    const uint32_t* cigar;               /* BAM encoding: op_len << 4 | op (MIDNSHP=X = 0..8) */
    uint32_t        n_cigar;
    uint32_t        seq_len;             /* bases */
    const uint8_t*  seq;                 /* HP_SEQ_ASCII: read.seq().as_bytes(); HP_SEQ_BAM4: read.seq().encoded (base k in nibble k) */
    const uint8_t*  qual;                /* read.qual() */
    uint32_t        seq_format;          /* HP_SEQ_* */
    uint32_t        reserved;
} hp_local_read;

#define HP_N_VARIANT_TYPES 11            /* VariantType::Unknown as usize + 1 */
typedef struct hp_read_stats {           /* ReadStats of one record (writers/phase_stats.rs:12-33) */
    uint64_t skipped_reads;              /* 1 when no allele was determined */
    uint64_t num_alleles;
    uint64_t exact_matches[HP_N_VARIANT_TYPES];
    uint64_t inexact_matches[HP_N_VARIANT_TYPES];
    uint64_t failed_matches[HP_N_VARIANT_TYPES];
    uint64_t allele0_matches[HP_N_VARIANT_TYPES];
    uint64_t allele1_matches[HP_N_VARIANT_TYPES];
    uint64_t local_aligned;
} hp_read_stats;

/* alleles / quals: n_reads x n_variants bytes, row-major (row r = the Vec<AlleleType> / Vec<u8> the reference
 * returns for reads[r]). stats: n_reads entries or NULL. A CIGAR with a Pad op returns HP_ERR_UNSUPPORTED (the
 * reference's rust-htslib 0.39.5 `aligned_pairs` panics on it); a variant type the reference has no
 * implementation for (read_parsing.rs:317,455 panic) returns HP_ERR_INVARIANT. */
int hp_local_realign_batch(const hp_local_read* reads, size_t n_reads, const hp_local_variant* variants,
                           size_t n_variants, uint8_t* alleles, uint8_t* quals, hp_read_stats* stats, int device_id);

/* ---- whole blocks: phaser::solve_block from the decoded records on (phaser.rs:513-630) ------------------------------- */
/* One entry runs, inside the library and for any number of blocks at once: graph-WFA for every record with overlaps
 * (read_parsing.rs:652-800, one device batch over ALL blocks), local re-alignment of the records whose WFA ran into
 * max_edit_distance and the order-dependent `global_disabled` switch (read_parsing.rs:556-600), quality assignment
 * (:803-835), ReadSegment::new, collapse per read name, the min_matched_alleles split (:611-629), the A* solver
 * (phaser.rs:541-543), get_solution_span_counts (:546) and haplotag_reads (:614-630). What stays with the caller is
 * I/O: BAM decoding and filtering (filter_out_alignment_record, aligned_pairs -> min/max position), VCF loading,
 * the PS tag lookup block_tags[first_het] and the sub-block split, which need variant positions the caller owns. */
typedef struct hp_block_record {          /* one BAM record that passed filter_out_alignment_record (read_parsing.rs:551) */
    int64_t              min_position;    /* first / last reference base of the alignment, inclusive (read_parsing.rs:672-685) */
    int64_t              max_position;
    const uint8_t*       read_align;      /* seq[read_start ..= read_end] (read_parsing.rs:738-742): base k of it is base
                                             read_offset + k of this buffer, in the block's seq_format (HP_SEQ_BAM4: pass
                                             read.seq().encoded and read_offset = read_start) */
    uint32_t             read_len;        /* bases */
    uint32_t             qname_id;        /* records of one read name share an id: 0 .. n_qnames-1 */
    const hp_local_read* local;           /* CIGAR view of the record for local re-alignment: needed in local mode and when the
                                             record falls back (read_parsing.rs:556-575); NULL makes such a fallback an HP_ERR_ARG */
    uint32_t             read_offset;     /* bases to skip at read_align (0 for a buffer that starts at read_start) */
    uint32_t             reserved;
} hp_block_record;

typedef struct hp_block_input {
    uint64_t                block_index;  /* PhaseBlock::get_block_index() */
    const uint8_t*          reference;    /* chromosome bytes: reference[x - ref_base] is base x */
    uint64_t                ref_base;
    uint32_t                n_hets, n_homs, n_records, n_qnames;
    const hp_wfa_variant*   hets;         /* variant_calls as the graph builder reads them, position-sorted [n_hets] */
    const uint8_t*          het_types;    /* VariantType of each het (variants.rs:10-33) [n_hets] */
    const hp_local_variant* local_hets;   /* variant_calls as local_realignment reads them [n_hets]; may be NULL when no record
                                             can fall back (then a fallback is an HP_ERR_ARG) */
    const hp_wfa_variant*   homs;         /* hom_calls [n_homs] */
    const hp_block_record*  records;      /* in BAM order: the fallback switch depends on it (read_parsing.rs:597-600) */
    uint32_t                seq_format;   /* HP_SEQ_* of every records[i].read_align of this block */
    uint32_t                reserved;
} hp_block_input;

typedef struct hp_block_params {
    hp_astar_params astar;
    uint64_t wfa_prune_distance;          /* GlobalRealignmentConfig (cli.rs:189-210); UINT64_MAX disables pruning */
    uint64_t max_edit_distance;
    double   global_failure_ratio;
    uint64_t global_failure_minimum;
    uint64_t min_matched_alleles;         /* cli.rs:145 */
    uint32_t global_realignment;          /* 1: load_full_read_segments, 0: load_read_segments (--disable-global-realignment) */
    uint32_t reserved;
} hp_block_params;

/* Everything is caller-allocated; any array pointer may be NULL (then that output is skipped). Segments = the collapsed
 * ReadSegments with at least one set allele, in first-seen read-name order; seg_solver marks the ones that entered
 * the solver (get_num_set() >= min_matched_alleles), the others are "phasable only" (read_parsing.rs:620-627). */
typedef struct hp_block_output {
    uint8_t*        h1;                   /* [n_hets] AstarResult haplotypes */
    uint8_t*        h2;
    hp_phase_stats  stats;
    uint64_t*       span_counts;          /* [n_hets - 1] */
    uint32_t        n_segments;           /* out */
    uint32_t        n_solver;             /* out */
    uint32_t*       seg_qname;            /* [n_qnames] */
    uint32_t*       seg_start;            /* [n_qnames] region().start */
    uint32_t*       seg_end;              /* [n_qnames] region().end */
    uint8_t*        seg_solver;           /* [n_qnames] */
    uint8_t*        seg_haplotag;         /* [n_qnames] 0 / 1, 2 = untagged (tie) */
    uint32_t*       seg_first_het;        /* [n_qnames] first het the segment resolves, UINT32_MAX when untagged */
    uint64_t*       seg_row_off;          /* [n_qnames + 1] cell offsets of the segments' rows */
    uint8_t*        seg_alleles;          /* AlleleType per cell */
    uint8_t*        seg_quals;
    uint64_t        seg_cell_cap;         /* in: capacity of seg_alleles / seg_quals (HP_ERR_ARG when too small) */
    uint64_t        num_reads;            /* ReadStats of the loader: records in solver segments */
    uint64_t        skipped_reads;        /* records without overlaps, skipped by local re-alignment, or in dropped segments */
    uint64_t        global_aligned;
    uint64_t        local_aligned;
    uint64_t*       edit_distances;       /* [n_records] wfa_score of every record that was not skipped, in BAM order */
    uint64_t        n_edit_distances;     /* out */
    int32_t         status;               /* out: HP_OK, or HP_BLOCK_UNSUPPORTED (h1/h2/stats/spans/tags are not filled; the segments are,
                                             unless it was a record's graph-WFA job that fell outside the device limits: then n_segments = 0) */
    uint32_t        reserved;
    /* The rest of the loader's ReadStats (writers/phase_stats.rs:12-33; returned at phaser.rs:645, written at phase_stats.rs:288):
     * `joint_stats += read_stats` over EVERY record of the block, skipped or not, BEFORE the collapse (read_parsing.rs:88, :607).
     * A globally re-aligned record counts per het of its overlap range (read_parsing.rs:803-850): Ambiguous -> failed_matches,
     * Reference / Alternate -> inexact_matches (`exact_allele` is false upstream) + allele0 / allele1_matches + num_alleles; a
     * record that fell back (or every record in local mode) brings local_realignment's own counts (read_parsing.rs:121-503). */
    uint64_t        num_alleles;
    uint64_t        exact_matches[HP_N_VARIANT_TYPES];
    uint64_t        inexact_matches[HP_N_VARIANT_TYPES];
    uint64_t        failed_matches[HP_N_VARIANT_TYPES];
    uint64_t        allele0_matches[HP_N_VARIANT_TYPES];
    uint64_t        allele1_matches[HP_N_VARIANT_TYPES];
} hp_block_output;

int hp_solve_blocks(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out, int device_id);

/* Resident form: hp_blockset_create lays the blocks' sequences out and uploads them (the caller's buffers must stay
 * valid until hp_blockset_destroy); hp_blockset_solve runs the whole path on the resident data and may be called
 * repeatedly, from any thread (bench.py's secondary, inputs-in-HBM figure). stage_ms (may be NULL) receives 8 numbers in ms: [0] graph-WFA stage wall time
 * (device graph build + alignment + allele rows + download), [1] fallback / replay / row assembly (host), [2] A* pack +
 * upload, [3] A* solve, [4] post-processing, [5] total, [6] graph-WFA kernels (HIP events), [7] A* kernel (HIP events). */
typedef struct hp_blockset hp_blockset;
hp_blockset* hp_blockset_create(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, int device_id, int* status);
int  hp_blockset_solve(hp_blockset* bs, hp_block_output* out, double* stage_ms);
/* Work counters of the last hp_blockset_solve (the roofline numerators of SURVEY.md 8d): out[0] reads aligned by the compact
 * graph-WFA kernel, [1] their read bases, [2] bytes of the graph nodes their best alignments traverse, [3] their (node,
 * diagonal) wave updates; [4] A* cells, [5] A* read evaluations; [6] hets, [7] solver rows. */
int  hp_blockset_work(const hp_blockset* bs, uint64_t out[8]);
void hp_blockset_destroy(hp_blockset* bs);

/* Pipelined form: HiPhase sees every block once (src/main.rs:337-408; src/phaser.rs:513-543 loads a block's reads, then solves it),
 * so a caller that hands over one block set after the other wants set k + 2 laid out and crossing PCIe while set k + 1 is aligned
 * and set k is solved. hp_blockstream_submit queues a set and returns its ticket at once (it blocks only while `depth` sets are in
 * flight: the back-pressure of the reference's bounded job queue, main.rs:362-383); hp_blockstream_wait returns when that set's
 * results are in its `out` array. Sets complete in submission order; results are identical to hp_solve_blocks on the same set.
 * `in`, everything it points at, and `out` must stay valid until the wait for that ticket returns. Submit and wait may be called
 * from different threads. depth: 0 = 5. stage_ms (16 doubles, may be NULL): [0] overlaps + layout (host), [1] staging copy + PCIe
 * + base expansion, [2] graph-WFA stage, [3] fallback / replay / rows (host), [4] A* pack + upload, [5] A* solve, [6] post-processing
 * + outputs, [7] latency submit -> done, [8] graph-WFA kernels (HIP events), [9] A* kernels (HIP events), [10] bytes host -> device,
 * [11] time spent waiting between stages, [12..15] wall time of stage 1 (layout + PCIe) / 2 (graph-WFA) / 3 (rows) / 4 + 5 (A* pack; A* + post). work (8 values, may be NULL): as hp_blockset_work.
 * device_id >= 0: one six-stage pipeline on that device. device_id == -1: one pipeline per visible device behind the same submit /
 * wait - a set goes to the pipeline with the fewest records in flight, submit blocks only while every pipeline holds `depth` sets,
 * the sets of one device complete in order (independent blocks: no exchange between devices; the reference's fan-out is
 * main.rs:332-408, its writers re-order by block index, writers/ordered_vcf_writer.rs:158-170). hp_blockstream_devices: pipelines. */
typedef struct hp_blockstream hp_blockstream;
hp_blockstream* hp_blockstream_create(const hp_block_params* p, int device_id, uint32_t depth, int* status);
int  hp_blockstream_devices(const hp_blockstream* s);
int  hp_blockstream_submit(hp_blockstream* s, size_t n_blocks, const hp_block_input* in, hp_block_output* out, uint64_t* ticket);
int  hp_blockstream_wait(hp_blockstream* s, uint64_t ticket, double* stage_ms, uint64_t* work);
void hp_blockstream_destroy(hp_blockstream* s);   /* finishes the sets still in flight first */

/* Asynchronous per-block entry. HiPhase keeps `job_slots = 40 x threads` phase blocks queued but only `threads` calls of
 * solve_block in flight (src/main.rs:328,344-383); a worker that SUBMITS its block and goes on to load the next one keeps all of
 * them in flight on the device side, where they are merged into block sets that travel through the same per-device pipelines as
 * hp_blockstream_* (INTEGRATION.md 3c has the ~30-line main.rs patch). hp_block_submit queues n_blocks blocks (usually 1) and
 * returns a ticket at once; hp_block_wait returns when their results are in `out` (same results and statuses as hp_solve_blocks)
 * and consumes the ticket. `in`, everything it points at, and `out` must stay valid until the wait returns; any thread may wait.
 * device_id: -1 = any visible device, >= 0 = that one. The blocking call hp_solve_blocks(1, ...) is submit + wait. */
int hp_block_submit(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out, int device_id, uint64_t* ticket);
int hp_block_wait(uint64_t ticket);

/* ---- misc --------------------------------------------------------------------------------- */
int         hp_device_count(void);
int         hp_default_device(void);
const char* hp_last_error(void);        /* thread-local, never NULL */
const char* hp_version(void);
/* Calls of hp_astar_solve / hp_wfa_assign_batch / hp_solve_blocks that are in flight at the same time (HiPhase's worker pool
 * calls solve_block once per block, main.rs:385-408) are merged into one device batch behind the unchanged signatures;
 * results are identical either way. 0 turns the merging off (also: HP_COALESCE=0), returns the previous setting. */
int         hp_set_coalescing(int on);
/* HIP-event time (ms) of the kernel(s) launched by the last hp_wfa_assign_batch / hp_edit_distance_batch /
 * hp_astar_solve* call made on this thread (diagnostics for bench/roofline reporting). Only meaningful for calls that ran
 * on the calling thread: a call merged with other callers' (coalescing on, the default) or handed to the multi-device
 * dispatcher runs on a service thread and leaves this thread's value as it was - turn coalescing off
 * (hp_set_coalescing(0)) around a measurement, or use the stage times hp_blockstream_wait / hp_blockset_solve return. */
double      hp_last_kernel_ms(void);
/* How the library's host threads wait for the device: -1 not decided yet (no entry point has touched a device), 1 blocking
 * (hipDeviceScheduleBlockingSync on every device: set by the library on the devices nothing in the process had used, or found set on
 * the ones the host had initialised), 2 on some of them, 0 on none (the host initialised the devices in another mode - they keep it:
 * switching a device that has streams loses completions, DESIGN.md 5 - or HP_BLOCKING_SYNC=0). */
int         hp_runtime_wait_mode(void);
/* The library keeps device buffers it has let go of in a process-wide cache (hipMalloc / hipFree wait for every kernel on the
 * device): at most 2/9 of the device's memory (HP_DEV_CACHE_GB overrides). hp_trim_device_cache frees all of it - for a host
 * application that shares the GPU with another allocator - and returns the bytes handed back. */
size_t      hp_trim_device_cache(void);
/* Host memory the device reads IN PLACE. A block set whose records' bases (hp_block_record.read_align) all lie in memory from
 * hp_host_alloc is not staged: no copy into the library's pinned staging on host threads (1 GB per 60 000 hets at 30x - most of
 * the library's host CPU time); the copy engines take the blocks' address ranges as they are, one DMA per run of neighbouring
 * blocks. The natural caller-side arena: decode / gather a block's records into it one after the other, hand the block over,
 * reuse it once the call (or the wait) has returned. Records scattered so that the blocks' address hulls hold over 1.25 x the
 * bytes the records do are staged as ever. 64 bytes of slack behind the request (the device reads 16 bytes at a time). NULL
 * when the allocation fails; hp_host_free(NULL) is a no-op. */
void*       hp_host_alloc(size_t bytes);
void        hp_host_free(void* p);
/* Bytes of such memory the copy engines have read in place since the library was loaded (0 = every set so far was staged: a
 * loader's check that its arena is the one the library sees). */
uint64_t    hp_host_in_place_bytes(void);
/* Records of block sets that took their way out of the alignment stage EARLY since the library was loaded (an experiment switch,
 * HP_WFA2_ROUTE=1|2; 0 = never, the default: it measured slower): a record whose CIGAR (hp_block_record.local) begins
 * an operation every 20 bases or less is heading for the neighbourhood of max_edit_distance (read_parsing.rs:564-575:
 * Err(MaxEditDistance) -> local re-alignment, or an alignment of several hundred edits) and can be routed past the
 * several-reads-per-wavefront kernels to the reference-window test and the one-read-per-wavefront kernel, beside the set's launch
 * set instead of behind it. Routing only: the results are the same on every road. */
uint64_t    hp_wfa_routed_records(void);
/* Appends one block in the .hpbk capture format (hiphase_amd/block_io.py) to `path`: the solver's exact
 * input and - when h1, h2 and stats are given - the output the caller's own astar_solver produced for it. A HiPhase
 * built with this call at src/phaser.rs:541-543 writes the real HG002 blocks this repository cannot produce. */
int         hp_hpbk_append(const char* path, const hp_block_view* blk, const hp_astar_params* p, const uint8_t* h1,
                           const uint8_t* h2, const hp_phase_stats* stats);
/* JSON: sizeof / alignof / offsetof of every struct in this header as the library was compiled (generated by
 * scripts/gen_abi_layout.py) - diff the #[repr(C)] side of a binding against it once at start-up. */
const char* hp_abi_layout(void);
/* The same numbers one at a time, for a binding without a JSON parser (patches/0001: the Rust side checks every #[repr(C)] struct it
 * declares once at start-up): (size_t)-1 for a name the library does not know. */
size_t      hp_abi_sizeof(const char* struct_name);
size_t      hp_abi_offsetof(const char* struct_name, const char* field_name);

/* Deterministic synthetic block generator of SURVEY.md §8(d) (splitmix64). Fills caller-provided
 * buffers sized via hp_synth_block_size(). Used by tests and bench.py on both legs. */
typedef struct hp_synth_spec {
    uint32_t n_variants;  /* N */
    uint32_t coverage;    /* C */
    uint32_t span;        /* S */
    uint32_t reserved;
    double   error_rate;  /* e */
    double   ambig_rate;  /* a */
    uint64_t seed;
} hp_synth_spec;
/* returns R; *n_cells = upper bound on cells (R * min(N, ceil(1.5*S)+1)) */
uint32_t hp_synth_block_size(const hp_synth_spec* s, uint64_t* n_cells);
/* returns 0; fills arrays (read_start/end [R], row_off [R+1], alleles_2bit, quals, var_flags [N],
 * truth [N] may be NULL). Rows that end up with no set allele keep start==end (inert). */
int hp_synth_block(const hp_synth_spec* s, uint32_t* read_start, uint32_t* read_end, uint64_t* row_off,
                   uint8_t* alleles_2bit, uint8_t* quals, uint8_t* var_flags, uint8_t* truth);

/* Deterministic synthetic READ-BEARING block sets (hp_synth_reads.cpp): the whole-path workload of bench.py and of the block-level
 * tests, produced in the C layout hp_solve_blocks / hp_blockstream_submit take - what BASELINE.json configs[2-4] would decode
 * from a BAM + VCF + FASTA. Per block: a random reference; het + hom calls in the SURVEY.md 8(d) type mix (the rest after
 * frac_snv + frac_indel + frac_sv is tandem repeats; frac_multiallelic of the het tandem repeats are 1|2 genotypes); HiFi-like
 * reads carrying one haplotype with uniform edit noise; a noisy tail (noisy_fraction of the reads at noisy_noise: they exceed
 * max_edit_distance and fall back to local re-alignment); supplementary_fraction of the reads as two records of one read name.
 * Block sizes are lognormal (median 15 hets) capped at max_block_hets (docs/user_guide.md:257-260). A pure function of the spec. */
typedef struct hp_synth_reads_spec {
    uint64_t seed;
    uint32_t total_hets;        /* blocks are drawn until this many hets are reached */
    uint32_t max_block_hets;
    double   coverage, read_mean, read_sd, het_spacing, hom_ratio;
    double   frac_snv, frac_indel, frac_sv, frac_multiallelic;
    double   edit_noise, noisy_fraction, noisy_noise, supplementary_fraction;
    uint32_t seq_format;        /* HP_SEQ_* of the records' bases */
    uint32_t threads;           /* host threads for the generation; 0 = up to 32 */
    /* HiFi-shaped errors (round 5). hifi_sigma > 0: a record's error rate is drawn per READ from a lognormal with median
     * `edit_noise` and this sigma (clamped to [edit_noise / 20, 0.04]): median 0.2 %, sigma 0.8 puts 2 % of the reads beyond 1 %
     * and 0.2 % beyond 2 % - the shape of a HiFi run's per-read accuracy (median Q27-30, a tail at Q20). homopolymer_share > 0:
     * that share of the errors are insertions / deletions of the run's base inside homopolymer runs (the dominant HiFi error),
     * the rest uniform substitutions / insertions / deletions as before. Both 0 (hp_synth_reads_defaults): the uniform model. */
    double   hifi_sigma, homopolymer_share;
    /* Wrong-haplotype alleles (round 6: the deep-coverage frontier stress of BASELINE.json configs[4]). allele_switch > 0: at every
     * het a record spans it carries the OTHER haplotype's allele with this probability (a miscalled / chimeric cell): the allele
     * matrix then holds conflicting rows, the A* frontier grows and its pruning (astar_phaser.rs:564-585) gets to work - at 0.15 and
     * 60x every large block prunes. 0 (the defaults): reads follow their haplotype, sets are byte-for-byte what they were. */
    double   allele_switch;
} hp_synth_reads_spec;
typedef struct hp_synth_set hp_synth_set;
void hp_synth_reads_defaults(hp_synth_reads_spec* s);   /* the bench workload: 60 000 hets, 30x, 15 kb reads, 0.5 % edit noise, BAM 4-bit */
void hp_synth_reads_deep60(hp_synth_reads_spec* s);     /* configs[4]'s shape: 60x, 15 % wrong-haplotype cells, every tandem-repeat het multi-allelic (22 % of the hets), blocks up to the 4 165-het cap, 20 000 hets */
void hp_synth_reads_hifi(hp_synth_reads_spec* s);       /* the same with HiFi-shaped errors: per-read rate lognormal (median 0.2 %, sigma 0.8), half of the errors homopolymer indels, no separate noisy class */
hp_synth_set* hp_synth_reads_create(const hp_synth_reads_spec* s, int* status);
const hp_block_input* hp_synth_reads_inputs(const hp_synth_set* s, size_t* n_blocks);
/* out[0] blocks, [1] hets, [2] records, [3] read bases, [4] read names, [5] bytes of reads + references as handed over, [6] largest block (hets) */
void hp_synth_reads_info(const hp_synth_set* s, uint64_t out[8]);
const uint8_t* hp_synth_reads_truth(const hp_synth_set* s, size_t block);   /* [n_hets] the allele haplotype 0 carries */
void hp_synth_reads_destroy(hp_synth_set* s);
/* Moves every record's bases of the set into ONE buffer from `alloc` (and frees it with `dealloc` when the set is destroyed) - with
 * hp_host_alloc / hp_host_free: the set as a caller holds it that keeps its records in memory the device reads in place. */
int hp_synth_reads_relocate(hp_synth_set* s, void* (*alloc)(size_t), void (*dealloc)(void*));

/* ---- caller-side helpers around the block entries (hp_capture.cpp; host-only) ------------------------------------------------- */
/* Output buffers for n blocks, every array of hp_block_output allocated and sized from the inputs (seg_cell_cap: per read name
 * the hull of its records' het ranges). */
typedef struct hp_outputs hp_outputs;
hp_outputs* hp_outputs_create(const hp_block_input* in, size_t n);
hp_block_output* hp_outputs_array(hp_outputs* o);
void hp_outputs_destroy(hp_outputs* o);
/* every array (to its capacity) and every scalar result of every block filled with `fill`; pointers and capacities kept. Two sets
 * poisoned with different bytes compare equal only in what a solve really wrote (tests). */
void hp_outputs_poison(hp_outputs* o, uint8_t fill);
/* 1 when two outputs of the same block hold the same results in every field hp_solve_blocks fills, else 0 */
int hp_block_output_equal(const hp_block_input* in, const hp_block_output* a, const hp_block_output* b);
/* `.hpbr`: READ-BEARING capture of phase blocks - everything hp_solve_blocks reads for a block (reference hull, variant calls,
 * records with their bases and CIGAR views, parameters) and, when `expected` is given, the results the caller's own solve_block
 * produced for it. A HiPhase built with the patch of INTEGRATION.md 6 writes the real HG002 blocks this repository cannot
 * produce (BASELINE.json configs[2-4]); hp_hpbr_open loads a capture back as hp_block_input arrays (valid until hp_hpbr_close)
 * for hp_solve_blocks / hp_blockstream_submit / `bench.py --replay`. expected[i].status == INT32_MIN: no expected output stored. */
int hp_hpbr_append(const char* path, const hp_block_input* block, const hp_block_params* p, const hp_block_output* expected);
typedef struct hp_hpbr hp_hpbr;
hp_hpbr* hp_hpbr_open(const char* path, int* status);
const hp_block_input*  hp_hpbr_inputs(const hp_hpbr* h, size_t* n_blocks);
const hp_block_params* hp_hpbr_params(const hp_hpbr* h);      /* [n_blocks] */
const hp_block_output* hp_hpbr_expected(const hp_hpbr* h);    /* [n_blocks] */
void hp_hpbr_close(hp_hpbr* h);
const char* hp_hpbr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HIPHASE_GPU_H */
