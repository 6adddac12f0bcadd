// hp_wfa2_host.h — what hp_block.hip / hp_stream.hip see of the compact graph-WFA stage (hp_wfa2.hip).
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/hiphase_gpu.h"

namespace hp {

// One BAM record with overlaps, as a job of its block: the graph builder reads variant_calls[first_overlap..last_overlap) and
// hom_calls[first_hom_overlap..last_hom_overlap) (reference src/read_parsing.rs:688-730, 769-777), i.e. slices of the
// block's own vectors; the window is the record's [min_position, max_position + 1).
struct W2JobIn {
    uint32_t block;                 // index into the block array the session was prepared with
    uint32_t rec;                   // record index within the block
    uint32_t het_first, n_hets;     // block-local
    uint32_t hom_first, n_homs;
};

struct W2Session;
W2Session* w2_session_create();
void w2_session_destroy(W2Session* s);
// generic jobs (hp_wfa_assign_batch): layout by merging address ranges
int w2_session_prepare(W2Session* s, const hp_wfa_job* jobs, size_t n, int device_id);
// records of blocks: per-block layout (one reference hull, the block's variant vectors once), reads staged as the caller holds
// them (HP_SEQ_ASCII / HP_SEQ_BAM4) piece by piece while the previous piece crosses PCIe, expanded on the device
int w2_session_prepare_blocks(W2Session* s, const hp_block_input* in, size_t n_in, const W2JobIn* jobs, size_t n, int device_id);
// ... in two halves: the host-only layout (offsets, in-place runs, length order - no device call) and the fill + upload
int w2_session_layout_blocks(W2Session* s, const hp_block_input* in, size_t n_in, const W2JobIn* jobs, size_t n);
int w2_session_upload_blocks(W2Session* s, int device_id);
// prep[0] layout ms, [1] fill + upload ms, [2] total ms, [3] bytes host -> device
void w2_session_prepare_stats(const W2Session* s, double prep[4]);
int w2_session_run(W2Session* s, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* const* alleles, int defer);
int w2_session_finish(W2Session* s);
// after a run with defer = 2: waits until the first collection's results are on the host and hands them to the caller's arrays
// (on the calling thread's pool). Before it returns neither `out` nor w2_session_pending may be read. A no-op otherwise.
int w2_session_collected(W2Session* s);
void w2_session_pending(const W2Session* s, const uint32_t** ids, size_t* n);
void w2_session_work(const W2Session* s, uint64_t out[4]);
double w2_session_span_ms(const W2Session* s);
// the dense-band implementation (hp_wfa.hip): small batches and leftovers
int wfa_assign_batch_v1(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out,
                        uint8_t* const* alleles, int device_id);

// optional, per job of the next wfa_assign_batch_v1 call on this thread: an edit distance the job is known to exceed (it starts
// with a band that wide); nullptr = none
extern thread_local const uint32_t* g_wfa_min_ed_hint;

}  // namespace hp
