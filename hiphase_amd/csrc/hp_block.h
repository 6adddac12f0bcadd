// hp_block.h — state of a block set on its way through the path (hp_block.hip), shared with the pipelined form (hp_stream.hip).
#pragma once
#include "hp_common.h"
#include "hp_wfa2_host.h"

#include <memory>
#include <string>
#include <vector>

namespace hp {

// Cells of a block's segments: one bump allocator per block, kept from solve to solve (two heap allocations per record
// from 32 threads at once were most of the row stage)
struct Arena {
    std::vector<std::unique_ptr<uint8_t[]>> chunks;
    std::vector<size_t> caps;
    size_t cur = 0, used = 0;
    void reset() { cur = 0; used = 0; }
    uint8_t* get(size_t n) {
        while (cur < chunks.size() && used + n > caps[cur]) { ++cur; used = 0; }
        if (cur == chunks.size()) {
            const size_t c = std::max<size_t>(n, (size_t)1 << 16);
            chunks.emplace_back(new uint8_t[c]);
            caps.push_back(c);
        }
        uint8_t* p = chunks[cur].get() + used;
        used += n;
        return p;
    }
};
struct Segment {            // a ReadSegment: clipped row (read_segments.rs:19-62); end - start cells each in the block's arena
    uint32_t start = 0, end = 0;
    const uint8_t* alleles = nullptr;
    const uint8_t* quals = nullptr;
};
struct RecMeta { int64_t job = -1; uint32_t first = 0, last = 0; };

struct BlockState {         // per block, rebuilt by every solve
    std::vector<Segment> segs;            // collapsed, >= 1 set allele, first-seen read-name order
    std::vector<uint32_t> seg_qname;
    std::vector<uint8_t> seg_solver;
    std::vector<uint32_t> solver_rows;    // indices into segs
    uint64_t num_reads = 0, skipped_reads = 0, global_aligned = 0, local_aligned = 0;
    hp_read_stats rs{};                   // joint_stats += read_stats over every record (num_alleles and the five per-type arrays are used)
    std::vector<uint64_t> edit_distances;
    // the solver matrix as the C ABI takes it
    std::vector<uint32_t> read_start, read_end;
    std::vector<uint64_t> row_off;
    std::vector<uint8_t> alleles_2bit, quals, var_flags;
    // local re-alignment rows of the records that need them (fallbacks; every record in local mode): slot per record, -1 = none
    std::vector<int64_t> local_slot;
    std::vector<uint8_t> loc_alleles, loc_quals;
    std::vector<hp_read_stats> loc_stats;
    bool wfa_unsupported = false;         // a record's graph-WFA job lies outside the device kernels' limits: the block goes back to the caller
    std::vector<uint32_t> loc_need;       // records of the pre-pass (blockset_rows gathers them over all blocks: one device launch)
    std::vector<hp_local_read> loc_reads;
    Arena arena;
    void reset() {   // keeps every capacity
        segs.clear(); seg_qname.clear(); seg_solver.clear(); solver_rows.clear();
        num_reads = skipped_reads = global_aligned = local_aligned = 0;
        rs = hp_read_stats{};
        edit_distances.clear(); read_start.clear(); read_end.clear(); row_off.clear();
        alleles_2bit.clear(); quals.clear(); var_flags.clear();
        wfa_unsupported = false;
        local_slot.clear(); loc_alleles.clear(); loc_quals.clear(); loc_stats.clear(); loc_need.clear(); loc_reads.clear();
        arena.reset();
    }
};

}  // namespace hp

// One block set on its way through the path: ONE graph-WFA batch over the records of all its blocks, one resident A* batch.
// (Round 2 could split a set in two overlapped halves, HP_BLOCK_PIPELINE=1; measured three times, it lost every time - two
// launch sets pay two tails - and the pipelined form across sets, hp_stream.hip, is what overlaps the stages now.)
// The object is reusable: init() on a used set keeps every host and device allocation (a stream's slots are solved over and
// over; hipMalloc / hipFree synchronise the device).
struct hp_blockset {
    size_t n_blocks = 0;
    const hp_block_input* in = nullptr;
    hp_block_params prm{};
    int device = 0;
    std::vector<std::vector<hp::RecMeta>> meta;  // per block, per record (job = index into jobs)
    std::vector<uint32_t> job_first;             // per block: its first job (a block's jobs are contiguous, in record order)
    std::vector<hp::W2JobIn> jobs;               // records with overlaps, all blocks
    std::vector<uint64_t> job_alloff;            // per job: offset of its allele row in `alleles`
    std::vector<uint8_t> alleles;                // per-het AlleleTypes of every job, back to back
    std::vector<uint8_t*> allele_ptrs;
    std::vector<hp_wfa_result> wfa_out;
    hp::W2Session* wfa = nullptr;                // graph-WFA inputs resident on the device (sets of >= HP_WFA2_MIN_JOBS records)
    bool wfa_ready = false;                      // ... laid out and uploaded for the current blocks
    bool wfa_laid = false;                       // ... its host-only layout half done (blockset_layout), the upload still to come
    double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // stage times of the last solve
    double rows_ms[5] = {0, 0, 0, 0, 0};          // of the last blockset_rows: blocks that wait for nothing, blocks that held a late result, of which: their local re-alignment launch; the free blocks' launch; the wait for the first collection + its scatter
    double late_wait_ms = 0.0;                   // of the last blockset_rows: time spent waiting for the graph-WFA stage's late results
    double prep[4] = {0, 0, 0, 0};               // of the last init: layout ms, fill + upload ms, total ms, bytes host -> device
    size_t upload_min_jobs = 0;                  // (blockset_layout -> blockset_upload: sets below this take the latency path at solve time)
    uint64_t work[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // hp_blockset_work of the last solve
    std::vector<hp::BlockState> st;
    // between blockset_rows and blockset_solve: the A* batch of the set (packed, uploaded), the blocks in it
    hp_batch* batch = nullptr;
    std::vector<hp_block_view> views;
    std::vector<size_t> kept;
    std::vector<char> unsupported;
    // A SMALL set (the latency path: one wavefront per read, a pass or two of the dense-band kernel, 10-25 ms of waiting) inside a
    // pipeline: its alignment runs on the slot's helper thread and is joined by the rows stage (small_join), so that the alignment
    // stage - one thread - goes on with the next set: the passes of the sets in flight run side by side. Round 5: the blocking
    // per-block callers' merged sets are all of this kind, and their stage 2 was busy all of the time.
    bool small_async = false;                    // set by the pipeline that owns the slot
    std::unique_ptr<hp::HelperThread> small_helper;
    bool small_inflight = false;
    int small_rc = 0;
    std::string small_err;
    double small_kernel_ms = 0.0;
    std::vector<hp_wfa_job> small_jobs;
    std::vector<std::vector<uint8_t>> small_ascii;
    int small_join() {                           // waits for the helper's pass; its status
        if (!small_inflight) return 0;
        small_helper->wait();
        small_inflight = false;
        ms[6] = small_kernel_ms;
        if (small_rc != 0) hp::set_error("%s", small_err.c_str());
        return small_rc;
    }
    ~hp_blockset() { if (small_inflight && small_helper) small_helper->wait(); if (batch) hp_batch_destroy(batch); if (wfa) hp::w2_session_destroy(wfa); }
};

namespace hp {
// lays the set out and uploads its sequences (host threads + PCIe; no kernel but the expansion of the read bases) = the two below
int blockset_init(hp_blockset* bs, size_t n_blocks, const hp_block_input* in, const hp_block_params* p, int device_id);
// ... in two steps (a pipeline runs them as stages of their own: the next set's overlaps while this one's reads cross PCIe):
// validation, every record's overlaps, the job list (host threads; nothing touches the device)
int blockset_layout(hp_blockset* bs, size_t n_blocks, const hp_block_input* in, const hp_block_params* p, int device_id);
// staging copy + PCIe of the set's sequences
int blockset_upload(hp_blockset* bs);
// graph-WFA over every record with overlaps: device graph build, alignment, allele rows (returns after the first collection)
int blockset_wfa(hp_blockset* bs);
// fallback / replay / rows / collapse on host threads (waits for the alignment stage's late results)
int blockset_rows(hp_blockset* bs);
// the A* batch of the set packed (host threads) and uploaded
int blockset_pack(hp_blockset* bs);
// A*, span counts and haplotags, outputs
int blockset_solve(hp_blockset* bs, hp_block_output* out);

// One device's six-stage pipeline (hp_stream.hip): what hp_blockstream_* and the per-block dispatcher (hp_block.hip) both drive.
// submit blocks while `depth` sets are in flight; p == nullptr: the parameters the pipeline was created with; sets complete in
// submission order; wait hands the slot back.
struct Pipeline;
Pipeline* pipeline_create(const hp_block_params* p, int device_id, uint32_t depth, int* status);
int pipeline_submit(Pipeline* s, size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out, uint64_t* ticket);
int pipeline_wait(Pipeline* s, uint64_t ticket, double* stage_ms, uint64_t* work);
uint64_t pipeline_load(Pipeline* s, bool* has_free_slot);   // records in flight
void pipeline_wait_free(Pipeline* s);                       // returns when a slot is free
uint32_t pipeline_free_slots(Pipeline* s);
void pipeline_destroy(Pipeline* s);
}  // namespace hp
