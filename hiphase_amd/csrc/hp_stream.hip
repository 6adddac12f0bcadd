// hp_stream.hip — block sets through the path as a PIPELINE (hp_blockstream_*).
//
// HiPhase sees every phase block once (reference src/main.rs:337-408: blocks are generated, queued to the worker pool and
// written in order; src/phaser.rs:513-543: a block's reads are loaded, then it is solved). A caller that hands over one block
// set after the other therefore wants set k + 2 to be laid out and cross PCIe while set k + 1 is being aligned and set k is
// being solved - not one after the other. Six stages, one thread each, every stage with its own HIP streams, device-buffer
// cache, pinned staging and host worker pool (all of them per-thread state of the library), so they overlap on the host and
// on the device:
//
//   stage 0 (hp::blockset_layout) validation, overlaps of every record, the job list, and the host-only half of the sequence layout
//                                 (offsets, the runs the copy engines read in place, the length order): host threads only, 10 ms a set,
//                                 nothing of it in front of the set's turn on the PCIe link
//   stage 1 (hp::blockset_upload) the tables filled and everything sent: reads in place from hp_host_alloc memory, or staged piece by
//                                 piece as the caller holds them (ASCII or the BAM's own 4-bit codes) while the previous piece crosses
//   stage 2 (hp::blockset_wfa)    base expansion + device graph build + the graph-WFA launch set + allele rows, all QUEUED: the stage
//                                 does not wait for its own results (the session's helper thread collects them and goes on with the
//                                 late pass), so the next set's launch set follows this one's on the device
//   stage 3 (hp::blockset_rows)   the first collection handed over (w2_session_collected), fallback replay / qualities / collapse on host
//                                 threads (mostly a WAIT: for the class kernels, then for the late results of the set's alignment
//                                 stage - the largest class's tail, the leftovers)
//   stage 4 (hp::blockset_pack)   the A* batch packed (host threads) and uploaded
//   stage 5 (hp::blockset_solve)  A* (a latency-bound kernel: the host thread mostly waits), span counts and haplotags, outputs
//                                 into the caller's buffers
//
// hp::Pipeline is one device's stages. The public hp_blockstream is one Pipeline per device it was created for
// (device_id >= 0: that device; -1: every visible device, sets dealt to the least-loaded pipeline - blocks are independent,
// phaser.rs:406-411, so the node's GPUs need no exchange step; reference fan-out: main.rs:332-408, results re-ordered by
// block index, writers/ordered_vcf_writer.rs:158-170); the per-block entries (hp_solve_blocks(1, ...), hp_block_submit) feed
// the same pipelines through the dispatcher of hp_block.hip.
//
// Sets of one pipeline complete in submission order. `depth` slots hold the sets in flight; a slot keeps its host vectors and device buffers
// from one set to the next (hipMalloc / hipFree synchronise the device). submit() blocks while every slot is taken - the
// back-pressure the reference's bounded job queue applies (main.rs:362-383).
#include "hp_block.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace hp;

namespace {

double st_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

struct Slot {
    hp_blockset bs;
    enum State { FREE, QUEUED, DONE } state = FREE;
    uint64_t ticket = 0;
    hp_block_params prm{};   // the set's own parameters (the dispatcher merges callers with equal ones)
    uint64_t load = 0;       // records of the set (what "least loaded" counts)
    size_t n_blocks = 0;
    const hp_block_input* in = nullptr;
    hp_block_output* out = nullptr;
    int rc = HP_OK;
    std::string err;
    double t_submit = 0, t_begin[6] = {0, 0, 0, 0, 0, 0}, t_end[6] = {0, 0, 0, 0, 0, 0};
};

}  // namespace

struct hp::Pipeline {
    hp_block_params prm{};
    int device = 0;
    uint64_t load = 0;   // records in flight (guarded by m)
    std::vector<std::unique_ptr<Slot>> slots;
    std::mutex m;
    std::condition_variable cv;
    static constexpr int N_STAGES = 6;
    std::deque<Slot*> q[N_STAGES];     // waiting for stage 1 .. 5, in ticket order
    uint64_t next_ticket = 1;
    uint64_t next_in[N_STAGES] = {1, 1, 1, 1, 1, 1};   // the ticket each stage takes next
    double t_zero = 0.0;
    bool quit = false;
    // one thread per stage; a seventh only behind the experiment switches HP_STREAM_ROWS_THREADS=2 / HP_STREAM_WFA_THREADS=2 (a second
    // rows or alignment thread: two sets share the stage and still reach the next one in ticket order - measured equal / slower)
    static constexpr int N_THREADS = N_STAGES + 7;   // (up to eight threads for the last stage: HP_STREAM_SOLVE_THREADS)
    std::thread th[N_THREADS];
    std::unique_ptr<WorkerPool> pool[N_THREADS];
    void stage_thread(int t);
    int extra_stage = 2;
    int index = 0;   // among the pipelines alive when it was created (thread names)
    void stage_loop(int k);
};

void hp::Pipeline::stage_thread(int t) {
    // (the first pipeline of a process keeps the plain names bench.py's host_cpu table has always shown; further ones - several devices
    // behind one stream, the dispatcher's per-device pipelines - carry their index: hp-p3s5 = pipeline 3, stage 5)
    { char b[16]; if (index == 0) std::snprintf(b, sizeof b, "hp-s%d", t < N_STAGES ? t : extra_stage); else std::snprintf(b, sizeof b, "hp-p%ds%d", index % 100, t < N_STAGES ? t : extra_stage); name_thread(b); }
    WorkerPool::set_thread_pool(pool[t].get());
    stage_loop(t < N_STAGES ? t : extra_stage);   // (the last thread: a second one for the alignment stage - or, as an experiment, the rows stage)
}

void hp::Pipeline::stage_loop(int k) {
    (void)hp_set_device(device);
    // CU partitions for the stages (hp_common.h) are an experiment switch, off by default. Measured on the bench workload: the
    // persistent graph-WFA kernels fill every compute unit (three wavefronts per SIMD is all their registers allow), so another
    // stage's kernels wait for their workgroups to leave - but binding graph-WFA to 7/8 of the CUs (with the other stages confined
    // to the eighth, HP_STREAM_PARTITION=1, or on the whole device, =2) made the A* kernels 6 x slower (113 vs 18 ms: they start
    // behind the CU-masked queue's kernels) and the step 165 instead of 73 ms.
    static const int part = [] { const char* e = std::getenv("HP_STREAM_PARTITION"); return e ? std::atoi(e) : 0; }();
    if (part == 1) g_cu_partition = k == 2 ? 2 : 1;
    else if (part == 2) g_cu_partition = k == 2 ? 2 : 0;
    // ... what works instead: the alignment stage leaves a share of the wavefront slots empty (hp_wfa2.hip)
    static const int reserve = [] { const char* e = std::getenv("HP_STREAM_RESERVE_PCT"); return e ? std::max(0, std::atoi(e)) : 8; }();
    if (k == 2) g_wfa2_reserve_pct = reserve;
    g_host_share_div = (k == 1 || k == 3) ? 2 : 4;   // (the upload stage fills the tables of a gigabyte of reads, the row stage is the one whose host work sets the period; the others' parallel regions are short)
    for (;;) {
        Slot* s = nullptr;
        {
            std::unique_lock<std::mutex> lk(m);
            // a stage takes the sets strictly in ticket order (next_in[k]); with two threads in a stage each takes the next one
            cv.wait(lk, [&]() { return quit || (!q[k].empty() && q[k].front()->ticket == next_in[k]); });
            if (q[k].empty() || q[k].front()->ticket != next_in[k]) return;   // quit, nothing left to do
            s = q[k].front();
            q[k].pop_front();
            ++next_in[k];
        }
        s->t_begin[k] = st_now_ms();
        if (s->rc == HP_OK && s->n_blocks) {   // (an empty set, or one that failed an earlier stage, just travels on: tickets complete in order)
            int rc = HP_OK;
            try {   // (a stage thread must never take the host process down: a failed host allocation is the set's status, and the set travels on)
                if (k == 0) rc = blockset_layout(&s->bs, s->n_blocks, s->in, &s->prm, device);
                else if (k == 1) rc = blockset_upload(&s->bs);
                else if (k == 2) rc = blockset_wfa(&s->bs);
                else if (k == 3) rc = blockset_rows(&s->bs);
                else if (k == 4) rc = blockset_pack(&s->bs);
                else rc = blockset_solve(&s->bs, s->out);
                if (rc != HP_OK) { s->rc = rc; s->err = hp_last_error(); }
            } catch (const std::exception& e) { s->rc = HP_ERR_OOM; s->err = std::string("host allocation failed in a pipeline stage: ") + e.what(); }
            // (a set that fails travels on without its remaining stages - the rows stage, which joins a small set's alignment pass on the
            // slot's helper thread, among them: joined here, before the caller is told and frees the inputs that pass reads)
            if (s->rc != HP_OK) { const std::string keep = s->err; (void)s->bs.small_join(); s->err = keep; }
        }
        s->t_end[k] = st_now_ms();
        {
            std::unique_lock<std::mutex> lk(m);
            if (k + 1 < N_STAGES) {   // (ticket order: two threads of one stage may finish out of turn)
                auto it = q[k + 1].begin();
                while (it != q[k + 1].end() && (*it)->ticket < s->ticket) ++it;
                q[k + 1].insert(it, s);
            } else s->state = Slot::DONE;
            cv.notify_all();
        }
    }
}

hp::Pipeline* hp::pipeline_create(const hp_block_params* p, int device_id, uint32_t depth, int* status) {
    auto fail = [&](int rc) -> Pipeline* { if (status) *status = rc; return nullptr; };
    if (!p) { set_error("null argument"); return fail(HP_ERR_ARG); }
    if (hp_device_count() <= 0) { set_error("no HIP device visible; there is no CPU fallback"); return fail(HP_ERR_HIP); }
    if (depth == 0) depth = 5;
    if (depth > 16) { set_error("depth %u: at most 16 block sets in flight", depth); return fail(HP_ERR_ARG); }
    auto s = std::unique_ptr<Pipeline>(new Pipeline());
    s->prm = *p;
    s->device = device_id < 0 ? hp_default_device() : device_id;
    if (s->device >= hp_device_count()) { set_error("device %d: %d visible", s->device, hp_device_count()); return fail(HP_ERR_ARG); }
    if (hp_set_device(s->device) != hipSuccess) { set_error("hipSetDevice(%d) failed - no usable GPU; there is no CPU fallback", s->device); return fail(HP_ERR_HIP); }
    for (uint32_t i = 0; i < depth; ++i) { s->slots.emplace_back(new Slot()); s->slots.back()->bs.small_async = !(std::getenv("HP_STREAM_SMALL_ASYNC") && std::getenv("HP_STREAM_SMALL_ASYNC")[0] == '0'); }   // (hp_block.h: small sets' alignment off the stage thread)
    for (int k = 0; k < Pipeline::N_THREADS; ++k) s->pool[k].reset(new WorkerPool());
    Pipeline* raw = s.get();
    // One thread per stage. Two experiment switches add a fifth thread: HP_STREAM_WFA_THREADS=2 (a second alignment thread with its
    // own streams and scratch, so that the next set's kernels are queued while this one's drain: measured 53 vs 48 ms per step,
    // slower) and HP_STREAM_ROWS_THREADS=2 (a second rows thread: 65 vs 50 ms, slower) - the device is the bottleneck, one more set
    // in flight only adds contention.
    const char* wt = std::getenv("HP_STREAM_WFA_THREADS");
    const char* rt = std::getenv("HP_STREAM_ROWS_THREADS");
    // HP_STREAM_SOLVE_THREADS=n: n threads for the last stage - A* is a latency chain, not issue (3e8 busy cycles per set against the
    // alignment kernels' 2.3e9), so set k + 1's chain can run beside set k's.
    // Measured (round 5, default bench, three runs a side): a second thread 2.15 -> 2.19 M hets/s at depth 6, 2.29 M at depth 7 (one more set
    // in flight feeds the second chain), 2.34 M with the 48-variant warm-up of hp_astar.hip on top. Round 6: the deep-coverage workload
    // (hp_synth_reads_deep60: 60x, conflicting rows - the search of its largest block is 0.85 s of ONE wavefront) is bound by exactly this
    // chain: a set's A* takes 850 ms, the device idles, and the stream's period is 850 ms / threads - as many threads as the stream holds
    // sets (depth) by default, at most eight; a thread without a set sleeps on the pipeline's condition variable. =1: one thread.
    // Measured (round 6, three sizes, `gpurun_out/r6_solve`): the default and the HiFi-shaped bench 2.45-2.50 M hets/s at 2, 4 and 7 threads
    // alike (the third thread never finds a set waiting); deep60 41 k -> 65 k -> 82 k hets/s at 2 / 4 / 7 (14 steps, 0.86 s of them the
    // first set's way through).
    const char* sv = std::getenv("HP_STREAM_SOLVE_THREADS");
    int n_threads = Pipeline::N_STAGES + 1;
    if (rt && std::atoi(rt) >= 2) s->extra_stage = 3;
    else if (wt && std::atoi(wt) >= 2) s->extra_stage = 2;
    else {
        const int want = sv ? std::atoi(sv) : (int)depth;
        s->extra_stage = 5;
        n_threads = Pipeline::N_STAGES + std::max(0, std::min(want, 8) - 1);
    }
    s->index = g_pipelines.fetch_add(1);
    for (int k = 0; k < n_threads; ++k) s->th[k] = std::thread([raw, k]() { raw->stage_thread(k); });
    if (status) *status = HP_OK;
    return s.release();
}

int hp::pipeline_submit(Pipeline* s, size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out, uint64_t* ticket) {
    if (!s || !ticket || (n_blocks && (!in || !out))) { set_error("null argument"); return HP_ERR_ARG; }
    uint64_t load = 1;
    for (size_t b = 0; b < n_blocks; ++b) load += in[b].n_records;
    std::unique_lock<std::mutex> lk(s->m);
    Slot* slot = nullptr;
    s->cv.wait(lk, [&]() {
        for (auto& x : s->slots) if (x->state == Slot::FREE) { slot = x.get(); return true; }
        return false;
    });
    slot->state = Slot::QUEUED;
    slot->ticket = s->next_ticket++;
    slot->prm = p ? *p : s->prm;
    slot->load = load;
    s->load += load;
    slot->n_blocks = n_blocks; slot->in = in; slot->out = out;
    slot->rc = HP_OK; slot->err.clear();
    slot->t_submit = st_now_ms();
    *ticket = slot->ticket;
    s->q[0].push_back(slot);
    s->cv.notify_all();
    return HP_OK;
}

uint64_t hp::pipeline_load(Pipeline* s, bool* has_free_slot) {
    std::unique_lock<std::mutex> lk(s->m);
    if (has_free_slot) { *has_free_slot = false; for (auto& x : s->slots) if (x->state == Slot::FREE) *has_free_slot = true; }
    return s->load;
}

uint32_t hp::pipeline_free_slots(Pipeline* s) {
    std::unique_lock<std::mutex> lk(s->m);
    uint32_t n = 0;
    for (auto& x : s->slots) n += x->state == Slot::FREE ? 1u : 0u;
    return n;
}

void hp::pipeline_wait_free(Pipeline* s) {
    std::unique_lock<std::mutex> lk(s->m);
    s->cv.wait(lk, [&]() { for (auto& x : s->slots) if (x->state == Slot::FREE) return true; return false; });
}

int hp::pipeline_wait(Pipeline* s, uint64_t ticket, double* stage_ms, uint64_t* work) {
    if (!s) { set_error("null argument"); return HP_ERR_ARG; }
    std::unique_lock<std::mutex> lk(s->m);
    Slot* slot = nullptr;
    for (auto& x : s->slots) if (x->state != Slot::FREE && x->ticket == ticket) slot = x.get();
    if (!slot) { set_error("hp_blockstream_wait: ticket %llu is not in flight", (unsigned long long)ticket); return HP_ERR_ARG; }
    s->cv.wait(lk, [&]() { return slot->state == Slot::DONE; });
    const int rc = slot->rc;
    if (rc != HP_OK) set_error("%s", slot->err.c_str());
    if (std::getenv("HP_STREAM_TRACE")) {   // the set's way through the stages, ms since the stream's first submit
        if (s->t_zero == 0.0) s->t_zero = slot->t_submit;
        fprintf(stderr, "[hp] set %llu: submit %.1f | layout %.1f-%.1f | s1 %.1f-%.1f | s2 %.1f-%.1f | s3 %.1f-%.1f (first collection in hand after %.1f, free blocks %.1f [local re-alignment %.1f], waited %.1f for the late results, their blocks %.1f [%.1f]) | s4 %.1f-%.1f | s5 %.1f-%.1f (A* %.1f of which kernels %.1f, post %.1f)\n", (unsigned long long)ticket,
                slot->t_submit - s->t_zero, slot->t_begin[0] - s->t_zero, slot->t_end[0] - s->t_zero, slot->t_begin[1] - s->t_zero, slot->t_end[1] - s->t_zero, slot->t_begin[2] - s->t_zero, slot->t_end[2] - s->t_zero,
                slot->t_begin[3] - s->t_zero, slot->t_end[3] - s->t_zero, slot->bs.rows_ms[4], slot->bs.rows_ms[0] - slot->bs.rows_ms[4], slot->bs.rows_ms[3], slot->bs.late_wait_ms, slot->bs.rows_ms[1], slot->bs.rows_ms[2], slot->t_begin[4] - s->t_zero, slot->t_end[4] - s->t_zero, slot->t_begin[5] - s->t_zero, slot->t_end[5] - s->t_zero, slot->bs.ms[3], slot->bs.ms[7], slot->bs.ms[4]);
    }
    if (stage_ms) {
        const hp_blockset& B = slot->bs;
        stage_ms[0] = B.prep[0]; stage_ms[1] = B.prep[1];
        stage_ms[2] = B.ms[0]; stage_ms[3] = B.ms[1]; stage_ms[4] = B.ms[2]; stage_ms[5] = B.ms[3]; stage_ms[6] = B.ms[4];
        stage_ms[7] = slot->t_end[5] - slot->t_submit;
        stage_ms[8] = B.ms[6]; stage_ms[9] = B.ms[7];
        stage_ms[10] = B.prep[3];
        stage_ms[11] = (slot->t_begin[0] - slot->t_submit) + (slot->t_begin[1] - slot->t_end[0]) + (slot->t_begin[2] - slot->t_end[1]) + (slot->t_begin[3] - slot->t_end[2]) +
                       (slot->t_begin[4] - slot->t_end[3]) + (slot->t_begin[5] - slot->t_end[4]);
        // (stage 1 = the upload stage; the layout stage before it - host only, since round 4 a stage of its own - is stage_ms[0])
        stage_ms[12] = slot->t_end[1] - slot->t_begin[1]; stage_ms[13] = slot->t_end[2] - slot->t_begin[2]; stage_ms[14] = slot->t_end[3] - slot->t_begin[3];
        stage_ms[15] = (slot->t_end[4] - slot->t_begin[4]) + (slot->t_end[5] - slot->t_begin[5]);   // (pack + solve)
    }
    if (work) for (int i = 0; i < 8; ++i) work[i] = slot->bs.work[i];
    slot->bs.in = nullptr;   // (nothing of the caller's is kept)
    s->load -= slot->load;
    slot->state = Slot::FREE;
    s->cv.notify_all();
    return rc;
}

void hp::pipeline_destroy(Pipeline* s) {
    if (!s) return;
    {
        std::unique_lock<std::mutex> lk(s->m);
        // sets still in flight are finished first (their buffers belong to the caller)
        s->cv.wait(lk, [&]() {
            for (auto& x : s->slots) if (x->state == Slot::QUEUED) return false;
            return true;
        });
        s->quit = true;
        s->cv.notify_all();
    }
    if (std::getenv("HP_STREAM_TRACE")) fprintf(stderr, "[hp] device-wide waits for memory so far (hipMalloc / hipHostMalloc calls): %d\n", hp::g_device_syncing_allocs.load()), fprintf(stderr, "[hp] streams created by the library so far: %d (GPU_MAX_HW_QUEUES=%s)\n", hp::g_streams_created.load(), std::getenv("GPU_MAX_HW_QUEUES") ? std::getenv("GPU_MAX_HW_QUEUES") : "unset");
    for (auto& t : s->th) if (t.joinable()) t.join();
    g_pipelines.fetch_sub(1);
    delete s;
}

// ---- the public stream: one pipeline per device -------------------------------------------------------------------------------
// device_id >= 0: that device. device_id == -1: every visible device (HP_STREAM_DEVICES=n: n pipelines, pipeline v on device
// v % visible - the test hook for a box with fewer GPUs, as HP_QUEUE_WORKERS is for the dispatcher): a set goes to the pipeline
// with the fewest records in flight among those with a free slot (the first one on ties, so one device behaves as before), and
// submit blocks only while EVERY pipeline is full. Ticket = the pipeline's own ticket << 8 | pipeline: wait(ticket) finds its
// set without a table; the sets of one device complete in order, a caller that waits in submission order (bench.py, the
// INTEGRATION.md patch) gets its results in submission order.
struct hp_blockstream {
    std::vector<hp::Pipeline*> pipes;
    std::mutex m;
    std::condition_variable cv;   // a wait() freed a slot somewhere
};

extern "C" hp_blockstream* hp_blockstream_create(const hp_block_params* p, int device_id, uint32_t depth, int* status) {
    auto fail = [&](int rc) -> hp_blockstream* { if (status) *status = rc; return nullptr; };
    if (!p) { set_error("null argument"); return fail(HP_ERR_ARG); }
    const int ndev = hp_device_count();
    if (ndev <= 0) { set_error("no HIP device visible; there is no CPU fallback"); return fail(HP_ERR_HIP); }
    if (device_id >= ndev) { set_error("device %d: %d visible", device_id, ndev); return fail(HP_ERR_ARG); }
    std::vector<int> devs;
    if (device_id >= 0) devs.push_back(device_id);
    else {
        const char* e = std::getenv("HP_STREAM_DEVICES");
        const int n = e ? std::max(1, std::min(64, std::atoi(e))) : ndev;
        for (int v = 0; v < n; ++v) devs.push_back(v % ndev);
    }
    auto s = std::unique_ptr<hp_blockstream>(new hp_blockstream());
    for (int d : devs) {
        int rc = HP_OK;
        hp::Pipeline* pl = hp::pipeline_create(p, d, depth, &rc);
        if (!pl) { for (auto* x : s->pipes) hp::pipeline_destroy(x); return fail(rc); }
        s->pipes.push_back(pl);
    }
    if (status) *status = HP_OK;
    return s.release();
}

extern "C" int hp_blockstream_devices(const hp_blockstream* s) { return s ? (int)s->pipes.size() : 0; }

extern "C" int hp_blockstream_submit(hp_blockstream* s, size_t n_blocks, const hp_block_input* in, hp_block_output* out, uint64_t* ticket) {
    if (!s || !ticket || (n_blocks && (!in || !out))) { set_error("null argument"); return HP_ERR_ARG; }
    size_t pick = 0;
    if (s->pipes.size() > 1) {
        std::unique_lock<std::mutex> lk(s->m);   // (one submitter at a time chooses: two must not both take the last free slot's pipeline)
        for (;;) {
            uint64_t best = UINT64_MAX;
            bool any = false;
            for (size_t k = 0; k < s->pipes.size(); ++k) {
                bool free_slot = false;
                const uint64_t l = hp::pipeline_load(s->pipes[k], &free_slot);
                if (free_slot && l < best) { best = l; pick = k; any = true; }
            }
            if (any) break;
            s->cv.wait_for(lk, std::chrono::milliseconds(2));   // every pipeline is full: the back-pressure of main.rs:362-383
        }
        uint64_t t = 0;
        const int rc = hp::pipeline_submit(s->pipes[pick], n_blocks, in, nullptr, out, &t);   // (has a free slot: does not block)
        if (rc == HP_OK) *ticket = (t << 8) | (uint64_t)pick;
        return rc;
    }
    uint64_t t = 0;
    const int rc = hp::pipeline_submit(s->pipes[0], n_blocks, in, nullptr, out, &t);
    if (rc == HP_OK) *ticket = t << 8;
    return rc;
}

extern "C" int hp_blockstream_wait(hp_blockstream* s, uint64_t ticket, double* stage_ms, uint64_t* work) {
    if (!s) { set_error("null argument"); return HP_ERR_ARG; }
    const size_t k = (size_t)(ticket & 0xFFu);
    if (k >= s->pipes.size()) { set_error("hp_blockstream_wait: ticket %llu is not in flight", (unsigned long long)ticket); return HP_ERR_ARG; }
    const int rc = hp::pipeline_wait(s->pipes[k], ticket >> 8, stage_ms, work);
    if (s->pipes.size() > 1) { std::lock_guard<std::mutex> lk(s->m); s->cv.notify_all(); }
    return rc;
}

extern "C" void hp_blockstream_destroy(hp_blockstream* s) {   // finishes the sets still in flight first
    if (!s) return;
    for (auto* p : s->pipes) hp::pipeline_destroy(p);
    delete s;
}
