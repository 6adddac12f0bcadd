// hp_wfa2.hip — host side of the second-generation graph-WFA stage (hp_wfa2_kernel.hip).
//
// hp_wfa_assign_batch (the seam at reference src/read_parsing.rs:769-800) now does nothing per read on the host:
//   1. the batch is laid out for the device - the union of the jobs' reference windows once, every distinct
//      hp_wfa_variant once (jobs pass slices of their block's variant vectors: the address ranges are merged exactly
//      like the reference windows), the read bases - and uploaded from pinned staging;
//   2. hp_wfa2_build_kernel builds every read's graph on the device (wfa_graph.rs:119-284);
//   3. hp_wfa2_kernel<G, W> aligns the reads, several per wavefront, one launch per graph-size class;
//   4. hp_wfa2_map_kernel turns traversed nodes into the per-het allele row (read_parsing.rs:790-800);
//   5. reads that outgrow the compact state or the device builder's fixed queues (W2_ST_NEED_BIG / W2B_NEED_HOST)
//      are re-run by the dense-band path of hp_wfa.hip (host graph build + hp_wfa_kernel). Same results either way.
#include "hp_wfa2_kernel.hip"
#include "hp_wfa3_kernel.hip"
#include "hp_wfa2_host.h"

#include <algorithm>
#include <array>
#include <memory>
#include <atomic>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <pthread.h>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace hp {

namespace {

double w2_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}


inline uint32_t w2_hcap_log2() {   // keys per group's overflow hash set: 2^n x 8 B (HP_WFA2_HCAP_LOG2 for experiments; 2 048 before the
                                     // per-node records took over - 128 serve the bench and 2-4 % noise just as well)
    static const uint32_t v = [] { const char* e = std::getenv("HP_WFA2_HCAP_LOG2"); const int x = e ? std::atoi(e) : 8; return (uint32_t)std::min(14, std::max(6, x)); }();
    return v;
}
#define W2_HCAP_LOG2 (w2_hcap_log2())
constexpr uint32_t W2_REGIONS = 5;   // scratch regions of a context: one per smaller class, three taking turns for the largest

// per-(thread, device) state that survives across calls
struct W2Context {
    int device = -1;
    DevBuf htab;             // capped-diagonal hash sets, [groups][1 << W2_HCAP_LOG2]; zeroed once, then tagged
    DevBuf gsets;            // [groups][W2_SET_STRIDE_MAX] the arena slots' traversed-node sets
    uint32_t htab_groups = 0;
    int last_gen = 0;        // kernel generation of the last run on this scratch (the generations lay a group's region out differently)
    // scratch regions: [0], [1] the two smaller classes (htab_groups groups each, their own set strides), [2..4] the largest class
    // (large_groups groups each; three, taken in turns by consecutive runs)
    uint32_t large_groups = 0;
    size_t region_set_off[5] = {0, 0, 0, 0, 0};   // dwords into gsets
    size_t region_hash_off[5] = {0, 0, 0, 0, 0};  // groups into htab
    uint32_t tag_next = 0;   // tags handed out so far (tag 0 = empty)
    // streams per CU partition (hp_common.h): the main stream, and one per graph-size class (the three launches overlap)
    // (the largest class's kernel outlives run() - it is the tail of the launch set, collected by late() - so consecutive runs of a
    // block stream take turns on three streams, and three scratch regions, for it: the next set's kernel neither queues behind this
    // one's tail nor is waited for by this one's late())
    struct Streams { hipStream_t stream = nullptr; hipStream_t cstream[3] = {nullptr, nullptr, nullptr}; hipStream_t c2x[2] = {nullptr, nullptr}; } ps[3];
    uint32_t run_no = 0;
    hipEvent_t cfork = nullptr, cjoin[3] = {nullptr, nullptr, nullptr};
    PinBuf stage;            // upload staging: seq bytes, then the tables
    PinBuf down;             // download staging
    std::unique_ptr<HelperThread> helper;   // runs the leftovers' dense-band pass beside the scatter of the other results
    void drop_streams() {
        for (auto& s : ps) {
            if (s.stream) { (void)hipStreamDestroy(s.stream); s.stream = nullptr; }
            for (int k = 0; k < 3; ++k) if (s.cstream[k]) { (void)hipStreamDestroy(s.cstream[k]); s.cstream[k] = nullptr; }
            for (int k = 0; k < 2; ++k) if (s.c2x[k]) { (void)hipStreamDestroy(s.c2x[k]); s.c2x[k] = nullptr; }
        }
    }
    // created on first use in that partition. classes: the five streams of the class launches too - only a thread that runs launch
    // sets needs them (the layout stage of a block stream does not), and every stream a process creates beyond the runtime's
    // hardware queues (GPU_MAX_HW_QUEUES) shares a queue with another one: a side path's kernel that lands in a persistent class
    // kernel's queue waits for that kernel, a class kernel behind a dense-band pass starts late - which streams share depends on
    // the order the threads got going, and a process where it went wrong ran its launch sets in 42-47 instead of 28 ms
    int streams(int part, Streams** out, bool classes) {
        Streams& s = ps[part];
        if (!s.stream) HP_HIP_CHECK(hp_stream_create(&s.stream, device));
        // (measured: low stream priority for these persistent kernels makes THEM 40 % slower - 48 vs 35 ms - and nothing else faster)
        if (classes) {
            for (int k = 0; k < 3; ++k) if (!s.cstream[k]) HP_HIP_CHECK(hp_stream_create(&s.cstream[k], device));
            for (int k = 0; k < 2; ++k) if (!s.c2x[k]) HP_HIP_CHECK(hp_stream_create(&s.c2x[k], device));
        }
        *out = &s;
        return HP_OK;
    }
    ~W2Context() {
        drop_streams();
        for (int k = 0; k < 3; ++k) if (cjoin[k]) (void)hipEventDestroy(cjoin[k]);
        if (cfork) (void)hipEventDestroy(cfork);
    }
};
thread_local W2Context g_w2;
// the calling thread's context, (re)bound to `device`: a session may be prepared on one thread and run on another (a block
// stream's stages, a caller's thread pool) - every thread brings its own streams, scratch sets and staging
W2Context& w2_context(int device) {
    W2Context& cx = g_w2;
    if (cx.device != device) {
        cx.htab.release(); cx.gsets.release(); cx.htab_groups = 0; cx.tag_next = 0;
        cx.drop_streams();
        cx.device = device;
    }
    return cx;
}

unsigned w2_host_threads(size_t n, size_t per_thread) {
    const char* tenv = std::getenv("HP_WFA_HOST_THREADS");
    unsigned nt = tenv ? (unsigned)std::atoi(tenv) : host_threads(8u);
    return (unsigned)std::min<size_t>(std::max(1u, nt), std::max<size_t>(1, n / per_thread));
}
template <class F> void w2_parallel(unsigned nt, F&& f) {
    WorkerPool::get().run(std::max(1u, nt), [&f, nt](unsigned t) { f(t, std::max(1u, nt)); });
}

// A few threads per device that run the block sets' EARLY passes, each one after the other (W2Session::early_pass: the records a set's layout
// routed past the compact kernels take their way out - reference-window test, dense band - while the set's launch set runs). One
// thread, hence one stream and one dense-band scratch per device however many sets are in flight: every stream a process creates
// beyond the runtime's hardware queues shares a queue with another one (W2Context::streams). Never destroyed: a set may still be
// in its hands when the static destructors run.
struct EarlyWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool started = false;   // (not th.joinable(): the thread is detached, and a detached thread is never joinable - every post() started another one)
    void post(std::function<void()> f) {
        std::unique_lock<std::mutex> lk(m);
        if (!started) {
            started = true;
            th = std::thread([this]() {
                (void)pthread_setname_np(pthread_self(), "hp-early");
                g_thread_stream_high = true;   // (a hardware queue that no persistent class kernel sits in: hp_api.hip, thread_stream)
                std::unique_lock<std::mutex> l2(m);
                for (;;) {
                    cv.wait(l2, [this]() { return !q.empty(); });
                    std::function<void()> t = std::move(q.front());
                    q.pop_front();
                    l2.unlock();
                    t();
                    l2.lock();
                }
            });
            th.detach();
        }
        q.push_back(std::move(f));
        cv.notify_all();
    }
};
// launch sets of a device between their launch and their first collection (W2Session::run, HP_WFA2_ROUTE)
std::atomic<int>& sets_aligning(int device) {
    static std::atomic<int> c[64];
    return c[(unsigned)device % 64u];
}
// (two per device, taken in turns: a pass is 20-25 ms of latency chain - the test, then up to max_edit_distance rounds of the dense
// band - and a stream's sets arrive every 18-20 ms; HP_EARLY_WORKERS)
EarlyWorker& early_worker(int device) {
    static std::mutex gm;
    static std::map<int, std::pair<std::vector<EarlyWorker*>, size_t>>* ws = new std::map<int, std::pair<std::vector<EarlyWorker*>, size_t>>();
    static const size_t per_device = [] { const char* e = std::getenv("HP_EARLY_WORKERS"); return (size_t)(e ? std::max(1, std::min(8, std::atoi(e))) : 2); }();
    std::lock_guard<std::mutex> lk(gm);
    auto& d = (*ws)[device];
    if (d.first.empty()) for (size_t k = 0; k < per_device; ++k) d.first.push_back(new EarlyWorker());
    return *d.first[d.second++ % d.first.size()];
}

// union of address ranges [p, p + len): sorted, merged where they overlap or touch
struct Range { const uint8_t* lo; const uint8_t* hi; uint64_t dev; };
void merge_ranges(std::vector<std::pair<const uint8_t*, uint64_t>>& iv, std::vector<Range>& out, uint64_t elem) {
    std::sort(iv.begin(), iv.end());
    for (auto& v : iv) {
        // variants: only merge on element boundaries (a slice of the same array), bytes: always
        if (!out.empty() && v.first <= out.back().hi && (uint64_t)(v.first - out.back().lo) % elem == 0)
            out.back().hi = std::max(out.back().hi, v.first + v.second);
        else out.push_back({v.first, v.first + v.second, 0});
    }
}
// one region of `max_groups` groups per class in htab / gsets: the class launches run concurrently
// group_jobs > 0: a group leaves after that many jobs, the grid covers the list (at most max_groups groups: the scratch regions)
// which kernel generation aligns the classes: 3 = flat sorted slot lists (hp_wfa3_kernel.hip, round 4), 2 = per-node hull arenas
// (hp_wfa2_kernel.hip). HP_WFA_GEN=2 keeps the second generation selectable (A/B runs, the parity tests run both).
int w2_gen() {
    const char* e = std::getenv("HP_WFA_GEN");   // (read per call: tests switch it inside one process)
    return (e && e[0] == '2') ? 2 : 3;
}
template <int W, bool WIDE = false> constexpr size_t w2_group_dwords() {   // scratch per resident group: what either generation needs
    return W2Cfg<W, WIDE>::GROUP_DWORDS > W3Cfg<W, WIDE>::GROUP_DWORDS ? (size_t)W2Cfg<W, WIDE>::GROUP_DWORDS : (size_t)W3Cfg<W, WIDE>::GROUP_DWORDS;
}
template <int G, int W, bool WIDE = false> size_t w2_lds_bytes(int gen) {
    return (size_t)(gen == 3 && G <= 16 ? W3Cfg<W, WIDE>::BYTES : W2Cfg<W, WIDE>::BYTES) * (64 / G);
}
template <int G, int W, bool WIDE = false> uint32_t w2_grid(uint32_t n_items, int n_cu, uint32_t max_groups, uint32_t group_jobs = 0) {
    constexpr uint32_t NG = 64 / G;
    if (group_jobs) return std::max<uint32_t>(1u, std::min<uint32_t>((n_items + NG * group_jobs - 1) / (NG * group_jobs), max_groups / NG));
    const size_t lds = w2_lds_bytes<G, W, WIDE>(w2_gen());
    const size_t lds_alloc = (lds + 1279) / 1280 * 1280;   // gfx950 allocates LDS in 1280-byte granules
    uint32_t per_cu = (uint32_t)std::min<size_t>(32, (160 * 1024) / lds_alloc);
    if (const char* e = std::getenv("HP_WFA2_PER_CU")) per_cu = std::max(1, std::min((int)per_cu, std::atoi(e)));
    uint32_t grid = std::min<uint32_t>((n_items + NG - 1) / NG, (uint32_t)n_cu * per_cu);
    grid = std::min<uint32_t>(grid, max_groups / NG);
    return grid == 0 ? 1u : grid;
}
template <int G, int W, bool WIDE = false> int w2_launch(const W2Batch& B, uint32_t n_items, int n_cu, uint32_t max_groups, hipStream_t st, uint32_t* groups_used) {
    constexpr uint32_t NG = 64 / G;
    const int gen = G <= 16 ? w2_gen() : 2;
    const size_t lds = w2_lds_bytes<G, W, WIDE>(gen);
    const uint32_t grid = w2_grid<G, W, WIDE>(n_items, n_cu, max_groups, B.group_jobs);
    static std::atomic<int> attr_set{0};   // (per instantiation; bit g: generation g)
    if (lds > 64 * 1024 || !(attr_set.load() & (1 << gen))) {
        if constexpr (G <= 16) {
            if (gen == 3) HP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_wfa3_kernel<G, W, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (gen == 2) HP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_wfa2_kernel<G, W, WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(1 << gen);
    }
    *groups_used = grid * NG;
    if constexpr (G <= 16) {
        if (gen == 3) {
            hipLaunchKernelGGL((hp_wfa3_kernel<G, W, WIDE>), dim3(grid), dim3(64), lds, st, B);
            HP_HIP_CHECK(hipGetLastError());
            return HP_OK;
        }
    }
    hipLaunchKernelGGL((hp_wfa2_kernel<G, W, WIDE>), dim3(grid), dim3(64), lds, st, B);
    HP_HIP_CHECK(hipGetLastError());
    return HP_OK;
}

}  // namespace

// A batch laid out and uploaded once (prepare) can be aligned any number of times (run): what the block-level resident
// form (hp_block.hip) and the bench time is run(), with the sequences already in HBM.
struct W2Session {
    const hp_wfa_job* jobs = nullptr;    // generic mode (hp_wfa_assign_batch): the caller's jobs
    const hp_block_input* bl_in = nullptr;   // block mode (hp_block.hip): jobs are records of these blocks
    const W2JobIn* bl_jobs = nullptr;
    size_t n = 0;
    int device_id = -1;
    DevBuf d_qhead;                      // work-queue heads, class counts, hand-over words of the last run
    DevBuf d_packed, d_src_off, d_fmt;   // block mode: the reads as the caller holds them (ASCII or BAM 4-bit), expanded by hp_wfa2_unpack_kernel
    bool need_unpack = false;            // block mode: the reads are in d_packed and have not been expanded into d_seq yet
    uint64_t h2d_bytes = 0;              // of the last prepare
    double prep_ms[4] = {0, 0, 0, 0};    // of the last prepare: layout, fill + upload, total, -
    // job i as the dense-band path takes it (generic mode: the caller's; block mode: assembled from the block, a BAM 4-bit read
    // decoded on the host - a handful of jobs per batch)
    hp_wfa_job job_header(size_t i) const {   // job i's window and variant lists (no read)
        if (jobs) return jobs[i];
        const W2JobIn& ji = bl_jobs[i];
        const hp_block_input& B = bl_in[ji.block];
        hp_wfa_job j{};
        j.hets = B.hets + ji.het_first; j.n_hets = ji.n_hets;
        j.homs = ji.n_homs ? B.homs + ji.hom_first : nullptr; j.n_homs = ji.n_homs;
        return j;
    }
    hp_wfa_job materialize(size_t i, std::vector<std::vector<uint8_t>>& ascii_scratch) {
        if (jobs) return jobs[i];
        const W2JobIn& ji = bl_jobs[i];
        const hp_block_input& B = bl_in[ji.block];
        const hp_block_record& rec = B.records[ji.rec];
        hp_wfa_job j{};
        j.reference = B.reference; j.ref_base = B.ref_base;
        j.ref_start = (uint64_t)rec.min_position; j.ref_end = (uint64_t)rec.max_position + 1;   // read_parsing.rs:772-773
        j.hets = B.hets + ji.het_first; j.n_hets = ji.n_hets;
        j.homs = ji.n_homs ? B.homs + ji.hom_first : nullptr; j.n_homs = ji.n_homs;
        j.read_len = rec.read_len;
        if (B.seq_format == HP_SEQ_BAM4) {
            ascii_scratch.emplace_back((size_t)rec.read_len + 1);
            decode_bam4(rec.read_align, rec.read_offset, rec.read_len, ascii_scratch.back().data());
            j.read = ascii_scratch.back().data();
        } else j.read = rec.read_align + rec.read_offset;
        return j;
    }
    std::vector<W2Job> dj;
    std::vector<W2Variant> vars;
    std::vector<uint32_t> len_order;   // job ids, longest read first (stable)
    uint64_t seq_bytes = 0, alt_off = 0, node_tot = 0, edge_tot = 0, tag_tot = 0, allele_tot = 0;
    DevBuf d_seq, d_vars, d_jobs, d_nodes, d_edges, d_tags, d_info, d_order, d_len_order, d_cls, d_blockcnt, d_sets, d_score, d_status, d_alleles, d_work;
    double last_prepare_ms = 0.0;
    double last_span_ms = 0.0;   // of the last run: first class launch .. last class kernel done (the three run concurrently)
    uint64_t work_updates = 0, work_node_bytes = 0, work_read_bytes = 0, work_jobs = 0;   // of the last run (compact kernel only)
    // leftovers of the last run (jobs the dense-band path aligns): their pass runs on the context's helper thread; with
    // defer = true run() returns while it is still going and finish() waits for it (hp_block.hip assembles the blocks
    // that do not hold a leftover read in the meantime)
    struct Pending {
        bool on = false, posted = false, two_phase = false;
        std::vector<uint32_t> ids;      // every job whose result finish() delivers: held + big
        std::vector<uint32_t> held;     // still with the largest class's kernel when run() collected the others
        std::vector<uint32_t> held_nodes;   // their graphs' node counts (hp_wfa_result::n_nodes)
        std::vector<uint32_t> big;      // for the dense-band pass
        std::vector<uint32_t> big_ed;   // the edit distance each of them had reached when the compact kernel let go of it
        std::vector<uint32_t> big_nodes; // their graphs' node counts (0: the device builder left the graph to the host) | the limit that ended the compact attempt (hp_wfa2_kernel's `why`) << 24
        static uint32_t nodes_of(uint32_t x) { return x & 0xFFFFFFu; }
        static uint32_t why_of(uint32_t x) { return x >> 24; }
        hp_wfa_result* dst = nullptr;
        uint8_t* const* alleles = nullptr;
        uint64_t prune = 0, max_ed = 0;
        hipStream_t stream2 = nullptr;  // the largest class's stream
        W2Batch b2{};                   // the largest class's launch of this run (its scratch region; a second tag range)
        uint32_t large_groups = 0;
        int n_cu = 0;
        float ms_build = 0.f;
        int rc = HP_OK;
        std::string err;
    } pend;
    // what a dense-band pass over leftovers needs of its own (the late pass and the early pass of one set may run side by side)
    struct SubWork {
        std::vector<hp_wfa_job> sub;
        std::vector<hp_wfa_result> sub_out;
        std::vector<uint8_t*> sub_al;
        std::vector<std::vector<uint8_t>> ascii;   // block mode: decoded reads of the jobs that take the dense-band pass
    } late_sw;
    // The leftovers' way out (everything in `big`): the reference-window test (hp_wfa2_bound_kernel), then the dense-band pass for what
    // it leaves; results straight into dst / alleles. beside_launch_set: the test runs one wavefront per job and compares in place - a
    // workgroup that fits the slot a compute unit has free beside a resident launch set.
    int leftovers_out(std::vector<uint32_t>& big, std::vector<uint32_t>& big_ed, std::vector<uint32_t>& big_nodes, SubWork& sw, hp_wfa_result* dst,
                      uint8_t* const* alleles, uint64_t prune, uint64_t max_ed, bool beside_launch_set, double* t_bound, size_t* n_settled, double* kernel_ms);
    // Records the layout routed past the compact kernels (layout_blocks: by their CIGARs they are heading for the neighbourhood of
    // max_edit_distance - the noisy tail that outgrows every class's lists a dozen rounds in and was, until round 5, only sent on
    // its way out AFTER the class kernels: a chain of 20-40 ms behind the first collection that the set's rows waited for). Their
    // way out starts when the launch set does, on the device's early worker (early_pass); finish() joins it.
    struct Early {
        bool on = false, done = true;
        int rc = HP_OK;
        std::string err;
        std::vector<uint32_t> ids;         // ascending
        std::vector<uint8_t> mask;         // [n] 1 = routed
        std::vector<uint32_t> big, big_ed, big_nodes;
        SubWork sw;
        hp_wfa_result* dst = nullptr;
        uint8_t* const* alleles = nullptr;
        uint64_t prune = 0, max_ed = 0;
        double t_post = 0, t_start = 0, t_bound = 0, t_done = 0;
        size_t n_settled = 0;
    } early;
    std::mutex em;
    std::condition_variable ecv;
    std::atomic<bool> aligning{false};   // counted in sets_aligning(device_id)
    void aligning_done() { if (aligning.exchange(false)) sets_aligning(device_id).fetch_sub(1); }
    void early_pass();
    int wait_early() {
        std::unique_lock<std::mutex> lk(em);
        ecv.wait(lk, [this]() { return early.done; });
        const bool was_on = early.on;
        early.on = false;
        if (was_on && early.rc != HP_OK) { set_error("%s", early.err.c_str()); return early.rc; }
        return HP_OK;
    }
    int finish_late();
    DevBuf d_job_cls, d_handed, d_seen, d_held, d_hoff, d_hrec, d_hrows, d_wide, d_wide_sets, d_wide_hash;
    uint32_t wide_tag_next = 0;
    int wide_last_gen = 0;
    PinBuf late_down;                      // results of the held jobs
    PinBuf down;                           // results of a run's first collection (the session's own: the calling thread may have queued the next set's run before they are read)
    std::unique_ptr<HelperThread> helper;  // runs late() when run() defers
    // what a run's first collection needs once its results are on the host (collect_host, scatter): kept here because with
    // defer = 2 the thread that launched the set does not wait for them (see run())
    struct RunState {
        size_t dn_score = 0, dn_work = 0, dn_info = 0, dn_cnt = 0, dn_al = 0;
        hp_wfa_result* out = nullptr;
        uint8_t* const* alleles = nullptr;
        bool two_phase = false, verbose = false;
        uint64_t prune = 0, max_ed = 0;
        hipStream_t stream2 = nullptr;
        W2Batch b2{};
        uint32_t large_groups = 0, groups_used[3] = {0, 0, 0};
        int n_cu = 0, defer = 0;
        double t0 = 0, t_built = 0, t_cls = 0, t_done = 0;
    } rs;
    hipEvent_t ev_c = nullptr;             // the first collection's copies are done
    bool async_inflight = false;           // run() returned before its first collection: the helper thread collects, then runs late()
    std::mutex cm;
    std::condition_variable ccv;
    bool collected = false, scattered = false;
    int collect_rc = HP_OK;
    std::string collect_err;
    int collect_host();                    // the first collection's results sorted out on the host: what is still held, what goes to the dense band
    int scatter();                         // ... and handed to the caller's arrays
    int wait_collected();                  // defer = 2: waits for collect_host (helper thread), then scatters on the calling thread
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // build start / end, class launches start / end
    int late();                            // second collection + dense-band pass; on the helper thread when deferred
    double late_kernel_ms = 0.0;
    std::mutex work_m;
    int prepare(const hp_wfa_job* jobs_, size_t n_, int device);
    // block mode in two halves: layout_blocks is host only (offsets, the in-place runs, the length order: a block stream runs it in its
    // layout stage, ahead of the set's turn on the PCIe link), upload_blocks fills the tables and sends everything; prepare_blocks = both
    struct BlockLay { int64_t lo = INT64_MAX, hi = INT64_MIN; uint64_t ref_dev = 0, pool_off = 0; uint32_t var_base = 0; };
    struct Run { uintptr_t lo, hi; uint64_t dev; uintptr_t range; };   // range: start of the hp_host_alloc range that owns [lo, hi) - a hull / run never leaves it
    struct Lay {
        bool valid = false, in_place = false;
        size_t n_in = 0;
        std::vector<BlockLay> bl;
        std::vector<uint64_t> src_off;
        std::vector<uint8_t> fmt;
        std::vector<Run> runs;
        std::vector<uint32_t> suspects;   // jobs routed past the compact kernels (ascending ids; fmt bit W2_FMT_SUSPECT)
        uint64_t n_vars = 0, pool_bytes = 0, ref_bytes = 0, reads_dev = 0, dev_reads = 0, packed = 0, in_place_bytes = 0;
        double t0 = 0.0, t_lay = 0.0;
    } lay;
    int layout_blocks(const hp_block_input* in, size_t n_in, const W2JobIn* jin, size_t n_);
    int upload_blocks(int device);
    int prepare_blocks(const hp_block_input* in, size_t n_in, const W2JobIn* jin, size_t n_, int device) {
        const int rc = layout_blocks(in, n_in, jin, n_);
        return rc != HP_OK ? rc : upload_blocks(device);
    }
    // defer: 0 = everything is in `out` on return; 1 = the dense-band pass of the leftovers may still run (finish() waits);
    // 2 = also the largest class's kernel (two phases) - the caller must not start another run on this thread before finish()
    int run(uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* const* alleles, int defer = 0);
    int finish();
    ~W2Session() {
        if ((async_inflight || (pend.on && pend.posted)) && helper) helper->wait();
        { std::unique_lock<std::mutex> lk(em); ecv.wait(lk, [this]() { return early.done; }); }
        aligning_done();
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (ev_c) (void)hipEventDestroy(ev_c);
    }
};

int W2Session::prepare(const hp_wfa_job* jobs_, size_t n_, int device) {
    // a deferred late() of the previous run still writes the caller's result arrays and reads the tables laid out below: join it
    // first, whatever path the caller took out of that run (its status belongs to that run, not to this one)
    if (pend.on || async_inflight || early.on) (void)finish();
    jobs = jobs_; n = n_; bl_in = nullptr; bl_jobs = nullptr; need_unpack = false;
    if (n == 0) return HP_OK;
    if (!jobs) { set_error("null argument"); return HP_ERR_ARG; }
    if (n > 0x3FFFFFFFull) { set_error("too many jobs"); return HP_ERR_ARG; }
    const double t0 = w2_now_ms();
    for (size_t i = 0; i < n; ++i) {
        const hp_wfa_job& j = jobs[i];
        if (!j.reference || (!j.read && j.read_len)) { set_error("job %zu: null sequence", i); return HP_ERR_ARG; }
        if (j.ref_end < j.ref_start || j.ref_start < j.ref_base) { set_error("bad reference window"); return HP_ERR_ARG; }
        if ((j.n_hets && !j.hets) || (j.n_homs && !j.homs)) { set_error("job %zu: null variant list", i); return HP_ERR_ARG; }
        if (j.ref_end - j.ref_start >= 0x7FFFFFFFull) { set_error("job %zu: reference window too long", i); return HP_ERR_UNSUPPORTED; }
    }
    seq_bytes = 0; node_tot = edge_tot = tag_tot = allele_tot = 0;

    // ---- 1. layout: merged reference windows, distinct variants, read bases ------------------------------------------
    std::vector<Range> ref_ranges, var_ranges;
    {
        std::vector<std::pair<const uint8_t*, uint64_t>> iv;
        iv.reserve(n);
        for (size_t i = 0; i < n; ++i)
            if (jobs[i].ref_end > jobs[i].ref_start) iv.push_back({jobs[i].reference + (jobs[i].ref_start - jobs[i].ref_base), jobs[i].ref_end - jobs[i].ref_start});
        merge_ranges(iv, ref_ranges, 1);
        iv.clear();
        for (size_t i = 0; i < n; ++i) {
            if (jobs[i].n_hets) iv.push_back({reinterpret_cast<const uint8_t*>(jobs[i].hets), (uint64_t)jobs[i].n_hets * sizeof(hp_wfa_variant)});
            if (jobs[i].n_homs) iv.push_back({reinterpret_cast<const uint8_t*>(jobs[i].homs), (uint64_t)jobs[i].n_homs * sizeof(hp_wfa_variant)});
        }
        merge_ranges(iv, var_ranges, sizeof(hp_wfa_variant));
    }
    uint64_t n_vars = 0;
    for (auto& r : ref_ranges) { r.dev = seq_bytes; seq_bytes += ((uint64_t)(r.hi - r.lo) + 15) & ~15ull; }
    for (auto& r : var_ranges) { r.dev = n_vars; n_vars += (uint64_t)(r.hi - r.lo) / sizeof(hp_wfa_variant); }
    if (n_vars >= 0xFFFFFFF0ull) { set_error("too many variants"); return HP_ERR_UNSUPPORTED; }
    // allele pool: every distinct variant's truncated alleles once
    vars.assign((size_t)n_vars, W2Variant{});
    uint64_t pool_bytes = 0;
    for (auto& r : var_ranges) {
        const hp_wfa_variant* hv = reinterpret_cast<const hp_wfa_variant*>(r.lo);
        const size_t cnt = (size_t)(r.hi - r.lo) / sizeof(hp_wfa_variant);
        for (size_t k = 0; k < cnt; ++k) {
            W2Variant& w = vars[(size_t)r.dev + k];
            w.position = hv[k].position; w.ref_len = hv[k].ref_len; w.flags = hv[k].flags;
            w.a0_off = w.a0_len = 0;
            if (hv[k].flags & 2u) {
                if (!hv[k].allele0 && hv[k].allele0_len) { set_error("variant with null allele0"); return HP_ERR_ARG; }
                w.a0_off = (uint32_t)pool_bytes; w.a0_len = hv[k].allele0_len; pool_bytes += hv[k].allele0_len;
            }
            if (!hv[k].allele1 && hv[k].allele1_len) { set_error("variant with null allele1"); return HP_ERR_ARG; }
            w.a1_off = (uint32_t)pool_bytes; w.a1_len = hv[k].allele1_len; pool_bytes += hv[k].allele1_len;
            if (pool_bytes >= 0xFFFFFF00ull) { set_error("allele pool exceeds 4 GiB"); return HP_ERR_UNSUPPORTED; }
        }
    }
    alt_off = seq_bytes;
    seq_bytes += (pool_bytes + 15) & ~15ull;
    dj.assign(n, W2Job{});
    for (size_t i = 0; i < n; ++i) {
        const hp_wfa_job& j = jobs[i];
        W2Job& d = dj[i];
        d = W2Job{};
        d.ref_start = (int64_t)j.ref_start;
        d.ref_len = (uint32_t)(j.ref_end - j.ref_start);
        if (d.ref_len) {
            const uint8_t* p = j.reference + (j.ref_start - j.ref_base);
            auto it = std::upper_bound(ref_ranges.begin(), ref_ranges.end(), p, [](const uint8_t* q, const Range& r) { return q < r.lo; });
            --it;
            d.ref_off = it->dev + (uint64_t)(p - it->lo);
        }
        auto var_index = [&](const hp_wfa_variant* p, uint32_t cnt, uint32_t& first) -> bool {
            first = 0;
            if (cnt == 0) return true;
            const uint8_t* q = reinterpret_cast<const uint8_t*>(p);
            auto it = std::upper_bound(var_ranges.begin(), var_ranges.end(), q, [](const uint8_t* a, const Range& r) { return a < r.lo; });
            while (it != var_ranges.begin()) {   // ranges that could not be merged may nest: find the one holding the slice
                --it;
                if (q >= it->lo && q + (uint64_t)cnt * sizeof(hp_wfa_variant) <= it->hi && (uint64_t)(q - it->lo) % sizeof(hp_wfa_variant) == 0) {
                    first = (uint32_t)(it->dev + (uint64_t)(q - it->lo) / sizeof(hp_wfa_variant));
                    return true;
                }
            }
            return false;
        };
        if (!var_index(j.hets, j.n_hets, d.het_first) || !var_index(j.homs, j.n_homs, d.hom_first)) { set_error("internal: variant slice not found"); return HP_ERR_INVARIANT; }
        d.n_hets = j.n_hets; d.n_homs = j.n_homs;
        d.read_off = seq_bytes; d.read_len = j.read_len;
        seq_bytes += ((uint64_t)j.read_len + 15) & ~15ull;
        const uint64_t V = (uint64_t)j.n_hets + j.n_homs;
        const uint64_t ncap = 5 * V + 2, ecap = 2 * ncap, tcap = 2 * (uint64_t)j.n_hets + 2;   // (ecap: u16 units of the overflow list)
        if (node_tot + ncap >= 0xFFFFFFF0ull || edge_tot + ecap >= 0xFFFFFFF0ull || allele_tot + j.n_hets >= 0xFFFFFFF0ull) { set_error("batch too large"); return HP_ERR_UNSUPPORTED; }
        d.node_off = (uint32_t)node_tot; d.node_cap = (uint32_t)ncap; node_tot += ncap;
        d.edge_off = (uint32_t)edge_tot; d.edge_cap = (uint32_t)ecap; edge_tot += ecap;
        d.tag_off = (uint32_t)tag_tot; d.tag_cap = (uint32_t)tcap; tag_tot += tcap;
        d.allele_off = (uint32_t)allele_tot; allele_tot += j.n_hets;
    }
    seq_bytes += 256;   // the 32-byte compares may run past the last base of the last read
    len_order.resize(n);
    std::iota(len_order.begin(), len_order.end(), 0u);
    std::stable_sort(len_order.begin(), len_order.end(), [&](uint32_t a, uint32_t b) { return jobs[a].read_len > jobs[b].read_len; });


    // ---- from here on a GPU is mandatory (no CPU fallback) -------------------------------------------------------------
    if (device < 0) device = hp_default_device();
    device_id = device;
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed - no usable GPU; there is no CPU fallback", device_id); return HP_ERR_HIP; }
    W2Context& cx = w2_context(device_id);
    W2Context::Streams* cs_ = nullptr;
    { const int rc0 = cx.streams(g_cu_partition, &cs_, false); if (rc0 != HP_OK) return rc0; }
    hipStream_t st = cs_->stream;
    struct StreamDrain { hipStream_t s; ~StreamDrain() { (void)hipStreamSynchronize(s); } } drain{st};

    // ---- 2. stage + upload -----------------------------------------------------------------------------------------------
    int rc;
    if ((rc = cx.stage.reserve(seq_bytes)) != HP_OK) return rc;
    {
        uint8_t* sb = cx.stage.p;
        uint64_t ref_total = 0;
        for (auto& r : ref_ranges) ref_total += (uint64_t)(r.hi - r.lo);
        uint64_t read_total = 0;
        for (size_t i = 0; i < n; ++i) read_total += jobs[i].read_len;
        const unsigned nt = w2_host_threads((size_t)((ref_total + read_total) >> 16) + 1, 16);
        w2_parallel(nt, [&](unsigned t, unsigned T) {
            constexpr uint64_t PIECE = 256u << 10;
            uint64_t piece = 0;
            for (const Range& r : ref_ranges) {
                const uint64_t len = (uint64_t)(r.hi - r.lo);
                for (uint64_t o = 0; o < len; o += PIECE, ++piece)
                    if (piece % T == t) std::memcpy(sb + r.dev + o, r.lo + o, (size_t)std::min(PIECE, len - o));
            }
            for (size_t i = n * t / T; i < n * (t + 1) / T; ++i)
                if (jobs[i].read_len) std::memcpy(sb + dj[i].read_off, jobs[i].read, jobs[i].read_len);
            if (t == 0) {
                for (auto& r : var_ranges) {
                    const hp_wfa_variant* hv = reinterpret_cast<const hp_wfa_variant*>(r.lo);
                    const size_t cnt = (size_t)(r.hi - r.lo) / sizeof(hp_wfa_variant);
                    for (size_t k = 0; k < cnt; ++k) {
                        const W2Variant& w = vars[(size_t)r.dev + k];
                        if (w.a0_len) std::memcpy(sb + alt_off + w.a0_off, hv[k].allele0, w.a0_len);
                        if (w.a1_len) std::memcpy(sb + alt_off + w.a1_off, hv[k].allele1, w.a1_len);
                    }
                }
                std::memset(sb + seq_bytes - 256, 0, 256);
            }
        });
    }
    if ((rc = d_seq.alloc(seq_bytes)) || (rc = d_vars.alloc(std::max<size_t>(1, vars.size()) * sizeof(W2Variant))) || (rc = d_jobs.alloc(n * sizeof(W2Job))) ||
        (rc = d_nodes.alloc((size_t)node_tot * sizeof(W2Node))) || (rc = d_edges.alloc((size_t)edge_tot * 2)) || (rc = d_tags.alloc((size_t)tag_tot * 4)) ||
        (rc = d_info.alloc(n * sizeof(W2Info))) || (rc = d_order.alloc(n * 12)) || (rc = d_len_order.alloc(n * 4)) || (rc = d_cls.alloc(n + 16)) || (rc = d_job_cls.alloc(n + 16)) || (rc = d_handed.alloc(n + 16)) || (rc = d_seen.alloc(n * 4 + 16)) || (rc = d_blockcnt.alloc(((n + 255) / 256 + 1) * 16)) || (rc = d_sets.alloc(n * W2_SET_STRIDE * 4)) ||
        (rc = d_score.alloc(n * 8)) || (rc = d_work.alloc(n * 8 + 16)) || (rc = d_status.alloc(n * 4)) || (rc = d_alleles.alloc(std::max<uint64_t>(allele_tot, 16))))
        return rc;
    HP_HIP_CHECK(hipMemcpyAsync(d_seq.p, cx.stage.p, seq_bytes, hipMemcpyHostToDevice, st));
    // the small tables are pageable std::vectors that live until the end of this function; every stream operation that
    // reads them is waited for below (hipStreamSynchronize) before they go out of scope
    if (!vars.empty()) HP_HIP_CHECK(hipMemcpyAsync(d_vars.p, vars.data(), vars.size() * sizeof(W2Variant), hipMemcpyHostToDevice, st));
    HP_HIP_CHECK(hipMemcpyAsync(d_jobs.p, dj.data(), n * sizeof(W2Job), hipMemcpyHostToDevice, st));
    HP_HIP_CHECK(hipMemcpyAsync(d_len_order.p, len_order.data(), n * 4, hipMemcpyHostToDevice, st));

    if (hipStreamSynchronize(st) != hipSuccess) { set_error("upload failed"); return HP_ERR_HIP; }
    last_prepare_ms = w2_now_ms() - t0;
    return HP_OK;
}

// Block mode. What prepare() finds out by sorting and merging address ranges is known here by construction: a block's jobs
// read windows of ONE reference buffer (their hull is uploaded once) and slices of the block's two variant vectors (uploaded
// once, hets then homs). Everything per job is independent of every other job, so layout, table fill and the staging copy
// run on host threads; the reads are staged in pieces and each piece's DMA runs while the next one is being filled.
int W2Session::layout_blocks(const hp_block_input* in, size_t n_in, const W2JobIn* jin, size_t n_) {
    if (pend.on || async_inflight || early.on) (void)finish();   // (as in prepare(): never lay a set out under a late pass that is still running)
    jobs = nullptr; bl_in = in; bl_jobs = jin; n = n_;
    h2d_bytes = 0; prep_ms[0] = prep_ms[1] = prep_ms[2] = prep_ms[3] = 0.0;
    lay.valid = false; lay.n_in = n_in;
    if (n == 0) { lay.valid = true; return HP_OK; }
    if (!in || !jin) { set_error("null argument"); return HP_ERR_ARG; }
    if (n > 0x3FFFFFFFull) { set_error("too many jobs"); return HP_ERR_ARG; }
    const double t0 = w2_now_ms();
    seq_bytes = 0; node_tot = edge_tot = tag_tot = allele_tot = 0;
    // ---- 1. per block: hull of its jobs' windows, variant base, allele pool -------------------------------------------------
    std::vector<BlockLay>& bl = lay.bl;
    bl.assign(n_in, BlockLay{});
    for (size_t i = 0; i < n; ++i) {
        const W2JobIn& ji = jin[i];
        if (ji.block >= n_in || ji.rec >= in[ji.block].n_records) { set_error("job %zu: bad block / record index", i); return HP_ERR_ARG; }
        const hp_block_record& rec = in[ji.block].records[ji.rec];
        if (!rec.read_align && rec.read_len) { set_error("job %zu: null sequence", i); return HP_ERR_ARG; }
        if (rec.max_position - rec.min_position >= 0x7FFFFFF0ll) { set_error("job %zu: reference window too long", i); return HP_ERR_UNSUPPORTED; }
        BlockLay& L = bl[ji.block];
        L.lo = std::min(L.lo, rec.min_position); L.hi = std::max(L.hi, rec.max_position + 1);
    }
    uint64_t n_vars = 0, pool_bytes = 0, ref_bytes = 0;
    for (size_t b = 0; b < n_in; ++b) {
        BlockLay& L = bl[b];
        if (L.lo > L.hi) continue;   // no job of this block
        const hp_block_input& B = in[b];
        if (B.seq_format != HP_SEQ_ASCII && B.seq_format != HP_SEQ_BAM4) { set_error("block %zu: unknown seq_format %u", b, B.seq_format); return HP_ERR_ARG; }
        L.ref_dev = ref_bytes; ref_bytes += ((uint64_t)(L.hi - L.lo) + 15) & ~15ull;
        L.var_base = (uint32_t)n_vars; n_vars += (uint64_t)B.n_hets + B.n_homs;
        L.pool_off = pool_bytes;
        for (int pass = 0; pass < 2; ++pass) {
            const hp_wfa_variant* hv = pass ? B.homs : B.hets;
            const uint32_t cnt = pass ? B.n_homs : B.n_hets;
            for (uint32_t k = 0; k < cnt; ++k) {
                if ((hv[k].flags & 2u) && !hv[k].allele0 && hv[k].allele0_len) { set_error("variant with null allele0"); return HP_ERR_ARG; }
                if (!hv[k].allele1 && hv[k].allele1_len) { set_error("variant with null allele1"); return HP_ERR_ARG; }
                pool_bytes += ((hv[k].flags & 2u) ? hv[k].allele0_len : 0u) + (uint64_t)hv[k].allele1_len;
            }
        }
        if (n_vars >= 0xFFFFFFF0ull || pool_bytes >= 0xFFFFFF00ull) { set_error("too many variants / allele pool exceeds 4 GiB"); return HP_ERR_UNSUPPORTED; }
    }
    alt_off = ref_bytes;
    const uint64_t reads_dev = ref_bytes + ((pool_bytes + 15) & ~15ull);   // device: [reference hulls][allele pool][one byte per base][pad]
    // ---- 2. per job: offsets (serial prefix sums, a few bytes per job) ------------------------------------------------------------
    dj.assign(n, W2Job{});
    std::vector<uint64_t>& src_off = lay.src_off;
    std::vector<uint8_t>& fmt = lay.fmt;
    src_off.assign(n + 1, 0); fmt.assign(n, 0);
    uint64_t dev_reads = 0, packed = 0;
    // Which records are heading for the neighbourhood of max_edit_distance? The record's own CIGAR says (the view local re-alignment
    // reads, when the caller handed one over): an alignment operation begins every 1 / (2 x indel rate) bases - one op in 150 at HiFi
    // error rates, one in 15 on a read with 5 % noise (with = / X CIGARs the substitutions count too). Such a read holds fifty
    // diagonals per node, outgrows every class's lists a dozen rounds in and ends in the reference-window test and the dense band
    // anyway (hp_wfa3_kernel's `hopeless`): routed there NOW, its way out runs beside the set's launch set instead of behind it
    // (W2Session::Early). Routing only - every road computes the same result. HP_WFA2_SUSPECT_OPS: ops per 1 000 bases from which a
    // record is routed (50; 0 = never); more than HP_WFA2_SUSPECT_MAX of them in a set (1 024: a noisy SET is the wide-table
    // launch's business, late()) and nobody is.
    const uint32_t sus_ops = [] { const char* e = std::getenv("HP_WFA2_SUSPECT_OPS"); return (uint32_t)(e ? std::max(0, std::atoi(e)) : 50); }();   // (read per set: a dozen nanoseconds, and a test can switch it)
    const size_t sus_max = [] { const char* e = std::getenv("HP_WFA2_SUSPECT_MAX"); return (size_t)(e ? std::max(0, std::atoi(e)) : 1024); }();
    std::vector<uint32_t>& suspects = lay.suspects;
    suspects.clear();
    for (size_t i = 0; i < n; ++i) {
        const W2JobIn& ji = jin[i];
        const hp_block_input& B = in[ji.block];
        const hp_block_record& rec = B.records[ji.rec];
        if (sus_ops && rec.local && rec.local->n_cigar >= 100u && (uint64_t)rec.local->n_cigar * 1000u >= (uint64_t)sus_ops * rec.read_len) suspects.push_back((uint32_t)i);
        W2Job& d = dj[i];
        d.read_off = reads_dev + dev_reads; d.read_len = rec.read_len;
        dev_reads += ((uint64_t)rec.read_len + 15) & ~15ull;
        const uint32_t odd = B.seq_format == HP_SEQ_BAM4 ? (rec.read_offset & 1u) : 0u;
        const uint64_t pb = B.seq_format == HP_SEQ_BAM4 ? ((uint64_t)odd + rec.read_len + 1) / 2 : rec.read_len;
        src_off[i] = packed; fmt[i] = (uint8_t)(B.seq_format | (odd << 4));
        packed += ((pb + 15) & ~15ull) + 16;   // (the expansion reads 16 bytes at a time)
        const uint64_t V = (uint64_t)ji.n_hets + ji.n_homs;
        const uint64_t ncap = 5 * V + 2, ecap = 2 * ncap, tcap = 2 * (uint64_t)ji.n_hets + 2;
        if (node_tot + ncap >= 0xFFFFFFF0ull || edge_tot + ecap >= 0xFFFFFFF0ull || allele_tot + ji.n_hets >= 0xFFFFFFF0ull) { set_error("batch too large"); return HP_ERR_UNSUPPORTED; }
        d.node_off = (uint32_t)node_tot; d.node_cap = (uint32_t)ncap; node_tot += ncap;
        d.edge_off = (uint32_t)edge_tot; d.edge_cap = (uint32_t)ecap; edge_tot += ecap;
        d.tag_off = (uint32_t)tag_tot; d.tag_cap = (uint32_t)tcap; tag_tot += tcap;
        d.allele_off = (uint32_t)allele_tot; allele_tot += ji.n_hets;
    }
    src_off[n] = packed;
    if (suspects.size() > sus_max) suspects.clear();
    for (uint32_t i : suspects) fmt[i] |= (uint8_t)W2_FMT_SUSPECT;
    // Do the records' bases all lie in host memory the copy engines read in place (hp_host_alloc)? Then nothing of them is staged:
    // the blocks' address hulls, merged where they touch, cross PCIe as they are - one DMA per run of blocks - and a record's source
    // offset is its place in that image. (Blocks gathered one after the other into an arena give a handful of runs; records
    // scattered so that the hulls hold much more than the records do take the staged way.)
    std::vector<Run>& runs = lay.runs;
    runs.clear();
    bool in_place = host_ranges_any() && !std::getenv("HP_NO_IN_PLACE");
    uint64_t in_place_bytes = 0;
    if (in_place) {
        uintptr_t r_lo = 0, r_hi = 0;   // the hp_host_alloc range the last record lay in
        uint64_t payload = 0;
        std::vector<Run> hull;          // per block
        size_t cur_block = (size_t)-1;
        for (size_t i = 0; i < n && in_place; ++i) {
            const hp_block_input& B = in[jin[i].block];
            const hp_block_record& rec = B.records[jin[i].rec];
            if (!rec.read_len) continue;
            const uint8_t* p = B.seq_format == HP_SEQ_BAM4 ? rec.read_align + (rec.read_offset >> 1) : rec.read_align + rec.read_offset;
            const uint64_t pb = B.seq_format == HP_SEQ_BAM4 ? ((uint64_t)(rec.read_offset & 1u) + rec.read_len + 1) / 2 : rec.read_len;
            const uintptr_t a = (uintptr_t)p, e = a + pb + 24;   // (the expansion reads 16 bytes at a time, the odd-nibble shift 8 further)
            if (!(a >= r_lo && e <= r_hi) && !(host_range_of(p, &r_lo, &r_hi) && e <= r_hi)) { in_place = false; break; }
            // (several arenas - one per worker thread, and the dispatcher merges blocks of many callers into one set: a hull only grows
            // inside the pinned range its records lie in, so the DMA below never reads the gap between two allocations)
            if (jin[i].block == cur_block && hull.back().range == r_lo && a + (1u << 20) >= hull.back().lo && e <= hull.back().hi + (1u << 20)) {
                hull.back().lo = std::min(hull.back().lo, a); hull.back().hi = std::max(hull.back().hi, e);
            } else {   // (a record far from its block's others, or in another arena, opens a hull of its own)
                hull.push_back(Run{a, e, 0, r_lo});
                cur_block = jin[i].block;
            }
            payload += pb;
        }
        if (in_place) {
            std::sort(hull.begin(), hull.end(), [](const Run& x, const Run& y) { return x.lo < y.lo; });
            for (const Run& h : hull) {
                if (!runs.empty() && h.range == runs.back().range && h.lo <= runs.back().hi + 4096) runs.back().hi = std::max(runs.back().hi, h.hi);
                else runs.push_back(h);
            }
            uint64_t image = 0;
            for (Run& r : runs) { r.dev = image; image += ((r.hi - r.lo) + 15) & ~(uint64_t)15; }
            if (runs.size() > 4096 || image > payload + payload / 4 + (1u << 20)) in_place = false;   // too scattered: stage
            else {
                for (size_t i = 0; i < n; ++i) {
                    const hp_block_input& B = in[jin[i].block];
                    const hp_block_record& rec = B.records[jin[i].rec];
                    if (!rec.read_len) { src_off[i] = 0; continue; }
                    const uintptr_t a = (uintptr_t)(B.seq_format == HP_SEQ_BAM4 ? rec.read_align + (rec.read_offset >> 1) : rec.read_align + rec.read_offset);
                    auto it = std::upper_bound(runs.begin(), runs.end(), a, [](uintptr_t v, const Run& r) { return v < r.lo; });
                    --it;
                    src_off[i] = it->dev + (a - it->lo);
                }
                packed = image; src_off[n] = packed;
                in_place_bytes = image;
            }
        }
        if (!in_place) runs.clear();
    }
    seq_bytes = reads_dev + dev_reads + 256;
    // longest read first (stable counting sort; reads beyond 64 k bases share the first bucket - the order only steers the work queues)
    len_order.resize(n);
    {
        std::vector<uint32_t> cnt(65537, 0);
        for (size_t i = 0; i < n; ++i) cnt[65535u - std::min<uint32_t>(dj[i].read_len, 65535u) + 1u]++;
        for (size_t k = 1; k <= 65536; ++k) cnt[k] += cnt[k - 1];
        for (size_t i = 0; i < n; ++i) len_order[cnt[65535u - std::min<uint32_t>(dj[i].read_len, 65535u)]++] = (uint32_t)i;
    }
    vars.clear();
    lay.in_place = in_place; lay.n_vars = n_vars; lay.pool_bytes = pool_bytes; lay.ref_bytes = ref_bytes; lay.reads_dev = reads_dev;
    lay.dev_reads = dev_reads; lay.packed = packed; lay.in_place_bytes = in_place_bytes;
    lay.t0 = t0; lay.t_lay = w2_now_ms();
    prep_ms[0] = lay.t_lay - t0;
    lay.valid = true;
    return HP_OK;
}

int W2Session::upload_blocks(int device) {
    if (!lay.valid) { set_error("upload_blocks without layout_blocks"); return HP_ERR_ARG; }
    lay.valid = false;
    if (n == 0) return HP_OK;
    const hp_block_input* in = bl_in;
    const W2JobIn* jin = bl_jobs;
    const size_t n_in = lay.n_in;
    const std::vector<BlockLay>& bl = lay.bl;
    const std::vector<uint64_t>& src_off = lay.src_off;
    const std::vector<uint8_t>& fmt = lay.fmt;
    const std::vector<Run>& runs = lay.runs;
    const bool in_place = lay.in_place;
    const uint64_t n_vars = lay.n_vars, ref_bytes = lay.ref_bytes, reads_dev = lay.reads_dev, packed = lay.packed, in_place_bytes = lay.in_place_bytes;
    const double t_up0 = w2_now_ms();

    // ---- from here on a GPU is mandatory (no CPU fallback) -------------------------------------------------------------
    if (device < 0) device = hp_default_device();
    device_id = device;
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed - no usable GPU; there is no CPU fallback", device_id); return HP_ERR_HIP; }
    W2Context& cx = w2_context(device_id);
    W2Context::Streams* cs_ = nullptr;
    { const int rc0 = cx.streams(g_cu_partition, &cs_, false); if (rc0 != HP_OK) return rc0; }
    hipStream_t st = cs_->stream;
    struct StreamDrain { hipStream_t s; ~StreamDrain() { (void)hipStreamSynchronize(s); } } drain{st};
    int rc;
    // pinned staging: [reference hulls][allele pool][W2Variant][W2Job][len_order][src_off][fmt][reads as the caller holds them]
    auto a64 = [](uint64_t x) { return (x + 63) & ~63ull; };
    const uint64_t o_vars = a64(reads_dev), o_jobs = a64(o_vars + n_vars * sizeof(W2Variant)), o_len = a64(o_jobs + n * sizeof(W2Job)),
                   o_src = a64(o_len + n * 4), o_fmt = a64(o_src + n * 8), o_packed = a64(o_fmt + n);
    if ((rc = cx.stage.reserve(o_packed + (in_place ? 0 : packed) + 64)) != HP_OK) return rc;
    if ((rc = d_seq.alloc(seq_bytes)) || (rc = d_packed.alloc(packed + 64)) || (rc = d_src_off.alloc(n * 8)) || (rc = d_fmt.alloc(n + 16)) ||
        (rc = d_vars.alloc(std::max<size_t>(1, (size_t)n_vars) * sizeof(W2Variant))) || (rc = d_jobs.alloc(n * sizeof(W2Job))) ||
        (rc = d_nodes.alloc((size_t)node_tot * sizeof(W2Node))) || (rc = d_edges.alloc((size_t)edge_tot * 2)) || (rc = d_tags.alloc((size_t)tag_tot * 4)) ||
        (rc = d_info.alloc(n * sizeof(W2Info))) || (rc = d_order.alloc(n * 12)) || (rc = d_len_order.alloc(n * 4)) || (rc = d_cls.alloc(n + 16)) || (rc = d_job_cls.alloc(n + 16)) || (rc = d_handed.alloc(n + 16)) || (rc = d_seen.alloc(n * 4 + 16)) || (rc = d_blockcnt.alloc(((n + 255) / 256 + 1) * 16)) || (rc = d_sets.alloc(n * W2_SET_STRIDE * 4)) ||
        (rc = d_score.alloc(n * 8)) || (rc = d_work.alloc(n * 8 + 16)) || (rc = d_status.alloc(n * 4)) || (rc = d_alleles.alloc(std::max<uint64_t>(allele_tot, 16))))
        return rc;
    uint8_t* sb = cx.stage.p;
    // ---- the reads in place: run by run, no host thread touches them - and they cross while the tables below are filled ----------
    if (in_place) {
        for (const Run& r : runs)
            HP_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(d_packed.p) + r.dev, reinterpret_cast<const void*>(r.lo), r.hi - r.lo, hipMemcpyHostToDevice, st));
        g_in_place_bytes.fetch_add(in_place_bytes);
    }
    const unsigned nt = [&] {
        const char* tenv = std::getenv("HP_WFA_HOST_THREADS");
        const unsigned want = tenv ? (unsigned)std::max(1, std::atoi(tenv)) : host_threads(16u);
        return (unsigned)std::min<uint64_t>(want, std::max<uint64_t>(1, (packed + ref_bytes) >> 20));
    }();
    // ---- 3a. blocks: reference hulls, variants, allele pool; jobs: the device job table -----------------------------------------
    {
        std::atomic<size_t> next_b{0};
        W2Variant* sv = reinterpret_cast<W2Variant*>(sb + o_vars);
        W2Job* sj = reinterpret_cast<W2Job*>(sb + o_jobs);
        w2_parallel(nt, [&](unsigned t, unsigned T) {
            for (;;) {
                const size_t b = next_b.fetch_add(1);
                if (b >= n_in) break;
                const BlockLay& L = bl[b];
                if (L.lo > L.hi) continue;
                const hp_block_input& B = in[b];
                std::memcpy(sb + L.ref_dev, B.reference + ((uint64_t)L.lo - B.ref_base), (size_t)(L.hi - L.lo));
                uint64_t po = L.pool_off;
                for (int pass = 0; pass < 2; ++pass) {
                    const hp_wfa_variant* hv = pass ? B.homs : B.hets;
                    const uint32_t cnt = pass ? B.n_homs : B.n_hets;
                    W2Variant* w = sv + L.var_base + (pass ? B.n_hets : 0u);
                    for (uint32_t k = 0; k < cnt; ++k) {
                        w[k].position = hv[k].position; w[k].ref_len = hv[k].ref_len; w[k].flags = hv[k].flags;
                        w[k].a0_off = w[k].a0_len = 0;
                        if (hv[k].flags & 2u) {
                            w[k].a0_off = (uint32_t)po; w[k].a0_len = hv[k].allele0_len;
                            if (hv[k].allele0_len) std::memcpy(sb + alt_off + po, hv[k].allele0, hv[k].allele0_len);
                            po += hv[k].allele0_len;
                        }
                        w[k].a1_off = (uint32_t)po; w[k].a1_len = hv[k].allele1_len;
                        if (hv[k].allele1_len) std::memcpy(sb + alt_off + po, hv[k].allele1, hv[k].allele1_len);
                        po += hv[k].allele1_len;
                    }
                }
            }
            for (size_t i = n * t / T; i < n * (t + 1) / T; ++i) {
                const W2JobIn& ji = jin[i];
                const hp_block_input& B = in[ji.block];
                const hp_block_record& rec = B.records[ji.rec];
                const BlockLay& L = bl[ji.block];
                W2Job& d = dj[i];
                d.ref_start = rec.min_position;
                d.ref_len = (uint32_t)(rec.max_position + 1 - rec.min_position);
                d.ref_off = L.ref_dev + (uint64_t)(rec.min_position - L.lo);
                d.het_first = L.var_base + ji.het_first; d.n_hets = ji.n_hets;
                d.hom_first = L.var_base + B.n_hets + ji.hom_first; d.n_homs = ji.n_homs;
                sj[i] = d;
            }
            if (t == 0) {
                std::memcpy(sb + o_len, len_order.data(), n * 4);
                std::memcpy(sb + o_src, src_off.data(), n * 8);
                std::memcpy(sb + o_fmt, fmt.data(), n);
            }
        });
    }
    if (reads_dev) HP_HIP_CHECK(hipMemcpyAsync(d_seq.p, sb, reads_dev, hipMemcpyHostToDevice, st));
    if (n_vars) HP_HIP_CHECK(hipMemcpyAsync(d_vars.p, sb + o_vars, n_vars * sizeof(W2Variant), hipMemcpyHostToDevice, st));
    HP_HIP_CHECK(hipMemcpyAsync(d_jobs.p, sb + o_jobs, n * sizeof(W2Job), hipMemcpyHostToDevice, st));
    HP_HIP_CHECK(hipMemcpyAsync(d_len_order.p, sb + o_len, n * 4, hipMemcpyHostToDevice, st));
    HP_HIP_CHECK(hipMemcpyAsync(d_src_off.p, sb + o_src, n * 8, hipMemcpyHostToDevice, st));
    HP_HIP_CHECK(hipMemcpyAsync(d_fmt.p, sb + o_fmt, n, hipMemcpyHostToDevice, st));
    h2d_bytes = reads_dev + n_vars * sizeof(W2Variant) + n * (sizeof(W2Job) + 13) + in_place_bytes;
    // ---- 3b. the reads (unless they are crossing in place since before 3a), staged piece by piece: piece k crosses PCIe while the host
    // threads fill piece k + 1 ----
    if (!in_place) {
        const char* penv = std::getenv("HP_STAGE_PIECE_MB");
        const uint64_t piece = (uint64_t)std::max(1, penv ? std::atoi(penv) : 48) << 20;
        size_t j0 = 0;
        while (j0 < n) {
            size_t j1 = j0;
            while (j1 < n && src_off[j1] - src_off[j0] < piece) ++j1;
            std::atomic<size_t> next_j{j0};
            w2_parallel(nt, [&](unsigned, unsigned) {
                for (;;) {
                    const size_t a = next_j.fetch_add(64);
                    if (a >= j1) break;
                    for (size_t i = a; i < std::min(j1, a + 64); ++i) {
                        const hp_block_input& B = in[jin[i].block];
                        const hp_block_record& rec = B.records[jin[i].rec];
                        if (!rec.read_len) continue;
                        if (B.seq_format == HP_SEQ_BAM4) {
                            const uint32_t odd = rec.read_offset & 1u;
                            std::memcpy(sb + o_packed + src_off[i], rec.read_align + (rec.read_offset >> 1), ((size_t)odd + rec.read_len + 1) / 2);
                        } else std::memcpy(sb + o_packed + src_off[i], rec.read_align + rec.read_offset, rec.read_len);
                    }
                }
            });
            const uint64_t lo = src_off[j0], hi = src_off[j1];
            HP_HIP_CHECK(hipMemcpyAsync(reinterpret_cast<uint8_t*>(d_packed.p) + lo, sb + o_packed + lo, hi - lo, hipMemcpyHostToDevice, st));
            h2d_bytes += hi - lo;
            j0 = j1;
        }
    }
    // (the base expansion is the first kernel of run(): launched here it would queue behind the persistent alignment kernels of
    // the set before this one and hold the staging thread up; there it runs when they are gone, at the head of its own set)
    need_unpack = true;
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("upload failed"); return HP_ERR_HIP; }
    const double t1 = w2_now_ms();
    prep_ms[1] = t1 - t_up0; prep_ms[2] = prep_ms[0] + prep_ms[1]; prep_ms[3] = (double)h2d_bytes;
    last_prepare_ms = prep_ms[2];
    return HP_OK;
}

int W2Session::run(uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* const* alleles, int defer) {
    if (n == 0) return HP_OK;
    if (pend.on || async_inflight || early.on) { const int rcp = finish(); if (rcp != HP_OK) return rcp; }
    if (!out) { set_error("null argument"); return HP_ERR_ARG; }
    if (max_ed > 60000) {   // outside the kernels' diagonal range: every job of the batch, softly (HP_WFA_UNSUPPORTED)
        for (size_t i = 0; i < n; ++i) { out[i] = hp_wfa_result{HP_WFA_UNSUPPORTED, 0, 0}; if (alleles && alleles[i] && dj[i].n_hets) std::memset(alleles[i], HP_ALLELE_NOOVERLAP, dj[i].n_hets); }
        return HP_OK;
    }
    const bool verbose = std::getenv("HP_DEBUG") != nullptr;
    const double t0 = w2_now_ms();
    g_last_kernel_ms = 0.0;
    work_updates = work_node_bytes = work_read_bytes = work_jobs = 0;
    late_kernel_ms = 0.0;
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed", device_id); return HP_ERR_HIP; }
    const int n_cu = partition_cu_count(device_id);
    W2Context& cx = w2_context(device_id);   // (the calling thread's: need not be the thread that prepared the session)
    W2Context::Streams* cs_ = nullptr;
    { const int rc0 = cx.streams(g_cu_partition, &cs_, true); if (rc0 != HP_OK) return rc0; }
    hipStream_t st = cs_->stream;
    int rc = HP_OK;
    // host tables that stream operations read or write; the guard below (destroyed first) drains the stream on every
    // exit path, so none of them goes out of scope with a copy in flight
    // results come back into pinned staging (a pageable destination costs a bounce through the runtime's own buffers)
    const size_t dn_score = (n * 4 + 15) / 16 * 16, dn_work = dn_score + n * 8, dn_info = dn_work + n * 8, dn_cnt = dn_info + n * sizeof(W2Info), dn_esc = dn_cnt + 16, dn_al = dn_esc + 16;
    if ((rc = down.reserve(dn_al + (size_t)allele_tot + 16)) != HP_OK) return rc;
    W2Info* info_pin = reinterpret_cast<W2Info*>(down.p + dn_info);
    const uint32_t* cls_n = reinterpret_cast<const uint32_t*>(down.p + dn_cnt);
    const uint32_t turn = cx.run_no++ % 3u;   // whose turn among the largest class's streams / scratch regions
    hipStream_t cls_stream[3] = {cs_->cstream[0], cs_->cstream[1], turn == 0 ? cs_->cstream[2] : cs_->c2x[turn - 1]};
    struct StreamDrain { hipStream_t s; hipStream_t* c; bool skip2; bool off; ~StreamDrain() { if (off) return; for (int k = 0; k < 3; ++k) if (c[k] && !(skip2 && k == 2)) (void)hipStreamSynchronize(c[k]); (void)hipStreamSynchronize(s); } } drain{st, cls_stream, false, false};
    const double t_stage = t0;

    // ---- 3. graphs on the device ----------------------------------------------------------------------------------------
    for (auto& e : ev) if (!e) HP_HIP_CHECK(hipEventCreate(&e));
    hipEvent_t e0 = ev[0], e1 = ev[1], e2 = ev[2], e3 = ev[3];
    {
        W2BuildArgs A{};
        A.jobs = d_jobs.as<W2Job>(); A.n_jobs = (uint32_t)n; A.vars = d_vars.as<W2Variant>();
        A.nodes = d_nodes.as<W2Node>(); A.edges = d_edges.as<uint16_t>(); A.tags = d_tags.as<uint32_t>();
        A.info = d_info.as<W2Info>();
        if (need_unpack) {
            W2UnpackArgs U{};
            U.jobs = d_jobs.as<W2Job>(); U.src_off = d_src_off.as<uint64_t>(); U.fmt_nib = d_fmt.as<uint8_t>(); U.n_jobs = (uint32_t)n;
            U.packed = d_packed.as<uint8_t>(); U.seq = d_seq.as<uint8_t>(); U.tail_off = seq_bytes - 256;
            hipLaunchKernelGGL(hp_wfa2_unpack_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, U);
            HP_HIP_CHECK(hipGetLastError());
            need_unpack = false;
        }
        HP_HIP_CHECK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(hp_wfa2_build_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, A);
        HP_HIP_CHECK(hipGetLastError());
        HP_HIP_CHECK(hipEventRecord(e1, st));
    }
    // ---- 4. classes by graph size, longest read first: on the device, nothing comes back to the host in between --------------
    // capped-diagonal hash sets: one per resident group, kept (and never cleared) across calls
    const uint32_t max_groups = (uint32_t)n_cu * 96u;   // per class: up to 12 resident workgroups of 8 groups per CU
    const uint32_t large_groups = (uint32_t)n_cu * 40u;    // the largest class: at most 10 workgroups of 4 groups per CU
    const size_t set_dwords = (size_t)max_groups * (w2_group_dwords<2>() + w2_group_dwords<4>()) + (size_t)3 * large_groups * w2_group_dwords<8>();
    const size_t hash_groups = (size_t)2 * max_groups + (size_t)3 * large_groups;
    if (cx.htab_groups < max_groups) {
        if ((rc = cx.htab.alloc((hash_groups << W2_HCAP_LOG2) * 8)) != HP_OK) return rc;
        if ((rc = cx.gsets.alloc(set_dwords * 4)) != HP_OK) return rc;
        HP_HIP_CHECK(hipMemsetAsync(cx.htab.p, 0, (hash_groups << W2_HCAP_LOG2) * 8, st));
        HP_HIP_CHECK(hipMemsetAsync(cx.gsets.p, 0, set_dwords * 4, st));   // (the capped records carry tags too)
        cx.htab_groups = max_groups; cx.large_groups = large_groups; cx.tag_next = 0;
        cx.region_set_off[0] = 0; cx.region_set_off[1] = (size_t)max_groups * w2_group_dwords<2>();
        cx.region_set_off[2] = cx.region_set_off[1] + (size_t)max_groups * w2_group_dwords<4>();
        cx.region_hash_off[0] = 0; cx.region_hash_off[1] = max_groups; cx.region_hash_off[2] = (size_t)2 * max_groups;
        for (int r = 3; r < 5; ++r) { cx.region_set_off[r] = cx.region_set_off[r - 1] + (size_t)large_groups * w2_group_dwords<8>(); cx.region_hash_off[r] = cx.region_hash_off[r - 1] + large_groups; }
    }
    // The records the layout marked (layout_blocks: by their CIGARs they are heading for the neighbourhood of max_edit_distance) CAN be
    // routed past the compact kernels, their way out starting beside this launch set on the device's early worker instead of behind
    // it. HP_WFA2_ROUTE = 0 (default) never, 1 only while no other launch set of this device is between its launch and its first
    // collection, 2 always. OFF by default because it loses (round 5, MI355X, three runs a side, profiles/DIARY.md): a resident launch
    // set holds every wavefront slot its three wavefronts per SIMD and the largest class's waiting consumers can take, and the early
    // pass's kernels get in only where a workgroup retires - the reference-window test (one wavefront per job, in place) takes 13-17 ms
    // there against 3-4 in the gap a draining launch set opens (the late pass's moment), the dense band 12-19 with a long wait in
    // front: a set's routed records were out after 25-200 ms where the late pass delivers them 40-45 ms after the launch. Default bench
    // 1.73-1.92 M hets/s against 2.33-2.39 M (a set 190-240 ms from submit to done against 160-165), `idle only` the same (launch
    // sets follow each other so closely that the device counts as idle every other set), one call over a whole set 505 k against
    // 704 k hets/s, 64 blocking callers 62 k against 97 k. Kept as a switch with its parity test: what would make it pay is a
    // dense-band kernel that fits beside three class wavefronts per SIMD (<= 40 registers), not a different moment.
    const int route_mode = [] { const char* e = std::getenv("HP_WFA2_ROUTE"); return e ? std::atoi(e) : 0; }();
    const bool route = bl_in && !lay.suspects.empty() && route_mode > 0 && (route_mode >= 2 || sets_aligning(device_id).load() == 0);
    aligning_done();
    aligning.store(true); sets_aligning(device_id).fetch_add(1);
    if ((rc = d_qhead.alloc(512)) != HP_OK) return rc;
    if ((uint64_t)cx.tag_next + n + 2 >= 0xFFFFFFF0ull || (cx.last_gen != 0 && cx.last_gen != w2_gen())) {   // (tags wrap; or the other generation's words could pass for tags)
        (void)hipStreamSynchronize(cs_->cstream[2]); (void)hipStreamSynchronize(cs_->c2x[0]); (void)hipStreamSynchronize(cs_->c2x[1]);   // (an earlier run's tail may still use its region)
        HP_HIP_CHECK(hipMemsetAsync(cx.htab.p, 0, (hash_groups << W2_HCAP_LOG2) * 8, st));
        HP_HIP_CHECK(hipMemsetAsync(cx.gsets.p, 0, set_dwords * 4, st));
        cx.tag_next = 0;
    }
    for (int k = 0; k < 3; ++k) {
        if (!cx.cjoin[k]) HP_HIP_CHECK(hipEventCreateWithFlags(&cx.cjoin[k], hipEventDisableTiming));
    }
    if (!cx.cfork) HP_HIP_CHECK(hipEventCreateWithFlags(&cx.cfork, hipEventDisableTiming));
    HP_HIP_CHECK(hipMemsetAsync(d_qhead.p, 0, 512, st));   // work-queue heads at dword 16 k, class counts at dwords 64..67
    cx.last_gen = w2_gen();
    const uint32_t tag_base = cx.tag_next;
    cx.tag_next += (uint32_t)n + 1u;
    uint32_t* d_counts = d_qhead.as<uint32_t>() + 64;
    uint32_t* d_esc = d_qhead.as<uint32_t>() + 96;   // a cache line of its own
    {
        HP_HIP_CHECK(hipMemsetAsync(d_sets.p, 0, n * W2_SET_STRIDE * 4, st));
        HP_HIP_CHECK(hipMemsetAsync(d_work.p, 0, n * 8, st));
        W2ClassArgs CA{};
        CA.jobs = d_jobs.as<W2Job>(); CA.info = d_info.as<W2Info>(); CA.len_order = d_len_order.as<uint32_t>(); CA.n_jobs = (uint32_t)n;
        CA.order = d_order.as<uint32_t>(); CA.counts = d_counts; CA.status = d_status.as<int32_t>();
        CA.cls = d_cls.as<uint8_t>(); CA.blockcnt = d_blockcnt.as<uint32_t>(); CA.esc = d_esc; CA.job_cls = d_job_cls.as<uint8_t>();
        HP_HIP_CHECK(hipMemsetAsync(d_handed.p, 0, n, st));
        { const char* e = std::getenv("HP_WFA2_USE_W2"); CA.use_w2 = (e && e[0] == '0') ? 0u : 1u; }
        CA.fmt = route ? d_fmt.as<uint8_t>() : nullptr;
        hipLaunchKernelGGL(hp_wfa2_classify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, CA);
        hipLaunchKernelGGL(hp_wfa2_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, CA);
        HP_HIP_CHECK(hipGetLastError());
    }
    // the class sizes come back (16 bytes, one short wait): a grid sized for the class fills every workgroup's groups, and
    // an empty class is not launched. Measured: launching each class with the whole batch's grid instead (no wait) cost
    // 3-4 ms of 22 - the W=8 class then spreads its few long jobs one per workgroup and holds LDS for idle groups.
    if ((rc = dev_copy(down.p + dn_cnt, d_counts, 16, st)) != HP_OK) return rc;   // (a kernel's store to pinned memory: a copy-engine transfer would queue behind the next set's upload)
    const bool has_early = route;
    if (has_early && (rc = dev_copy(info_pin, d_info.p, n * sizeof(W2Info), st)) != HP_OK) return rc;   // (the routed records' node counts: their results carry them)
    if (hipStreamSynchronize(st) != hipSuccess) { set_error("WFA graph-build kernel failed"); return HP_ERR_HIP; }
    const double t_built = w2_now_ms();
    W2Batch B{};
    B.jobs = d_jobs.as<W2Job>(); B.info = d_info.as<W2Info>(); B.tag_base = tag_base;
    B.nodes = d_nodes.as<W2Node>(); B.edges = d_edges.as<uint16_t>(); B.seq = d_seq.as<uint8_t>(); B.alt_off = alt_off;
    B.out_sets = d_sets.as<uint32_t>(); B.out_score = d_score.as<uint64_t>(); B.status = d_status.as<int32_t>(); B.out_work = d_work.as<uint32_t>();
    B.htab = cx.htab.as<uint64_t>(); B.hcap_log2 = W2_HCAP_LOG2; B.gsets = cx.gsets.as<uint32_t>(); B.set_stride = 0; B.prune_distance = prune_distance; B.max_ed = max_ed;
    const double t_cls = w2_now_ms();
    HP_HIP_CHECK(hipEventRecord(e2, st));
    HP_HIP_CHECK(hipEventRecord(cx.cfork, st));
    uint32_t groups_used[3] = {0, 0, 0};
    W2Batch pend_b2 = B;   // what the largest class was launched with (late() may launch it once more)
    { const size_t region = 2u + turn; pend_b2.htab = cx.htab.as<uint64_t>() + (cx.region_hash_off[region] << W2_HCAP_LOG2); pend_b2.gsets = cx.gsets.as<uint32_t>() + cx.region_set_off[region];
      pend_b2.set_stride = (uint32_t)w2_group_dwords<8>(); pend_b2.esc = d_esc; pend_b2.esc_order = d_order.as<uint32_t>() + (size_t)2 * n; pend_b2.handed = d_handed.as<uint8_t>(); }
    const uint32_t cls_cnt[3] = {cls_n[0], cls_n[1], cls_n[2]};
    const char* genv = std::getenv("HP_WFA2_G");   // experiment: lanes per read for the middle class
    const int gsel = genv ? std::atoi(genv) : 8;
    // escalation (W2Batch::esc): the two smaller classes hand jobs that outgrow their tables to the largest one, whose
    // kernel stays until both are gone. It is launched first, and with room for such jobs even when it has none of its own.
    const char* eenv = std::getenv("HP_WFA2_ESCALATE");
    const bool escalate = !(eenv && eenv[0] == '0') && (cls_cnt[0] || cls_cnt[1]);
    uint32_t grid_wg[3] = {0, 0, 0};
    // Groups each class may occupy. On its own a launch set fills the chip (each of the two smaller classes is sized for all of
    // it: the second one's workgroups move in as the first one's leave). As a stage of a block stream it must NOT: the kernels are
    // persistent - no workgroup leaves before its class has run dry - so the kernels of the other stages (the copy engine's
    // blits of the next set, the A* / Levenshtein / post-processing kernels of the previous one) would wait tens of
    // milliseconds for a slot. g_wfa2_reserve_pct leaves that share of the slots empty and splits the rest by job count.
    uint32_t capg[3] = {cx.htab_groups, cx.htab_groups, cx.htab_groups};
    if (g_wfa2_reserve_pct > 0 && (cls_cnt[0] || cls_cnt[1])) {
        const uint64_t total = (uint64_t)n_cu * 96u * (uint64_t)(100 - std::min(90, g_wfa2_reserve_pct)) / 100u;
        const uint64_t c0 = cls_cnt[0], c1 = cls_cnt[1];
        // (a job of the middle class - a larger graph - takes longer: 1.3 x in the second generation (spans 23 vs 29.5 ms at equal shares),
        // about 1.1 x in the third (21.9 vs 18.2 ms at 10 : 13). HP_WFA2_SPLIT = its weight in hundredths.)
        static const uint64_t w1 = [] { const char* e = std::getenv("HP_WFA2_SPLIT"); return (uint64_t)(e ? std::max(50, std::min(400, std::atoi(e))) : (w2_gen() == 3 ? 110 : 130)); }();
        uint64_t g0 = total * (100 * c0) / (100 * c0 + w1 * c1);
        if (c0) g0 = std::max<uint64_t>(g0, 64);
        if (c1) g0 = std::min<uint64_t>(g0, total - 64);
        capg[0] = (uint32_t)std::max<uint64_t>(8, g0 & ~7ull);
        capg[1] = (uint32_t)std::max<uint64_t>(8, (total - g0) & ~7ull);
    }
    // HP_WFA2_GROUP_JOBS=k: the two smaller classes' groups leave after k jobs each (more if the list would not fit the scratch
    // region's groups otherwise) instead of staying until the list is empty
    const char* gjenv = std::getenv("HP_WFA2_GROUP_JOBS");
    const uint32_t gj_want = gjenv ? (uint32_t)std::max(0, std::atoi(gjenv)) : 0u;
    uint32_t gjobs[3] = {0, 0, 0};
    if (gj_want) for (int k = 0; k < 2; ++k) { gjobs[k] = std::max<uint32_t>(gj_want, (cls_cnt[k] + cx.htab_groups - 1) / cx.htab_groups); capg[k] = cx.htab_groups; }
    if (cls_cnt[0]) grid_wg[0] = w2_grid<8, 2>(cls_cnt[0], n_cu, capg[0], gjobs[0]);
    if (cls_cnt[1]) grid_wg[1] = gsel == 16 ? w2_grid<16, 4>(cls_cnt[1], n_cu, capg[1], gjobs[1]) : gsel == 32 ? w2_grid<32, 4>(cls_cnt[1], n_cu, capg[1], gjobs[1]) : w2_grid<8, 4>(cls_cnt[1], n_cu, capg[1], gjobs[1]);
    // (room for the jobs handed over. Round 3 sized it for 2 % of the two smaller classes (1 200 hand-overs a set then); the third
    // generation hands over 160-200 a set, and 2 100 groups of consumers - 575 workgroups that mostly sleep on their tickets - sat in
    // exactly the wavefront slots the stream's reserve (HP_STREAM_RESERVE_PCT) is meant to leave to the neighbouring stages' kernels:
    // the launch set asked for 3 427 of the device's 3 072 slots. Round 5, three runs a side: 1 / 64 of the smaller classes' jobs
    // 2.36-2.39 M hets/s (class kernels' span 18.9-19.5 ms), 1 / 256: 2.43-2.45 M (17.9), 1 / 1 024: 2.44-2.50 M (18.0-18.2) - with
    // the reserve really free its size no longer matters between 4 and 12 %. 1 / 512 (>= 192 groups): room for 530 hand-overs a set;
    // what does not fit stays a leftover for the late pass, as ever. HP_WFA2_ESC_DIV to experiment.)
    const char* denv = std::getenv("HP_WFA2_ESC_DIV");
    const uint32_t esc_div = denv ? (uint32_t)std::max(1, std::atoi(denv)) : 512u;
    const uint32_t items2 = escalate ? cls_cnt[2] + std::max<uint32_t>(4u * 48u, (cls_cnt[0] + cls_cnt[1]) / esc_div) : cls_cnt[2];
    // two phases: the results of everything the two smaller classes finished themselves are collected as soon as THEIR
    // kernels are done; the largest class (its own jobs + what was handed over, the tail of the launch set) is collected
    // by late(), which the block layer overlaps with the row assembly of the blocks that do not wait for it
    const char* tpenv = std::getenv("HP_WFA2_TWO_PHASE");
    const bool two_phase = defer >= 2 && escalate && !(tpenv && tpenv[0] == '0');
    int launch_order[3] = {1, 0, 2};   // measured on the default bench (ms of the span): 102 22.5, 120 22.6, 012 22.8, 201 23.2
    if (const char* e = std::getenv("HP_WFA2_ORDER")) { if (std::strlen(e) == 3) for (int i = 0; i < 3; ++i) launch_order[i] = std::min(2, std::max(0, e[i] - '0')); }
    for (int li = 0; li < 3; ++li) {
        const int k = launch_order[li];
        if (cls_cnt[k] == 0 && !(k == 2 && escalate)) continue;
        hipStream_t cs = cls_stream[k];
        const size_t region = k < 2 ? (size_t)k : 2u + turn;
        HP_HIP_CHECK(hipStreamWaitEvent(cs, cx.cfork, 0));
        B.order = d_order.as<uint32_t>() + (size_t)k * n;
        B.n_items = cls_cnt[k];
        B.n_items_dev = d_counts + k;
        B.next = d_qhead.as<uint32_t>() + 16 * k;
        B.htab = cx.htab.as<uint64_t>() + (cx.region_hash_off[region] << W2_HCAP_LOG2);
        B.gsets = cx.gsets.as<uint32_t>() + cx.region_set_off[region];
        B.set_stride = k == 0 ? (uint32_t)w2_group_dwords<2>() : k == 1 ? (uint32_t)w2_group_dwords<4>() : (uint32_t)w2_group_dwords<8>();
        B.esc = d_esc; B.esc_order = d_order.as<uint32_t>() + (size_t)2 * n; B.handed = d_handed.as<uint8_t>();
        B.esc_role = !escalate ? 0u : (k == 2 ? 2u : 1u);
        B.esc_producers = grid_wg[0] + grid_wg[1];
        B.esc_limit = cls_cnt[2] + 2u * (items2 - cls_cnt[2]);   // two jobs for each group reserved for hand-overs
        B.group_jobs = gjobs[k];
        static const uint32_t hopeless_pct = [] { const char* e = std::getenv("HP_WFA2_HOPELESS"); return (uint32_t)(e ? std::max(10, std::min(100000, std::atoi(e))) : 50); }();
        B.hopeless_pct = hopeless_pct;
        if (k == 0) rc = w2_launch<8, 2>(B, B.n_items, n_cu, capg[0], cs, &groups_used[k]);
        else if (k == 1) rc = gsel == 16 ? w2_launch<16, 4>(B, B.n_items, n_cu, capg[1], cs, &groups_used[k])
                            : gsel == 32 ? w2_launch<32, 4>(B, B.n_items, n_cu, capg[1], cs, &groups_used[k])
                                         : w2_launch<8, 4>(B, B.n_items, n_cu, capg[1], cs, &groups_used[k]);
        else { rc = w2_launch<16, 8>(B, items2, n_cu, cx.large_groups, cs, &groups_used[k]); pend_b2 = B; }
        if (rc != HP_OK) return rc;
        if (two_phase && k == 2) { HP_HIP_CHECK(hipEventRecord(e3, cs)); continue; }   // collected later (late())
        HP_HIP_CHECK(hipEventRecord(cx.cjoin[k], cs));
        HP_HIP_CHECK(hipStreamWaitEvent(st, cx.cjoin[k], 0));
    }
    if (!two_phase) HP_HIP_CHECK(hipEventRecord(e3, st));
    if (has_early) {   // the routed records' way out starts now (the expanded bases, the job table and the builder's verdicts are on the device)
        early.ids = lay.suspects;
        early.mask.assign(n, 0);
        early.big = early.ids; early.big_ed.assign(early.ids.size(), (uint32_t)std::min<uint64_t>(max_ed, 0xFFFFFFu)); early.big_nodes.resize(early.ids.size());
        for (size_t k = 0; k < early.ids.size(); ++k) {
            const uint32_t i = early.ids[k];
            early.mask[i] = 1;
            early.big_nodes[k] = info_pin[i].status == W2B_OK ? info_pin[i].n_nodes : 0u;
        }
        early.dst = out; early.alleles = alleles; early.prune = prune_distance; early.max_ed = max_ed;
        early.rc = HP_OK; early.err.clear(); early.n_settled = 0;
        early.t_post = w2_now_ms();
        g_routed_records.fetch_add(early.ids.size());
        { std::lock_guard<std::mutex> lk(em); early.on = true; early.done = false; }
        W2Session* self = this;
        early_worker(device_id).post([self]() { self->early_pass(); });
    }
    {
        W2MapArgs M{};
        M.jobs = d_jobs.as<W2Job>(); M.info = d_info.as<W2Info>(); M.n_jobs = (uint32_t)n; M.tags = d_tags.as<uint32_t>();
        M.out_sets = d_sets.as<uint32_t>(); M.status = d_status.as<int32_t>(); M.alleles = d_alleles.as<uint8_t>();
        M.nodes = d_nodes.as<W2Node>(); M.out_work = d_work.as<uint32_t>();
        M.job_cls = d_job_cls.as<uint8_t>(); M.handed = d_handed.as<uint8_t>(); M.seen = d_seen.as<int32_t>(); M.first = two_phase ? 1u : 0u;
        hipLaunchKernelGGL(hp_wfa2_map_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, M);
        HP_HIP_CHECK(hipGetLastError());
    }
    // ---- 5. results --------------------------------------------------------------------------------------------------------
    if ((rc = dev_copy(info_pin, d_info.p, n * sizeof(W2Info), st)) || (rc = dev_copy(down.p + dn_esc, d_esc, 16, st)) || (rc = dev_copy(down.p, d_seen.p, n * 4, st)) ||
        (rc = dev_copy(down.p + dn_score, d_score.p, n * 8, st)) || (rc = dev_copy(down.p + dn_al, d_alleles.p, (size_t)allele_tot, st)) ||
        (rc = dev_copy(down.p + dn_work, d_work.p, n * 8, st)))
        return rc;
    rs = RunState{};
    rs.dn_score = dn_score; rs.dn_work = dn_work; rs.dn_info = dn_info; rs.dn_cnt = dn_cnt; rs.dn_al = dn_al;
    rs.out = out; rs.alleles = alleles; rs.two_phase = two_phase; rs.verbose = verbose; rs.prune = prune_distance; rs.max_ed = max_ed;
    rs.stream2 = cls_stream[2]; rs.b2 = pend_b2; rs.large_groups = cx.large_groups; rs.n_cu = n_cu; rs.defer = defer;
    for (int k = 0; k < 3; ++k) rs.groups_used[k] = groups_used[k];
    rs.t0 = t0; rs.t_built = t_built; rs.t_cls = t_cls;
    // Inside a block stream (defer = 2) the launching thread does not wait for the set: the class kernels take ~20 ms, and whatever this
    // thread did after them - the wait for the copies above, sorting out what is held and what is left over, handing the results to
    // the block layer's arrays: 2.5-3 ms a set - stood between this set's kernels and the next set's. The session's helper thread
    // waits for the copies, sorts the results out (collect_host) and goes on with late(); the block layer's row stage scatters
    // (wait_collected) before it reads anything. The next run() of the calling thread queues behind this one on the same streams.
    static const bool async_ok = [] { const char* e = std::getenv("HP_WFA2_ASYNC"); return !(e && e[0] == '0'); }();
    if (two_phase && defer >= 2 && async_ok) {
        if (!ev_c) HP_HIP_CHECK(hipEventCreateWithFlags(&ev_c, hipEventDisableTiming));
        HP_HIP_CHECK(hipEventRecord(ev_c, st));
        if (!helper) { helper.reset(new HelperThread()); helper->start(); }
        { std::lock_guard<std::mutex> lk(cm); collected = false; scattered = false; collect_rc = HP_OK; collect_err.clear(); }
        async_inflight = true;
        drain.off = true;   // (nothing of this frame is read or written by what is still queued: the results land in the session's `down`)
        const int part = g_cu_partition;
        W2Session* self = this;
        helper->post([self, part]() {
            // (inside a block stream the graph-WFA stage owns partition 2, which the NEXT set's persistent kernels fill: the
            // dense-band pass of this set's leftovers launches on the whole device and runs where there is room)
            g_cu_partition = part == 2 ? 0 : part;
            int rcc = HP_OK;
            if (hp_set_device(self->device_id) != hipSuccess || hipEventSynchronize(self->ev_c) != hipSuccess) { set_error("WFA kernel failed"); rcc = HP_ERR_HIP; }
            self->aligning_done();
            if (rcc == HP_OK) rcc = self->collect_host();
            if (rcc == HP_OK && self->pend.on) self->pend.posted = true;
            {
                std::lock_guard<std::mutex> lk(self->cm);
                self->collect_rc = rcc;
                if (rcc != HP_OK) self->collect_err = hp_last_error();
                self->collected = true;
            }
            self->ccv.notify_all();
            self->pend.rc = rcc;
            if (rcc != HP_OK) { self->pend.err = hp_last_error(); return; }
            if (!self->pend.on) return;
            self->pend.rc = self->late();
            if (self->pend.rc != HP_OK) self->pend.err = hp_last_error();
        });
        return HP_OK;
    }
    { const bool okk = hipStreamSynchronize(st) == hipSuccess; aligning_done(); if (!okk) { set_error("WFA kernel failed"); return HP_ERR_HIP; } }
#if W2_PROF
    (void)hipDeviceSynchronize();   // flushes the instrumented kernel's printf buffer
#endif
    if ((rc = collect_host()) != HP_OK) return rc;
    // ---- 6. what is not here yet (late()): on the session's helper thread when the caller defers, else right below ----
    if (pend.on && defer) {
        if (!helper) { helper.reset(new HelperThread()); helper->start(); }
        const int part = g_cu_partition;
        W2Session* self = this;
        pend.posted = true;
        helper->post([self, part]() {
            g_cu_partition = part == 2 ? 0 : part;
            self->pend.rc = self->late();
            if (self->pend.rc != HP_OK) self->pend.err = hp_last_error();
        });
        drain.skip2 = true;
    }
    struct Joiner { W2Session* s; bool armed; ~Joiner() { if (armed && s->pend.on && s->pend.posted) { s->helper->wait(); s->pend.on = false; } } } joiner{this, true};
    if ((rc = scatter()) != HP_OK) return rc;
    joiner.armed = false;
    if (defer) return HP_OK;
    return finish();
}

// The first collection sorted out on the host, once its copies have landed in `down`: which jobs are still with the largest class
// (held), which no class could align (big: the dense-band pass). Fills `pend`. On the helper thread when run() did not wait.
int W2Session::collect_host() {
    const int32_t* status = reinterpret_cast<const int32_t*>(down.p);
    const uint64_t* score = reinterpret_cast<const uint64_t*>(down.p + rs.dn_score);
    const W2Info* info = reinterpret_cast<const W2Info*>(down.p + rs.dn_info);
    const bool two_phase = rs.two_phase;
    float ms_build = 0.f, ms_wfa = 0.f;
    (void)hipEventElapsedTime(&ms_build, ev[0], ev[1]);
    if (!two_phase) (void)hipEventElapsedTime(&ms_wfa, ev[2], ev[3]);
    g_last_kernel_ms = (double)ms_build + (double)ms_wfa;
    last_span_ms = (double)ms_wfa;
    rs.t_done = w2_now_ms();
    // held: still with the largest class's kernel (two phases). big: no class took them (builder limits, graph size, read
    // length), or the largest class handed them back, or (one phase) nobody claimed them - the dense-band pass aligns those.
    pend = Pending{};
    pend.two_phase = two_phase; pend.dst = rs.out; pend.alleles = rs.alleles; pend.prune = rs.prune; pend.max_ed = rs.max_ed;
    pend.stream2 = rs.stream2; pend.ms_build = ms_build;
    bool early_on;
    { std::lock_guard<std::mutex> lk(em); early_on = early.on; }
    const uint8_t* routed = early_on ? early.mask.data() : nullptr;
    pend.b2 = rs.b2; pend.large_groups = rs.large_groups; pend.n_cu = rs.n_cu;
    for (size_t i = 0; i < n; ++i) {
        if (info[i].status == W2B_INVARIANT) { set_error("graph construction assert (wfa_graph.rs:170,257,276,281) on job %zu", i); return HP_ERR_INVARIANT; }
        if (routed && routed[i]) continue;   // (with the early pass since the launch set started; finish() delivers it: pend.ids below)
        if (status[i] == W2_ST_PENDING) { if (two_phase) { pend.held.push_back((uint32_t)i); pend.held_nodes.push_back(info[i].n_nodes); } else { pend.big.push_back((uint32_t)i); pend.big_ed.push_back(0); pend.big_nodes.push_back(info[i].status == W2B_OK ? info[i].n_nodes : 0u); } }
        else if (status[i] == W2_ST_NEED_BIG) { pend.big.push_back((uint32_t)i); pend.big_ed.push_back((uint32_t)(score[i] >> 8)); pend.big_nodes.push_back(info[i].status == W2B_OK ? (info[i].n_nodes | ((uint32_t)(score[i] & 0xFFu) << 24)) : 0u); }
    }
    if (rs.verbose && !pend.big.empty()) {   // which class handed jobs back, and which of its limits (hp_wfa2_kernel's `why`)
        uint32_t hist[3][10] = {};
        for (uint32_t i : pend.big) {
            const uint32_t nn = info[i].n_nodes, ne = info[i].n_edges;
            if (info[i].status != W2B_OK || (score[i] & 0xFF) == 0 || (score[i] & 0xFF) > 9) continue;
            const int k = (nn <= (uint32_t)W2Cfg<2>::MAXN && ne <= (uint32_t)W2Cfg<2>::MAXE) ? 0 : (nn <= (uint32_t)W2Cfg<4>::MAXN && ne <= (uint32_t)W2Cfg<4>::MAXE) ? 1 : 2;
            hist[k][score[i] & 0xFF]++;
        }
        for (int k = 0; k < 3; ++k) {
            fprintf(stderr, "[hp] wfa2: class %d handed back:", k);
            for (int r = 1; r < 10; ++r) if (hist[k][r]) fprintf(stderr, " reason %d x %u", r, hist[k][r]);
            fprintf(stderr, "\n");
        }
    }
    pend.ids = pend.held;
    pend.ids.insert(pend.ids.end(), pend.big.begin(), pend.big.end());
    if (routed) pend.ids.insert(pend.ids.end(), early.ids.begin(), early.ids.end());
    pend.on = !pend.ids.empty() || two_phase;
    return HP_OK;
}

// The first collection's results into the caller's arrays (host threads of the calling thread's pool).
int W2Session::scatter() {
    const int32_t* status = reinterpret_cast<const int32_t*>(down.p);
    const uint64_t* score = reinterpret_cast<const uint64_t*>(down.p + rs.dn_score);
    const uint32_t* work = reinterpret_cast<const uint32_t*>(down.p + rs.dn_work);
    const W2Info* info = reinterpret_cast<const W2Info*>(down.p + rs.dn_info);
    const uint32_t* cls_n = reinterpret_cast<const uint32_t*>(down.p + rs.dn_cnt);
    const uint8_t* al = down.p + rs.dn_al;
    hp_wfa_result* out = rs.out;
    uint8_t* const* alleles = rs.alleles;
    std::atomic<int64_t> bad{-1};
    {
        const unsigned nt = w2_host_threads(n, 8192);
        std::vector<uint64_t> acc((size_t)nt * 4, 0);
        w2_parallel(nt, [&](unsigned tid, unsigned nth) {
            const size_t lo = n * tid / nth, hi = n * (tid + 1) / nth;
            uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (size_t i = lo; i < hi; ++i) {
                if (status[i] == W2_ST_NEED_BIG || status[i] == W2_ST_PENDING) continue;
                if (status[i] != W2_ST_OK && status[i] != W2_ST_MAX_ED) { bad.store((int64_t)i); return; }
                a0 += work[2 * i]; a1 += work[2 * i + 1]; a2 += dj[i].read_len; ++a3;
                out[i].status = status[i] == W2_ST_OK ? HP_OK : HP_WFA_MAX_ED;
                out[i].n_nodes = info[i].n_nodes;
                out[i].score = score[i];
                if (alleles && alleles[i] && dj[i].n_hets) std::memcpy(alleles[i], al + dj[i].allele_off, dj[i].n_hets);
            }
            acc[(size_t)tid * 4] = a0; acc[(size_t)tid * 4 + 1] = a1; acc[(size_t)tid * 4 + 2] = a2; acc[(size_t)tid * 4 + 3] = a3;
        });
        uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (unsigned k = 0; k < nt; ++k) { s0 += acc[(size_t)k * 4]; s1 += acc[(size_t)k * 4 + 1]; s2 += acc[(size_t)k * 4 + 2]; s3 += acc[(size_t)k * 4 + 3]; }
        std::lock_guard<std::mutex> lk(work_m);   // (late() adds the held jobs' share, possibly at the same time)
        work_updates += s0; work_node_bytes += s1; work_read_bytes += s2; work_jobs += s3;
    }
    if (bad.load() >= 0) { set_error("job %lld: device status %d", (long long)bad.load(), status[bad.load()]); return HP_ERR_INVARIANT; }
    if (rs.verbose) {
        fprintf(stderr, "[hp] wfa2: %zu jobs (classes %zu/%zu/%zu, groups %u/%u/%u, %zu for the dense-band path of which %zu by size/builder, %zu collected later with the largest class): upload+build %.2f ms, queueing the classes %.2f ms%s, first results on the host after %.2f ms, scattered after %.2f ms\n",
                n, (size_t)cls_n[0], (size_t)cls_n[1], (size_t)cls_n[2], rs.groups_used[0], rs.groups_used[1], rs.groups_used[2], pend.ids.size() - pend.held.size(), (size_t)cls_n[3], pend.held.size(), rs.t_built - rs.t0, rs.t_cls - rs.t_built,
                rs.two_phase ? " (span known after the second collection)" : "", rs.t_done - rs.t0, w2_now_ms() - rs.t0);
        fflush(stderr);
    }
    return HP_OK;
}

int W2Session::wait_collected() {
    if (!async_inflight) return HP_OK;   // (run() waited itself: everything of the first collection is with the caller already)
    {
        std::unique_lock<std::mutex> lk(cm);
        ccv.wait(lk, [this]() { return collected; });
        if (collect_rc != HP_OK) { set_error("%s", collect_err.c_str()); return collect_rc; }
        if (scattered) return HP_OK;
        scattered = true;
    }
    return scatter();
}

// The leftovers' way out (everything in `big`): the reference-window test, then the dense-band pass for what it leaves. Runs on the
// session's helper thread (late()) and, for the records the layout routed past the compact kernels, on the device's early worker.
int W2Session::leftovers_out(std::vector<uint32_t>& big, std::vector<uint32_t>& big_ed, std::vector<uint32_t>& big_nodes, SubWork& sw, hp_wfa_result* dst,
                             uint8_t* const* alleles, uint64_t prune, uint64_t max_ed, bool beside_launch_set, double* t_bound, size_t* n_settled, double* kernel_ms) {
    if (!big.empty()) {
        {   // ascending job order (with the hints)
            std::vector<std::array<uint32_t, 3>> z(big.size());
            for (size_t k = 0; k < z.size(); ++k) z[k] = {big[k], k < big_ed.size() ? big_ed[k] : 0u, k < big_nodes.size() ? big_nodes[k] : 0u};
            std::sort(z.begin(), z.end());
            big_ed.resize(z.size()); big_nodes.resize(z.size());
            for (size_t k = 0; k < z.size(); ++k) { big[k] = z[k][0]; big_ed[k] = z[k][1]; big_nodes[k] = z[k][2]; }
        }
        // ---- the cheap exact verdict first (hp_wfa2_bound_kernel): a read that was deep into its alignment when the compact
        // kernels let go of it, and whose distance to the reference window alone exceeds max_edit_distance + D, is a
        // MaxEditDistance - no dense-band pass for it ----
        {
            const char* benv = std::getenv("HP_WFA2_BOUND");
            // (every leftover is tested: a read that aligns within the threshold ends the test after about as many rounds as it has
            // edits, and the noisy ones often leave the compact kernels early, on a full capped set. HP_WFA2_BOUND=n: only reads that
            // had reached n edits; 1000000 turns the shortcut off)
            const uint32_t min_ed = benv ? (uint32_t)std::max(0, std::atoi(benv)) : 0u;
            std::vector<uint32_t> cand, thr, cand_pos;
            for (size_t k = 0; k < big.size(); ++k) {
                if (big_ed[k] < min_ed || Pending::nodes_of(big_nodes[k]) == 0) continue;
                const hp_wfa_job j = job_header(big[k]);
                uint64_t D = 0;
                for (uint32_t v = 0; v < j.n_hets; ++v) D += std::max<uint64_t>({j.hets[v].ref_len, (j.hets[v].flags & 2u) ? j.hets[v].allele0_len : 0u, j.hets[v].allele1_len});
                for (uint32_t v = 0; v < j.n_homs; ++v) D += std::max<uint64_t>({j.homs[v].ref_len, (j.homs[v].flags & 2u) ? j.homs[v].allele0_len : 0u, j.homs[v].allele1_len});
                const uint64_t T = max_ed + D;
                if (T > W2_BOUND_MAX_T) continue;
                cand.push_back(big[k]); thr.push_back((uint32_t)T); cand_pos.push_back((uint32_t)k);
            }
            if (!cand.empty()) {
                hipStream_t bs = thread_stream(device_id);
                if (!bs) { set_error("stream creation failed"); return HP_ERR_HIP; }
                DevBuf d_ids, d_thr, d_exc;
                int rcb;
                if ((rcb = d_ids.alloc(cand.size() * 4)) || (rcb = d_thr.alloc(cand.size() * 4)) || (rcb = d_exc.alloc(cand.size() + 16))) return rcb;
                std::vector<uint8_t> exc(cand.size(), 0);
                struct Drain { hipStream_t s; ~Drain() { (void)hipStreamSynchronize(s); } } drain{bs};
                struct IoDrain { hipStream_t s; ~IoDrain() { dev_io_abort(s); } } io{bs};
                if ((rcb = dev_put(d_ids.p, cand.data(), cand.size() * 4, bs)) != HP_OK || (rcb = dev_put(d_thr.p, thr.data(), thr.size() * 4, bs)) != HP_OK) return rcb;
                W2BoundArgs BA{};
                BA.jobs = d_jobs.as<W2Job>(); BA.ids = d_ids.as<uint32_t>(); BA.thresh = d_thr.as<uint32_t>(); BA.n = (uint32_t)cand.size();
                BA.seq = d_seq.as<uint8_t>(); BA.exceeds = d_exc.as<uint8_t>();
                const uint32_t maxT = *std::max_element(thr.begin(), thr.end());
                // LDS: the two wavefront arrays, then room for the longest tested read + its window (most of the CU's 160 KB: these
                // are a few hundred single-wavefront workgroups, latency is what counts)
                uint32_t need_seq = 0;
                for (uint32_t i : cand) need_seq = std::max<uint32_t>(need_seq, ((dj[i].read_len + 31u) & ~15u) + ((dj[i].ref_len + 31u) & ~15u));
                const size_t lds_wf = (size_t)(2 * (2 * maxT + 3) + 2) * 4 + 16;
                BA.max_t = maxT;
                // (beside a resident launch set - the early pass - a compute unit has one wavefront slot and 11-23 KB of LDS to spare: one
                // wavefront per job, the sequences compared where they lie (L2), only the two wavefront arrays in LDS - 10 KB at T = 600.
                // A workgroup of four wavefronts with 46 KB waited for a launch set to drain, i.e. for the moment the late pass starts at.)
                BA.lds_seq = beside_launch_set ? 0u : std::min<uint32_t>(need_seq, W2_BOUND_LDS_SEQ);
                const size_t lds_total = lds_wf + BA.lds_seq;
                static const int bound_threads = [] { const char* e = std::getenv("HP_BOUND_THREADS"); return (e && std::atoi(e) == 64) ? 64 : 256; }();
                {   // (once per device, and for the most any launch asks for: launches of several threads must not lower it under each other)
                    static std::mutex attr_m;
                    static std::map<int, hipError_t> attr_done;
                    std::lock_guard<std::mutex> lk(attr_m);
                    auto it = attr_done.find(device_id);
                    if (it == attr_done.end()) {
                        const int most = (int)((size_t)(2 * (2 * W2_BOUND_MAX_T + 3) + 2) * 4 + 16 + W2_BOUND_LDS_SEQ);
                        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_wfa2_bound_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
                        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hp_wfa2_bound_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
                        it = attr_done.emplace(device_id, e).first;
                    }
                    HP_HIP_CHECK(it->second);
                }
                if (bound_threads == 64 || beside_launch_set) hipLaunchKernelGGL(hp_wfa2_bound_kernel<64>, dim3((unsigned)cand.size()), dim3(64), lds_total, bs, BA);
                else hipLaunchKernelGGL(hp_wfa2_bound_kernel<256>, dim3((unsigned)cand.size()), dim3(256), lds_total, bs, BA);
                HP_HIP_CHECK(hipGetLastError());
                if ((rcb = dev_get(exc.data(), d_exc.p, cand.size(), bs)) != HP_OK) return rcb;
                if (dev_io_sync(bs) != HP_OK) { set_error("WFA bound kernel failed"); return HP_ERR_HIP; }
                std::vector<uint32_t> keep, keep_ed, keep_nodes;
                size_t c = 0, settled = 0;
                for (size_t k = 0; k < big.size(); ++k) {
                    const bool tested = c < cand_pos.size() && cand_pos[c] == k;
                    if (tested && exc[c]) {
                        const uint32_t i = big[k];
                        dst[i].status = HP_WFA_MAX_ED; dst[i].n_nodes = Pending::nodes_of(big_nodes[k]); dst[i].score = max_ed;
                        if (alleles && alleles[i] && dj[i].n_hets) std::memset(alleles[i], HP_ALLELE_NOOVERLAP, dj[i].n_hets);
                        ++settled;
                        if (n_settled) ++*n_settled;
                    } else { keep.push_back(big[k]); keep_ed.push_back(big_ed[k]); keep_nodes.push_back(big_nodes[k]); }
                    if (tested) ++c;
                }
                big.swap(keep); big_ed.swap(keep_ed); big_nodes.swap(keep_nodes);
                if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] wfa2: %zu of %zu leftovers tested against the reference window alone, %zu settled as MaxEditDistance\n", cand.size(), cand.size() + big.size() - (cand.size() - settled), settled); fflush(stderr); }
            }
        }
    }
    if (t_bound) *t_bound = w2_now_ms();
    if (!big.empty()) {
        sw.sub.resize(big.size()); sw.sub_out.resize(big.size()); sw.sub_al.resize(big.size());
        sw.ascii.clear();
        for (size_t k = 0; k < big.size(); ++k) { sw.sub[k] = materialize(big[k], sw.ascii); sw.sub_al[k] = alleles ? alleles[big[k]] : nullptr; }
        // a read that was past the narrow band's edit distance when the compact kernel let go of it starts at full width
        g_wfa_min_ed_hint = big_ed.data();
        const int rc = wfa_assign_batch_v1(sw.sub.data(), sw.sub.size(), prune, max_ed, sw.sub_out.data(), alleles ? sw.sub_al.data() : nullptr, device_id);
        g_wfa_min_ed_hint = nullptr;
        if (rc != HP_OK) return rc;
        if (kernel_ms) *kernel_ms += g_last_kernel_ms;
        if (const char* dbg = std::getenv("HP_DEBUG")) if (std::atoi(dbg) >= 2)   // what the dense-band pass was given, and what came of it
            for (size_t k = 0; k < big.size(); ++k) {
                const hp_wfa_job j = job_header(big[k]);
                uint64_t D = 0, Dmax = 0;
                for (uint32_t v = 0; v < j.n_hets; ++v) { const uint64_t x = std::max<uint64_t>({j.hets[v].ref_len, (j.hets[v].flags & 2u) ? j.hets[v].allele0_len : 0u, j.hets[v].allele1_len}); D += x; Dmax = std::max(Dmax, x); }
                for (uint32_t v = 0; v < j.n_homs; ++v) { const uint64_t x = std::max<uint64_t>({j.homs[v].ref_len, (j.homs[v].flags & 2u) ? j.homs[v].allele0_len : 0u, j.homs[v].allele1_len}); D += x; Dmax = std::max(Dmax, x); }
                fprintf(stderr, "[hp] dense job: read %u b, window %u b, %u + %u variants, D %llu (largest %llu), let go at %u edits, %u nodes -> status %d score %llu\n", dj[big[k]].read_len, dj[big[k]].ref_len,
                        j.n_hets, j.n_homs, (unsigned long long)D, (unsigned long long)Dmax, big_ed[k], Pending::nodes_of(big_nodes[k]), sw.sub_out[k].status, (unsigned long long)sw.sub_out[k].score);
            }
        for (size_t k = 0; k < big.size(); ++k) dst[big[k]] = sw.sub_out[k];
    }
    big.clear(); big_ed.clear(); big_nodes.clear();
    return HP_OK;
}

// The second collection (two phases) and the dense-band pass. Runs on the session's helper thread when run() deferred.
int W2Session::late() {
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed", device_id); return HP_ERR_HIP; }
    const bool trace = std::getenv("HP_STREAM_TRACE") != nullptr;
    const double tl0 = w2_now_ms();
    double tl_early = tl0, tl_tail = tl0, tl_bound = tl0;
    size_t n_early = 0, n_late = 0;
    struct LateTrace { bool on; const double& t0; const double& te; const double& t1; const double& t2; const size_t& ne; const size_t& nl; ~LateTrace() { if (on) fprintf(stderr, "[hp] late: the first collection's %zu leftovers settled (reference-window test + dense-band pass) after %.1f ms, largest class done + held results after %.1f, their %zu leftovers' test after %.1f, dense-band pass after %.1f\n", ne, te - t0, t1 - t0, nl, t2 - t0, w2_now_ms() - t0); } } lt{trace, tl0, tl_early, tl_tail, tl_bound, n_early, n_late};
    // results of jobs the largest class's kernel wrote after run()'s collection, gathered on the device: ids + row offsets up, one
    // record + the allele row per job down. What it could not align joins pend.big.
    hipStream_t s2 = pend.stream2;
    auto collect = [&](const std::vector<uint32_t>& ids, const std::vector<uint32_t>& id_nodes) -> int {
        const size_t h = ids.size();
        int rc;
        std::vector<uint32_t> hoff(h + 1, 0);
        for (size_t k = 0; k < h; ++k) hoff[k + 1] = hoff[k] + dj[ids[k]].n_hets;
        const size_t dn_rec = 0, dn_rows = (h * sizeof(W2HeldRec) + 63) / 64 * 64;
        if ((rc = late_down.reserve(dn_rows + hoff[h] + 64)) != HP_OK) return rc;
        if ((rc = d_held.alloc(h * 4 + 16)) || (rc = d_hoff.alloc((h + 1) * 4 + 16)) || (rc = d_hrec.alloc(h * sizeof(W2HeldRec) + 16)) || (rc = d_hrows.alloc((size_t)hoff[h] + 16))) return rc;
        struct IoDrain { hipStream_t s; ~IoDrain() { dev_io_abort(s); } } io{s2};   // (dev_put / dev_get, hp_common.h: not the runtime's copies)
        if (h) {
            if ((rc = dev_put(d_held.p, ids.data(), h * 4, s2)) != HP_OK || (rc = dev_put(d_hoff.p, hoff.data(), (h + 1) * 4, s2)) != HP_OK) return rc;
            W2MapHeldArgs A{};
            W2MapArgs& M = A.M;
            M.jobs = d_jobs.as<W2Job>(); M.info = d_info.as<W2Info>(); M.n_jobs = (uint32_t)n; M.tags = d_tags.as<uint32_t>();
            M.out_sets = d_sets.as<uint32_t>(); M.status = d_status.as<int32_t>(); M.alleles = d_alleles.as<uint8_t>();
            M.nodes = d_nodes.as<W2Node>(); M.out_work = d_work.as<uint32_t>();
            A.held = d_held.as<uint32_t>(); A.hoff = d_hoff.as<uint32_t>(); A.n_held = (uint32_t)h;
            A.rec = d_hrec.as<W2HeldRec>(); A.rows = d_hrows.as<uint8_t>(); A.out_score = d_score.as<uint64_t>();
            hipLaunchKernelGGL(hp_wfa2_map_held_kernel, dim3((unsigned)((h + 63) / 64)), dim3(64), 0, s2, A);
            HP_HIP_CHECK(hipGetLastError());
            if ((rc = dev_get(late_down.p + dn_rec, d_hrec.p, h * sizeof(W2HeldRec), s2)) != HP_OK) return rc;
            if (hoff[h] && (rc = dev_get(late_down.p + dn_rows, d_hrows.p, hoff[h], s2)) != HP_OK) return rc;
        }
        if (dev_io_sync(s2) != HP_OK) { set_error("WFA kernel failed"); return HP_ERR_HIP; }
        const W2HeldRec* rec = reinterpret_cast<const W2HeldRec*>(late_down.p + dn_rec);
        const uint8_t* rows = late_down.p + dn_rows;
        uint64_t s0 = 0, s1 = 0, s2w = 0, s3 = 0;
        for (size_t hk = 0; hk < h; ++hk) {
            const uint32_t i = ids[hk];
            const int32_t sti = rec[hk].status;
            if (sti == W2_ST_NEED_BIG || sti == W2_ST_PENDING) { pend.big.push_back(i); pend.big_ed.push_back(sti == W2_ST_NEED_BIG ? (uint32_t)(rec[hk].score >> 8) : 0u); pend.big_nodes.push_back(Pending::nodes_of(id_nodes[hk]) | (sti == W2_ST_NEED_BIG ? (uint32_t)(rec[hk].score & 0xFFu) << 24 : 0u)); continue; }   // (PENDING: handed over, never claimed)
            if (sti != W2_ST_OK && sti != W2_ST_MAX_ED) { set_error("job %u: device status %d", i, sti); return HP_ERR_INVARIANT; }
            s0 += rec[hk].work_updates; s1 += rec[hk].work_bytes; s2w += dj[i].read_len; ++s3;
            pend.dst[i].status = sti == W2_ST_OK ? HP_OK : HP_WFA_MAX_ED;
            pend.dst[i].n_nodes = Pending::nodes_of(id_nodes[hk]);
            pend.dst[i].score = rec[hk].score;
            if (pend.alleles && pend.alleles[i] && dj[i].n_hets) std::memcpy(pend.alleles[i], rows + hoff[hk], dj[i].n_hets);
        }
        std::lock_guard<std::mutex> lk(work_m);
        work_updates += s0; work_node_bytes += s1; work_read_bytes += s2w; work_jobs += s3;
        return HP_OK;
    };
    auto dense_pass = [&]() -> int {
        return leftovers_out(pend.big, pend.big_ed, pend.big_nodes, late_sw, pend.dst, pend.alleles, pend.prune, pend.max_ed, false, &tl_bound, nullptr, &late_kernel_ms);
    };
    // What the two smaller classes' kernels left unaligned is known since run()'s collection: its way out starts NOW, beside the
    // largest class's kernel (the tail of the launch set: ~15 ms more on the bench workload), not after it - unless there is so much
    // of it that the wide-table launch below wants it in one piece.
    {
        const char* wenv0 = std::getenv("HP_WFA2_WIDE_MIN");
        const size_t wide_min0 = wenv0 ? (size_t)std::max(0, std::atoi(wenv0)) : (size_t)1024;
        const char* eenv = std::getenv("HP_WFA2_EARLY_DENSE");
        size_t n_large = 0;   // graphs of 257 .. 512 nodes: the 16-word launch takes them when a set has wide_min / 16 of them
        for (uint32_t x : pend.big_nodes) n_large += Pending::nodes_of(x) > (uint32_t)W2Cfg<8>::MAXN ? 1 : 0;
        if (pend.two_phase && !pend.big.empty() && !(eenv && eenv[0] == '0') && (wide_min0 == 0 || (pend.big.size() < wide_min0 && n_large < std::max<size_t>(1, wide_min0 / 16)))) {
            n_early = pend.big.size();
            const int rce = dense_pass();
            if (rce != HP_OK) return rce;
        }
    }
    tl_early = w2_now_ms();
    if (pend.two_phase) {
        const int rc = collect(pend.held, pend.held_nodes);
        if (rc != HP_OK) return rc;
        float ms_wfa = 0.f;
        (void)hipEventElapsedTime(&ms_wfa, ev[2], ev[3]);
        last_span_ms = (double)ms_wfa;
        late_kernel_ms = (double)pend.ms_build + (double)ms_wfa;
    }
    // Many leftovers (a set of reads with 2 % noise and more: most outgrow the smaller classes' slot tables, and the largest class
    // only takes over a small share while its kernel runs): a second launch of the largest class's kernel over them, on the whole
    // device and with wider slot tables (W2Cfg<8, true>), before anything goes to the dense-band pass (HP_WFA2_WIDE_MIN leftovers;
    // 0 = never). They start again from their first base.
    {
        const char* wenv = std::getenv("HP_WFA2_WIDE_MIN");
        const size_t wide_min = wenv ? (size_t)std::max(0, std::atoi(wenv)) : (size_t)1024;
        // (measured at 2 % noise, 108 k leftovers: this launch 147 ms; graphs of up to 128 nodes through an <8,4> launch with the same
        // tables first - 8 reads per wavefront, but 5 workgroups per CU and three tiles per node instead of two - 174 + 18 ms)
        // Graphs of 257 .. 512 nodes (no class of the launch set holds their traversed-node sets; a 20-kb read over 85+ calls builds
        // one) get the same launch with 16-word sets, W2Cfg<16, true>.
        for (int wide16 = 0; wide16 < 2; ++wide16) {
            const uint32_t n_lo = wide16 ? (uint32_t)W2Cfg<8>::MAXN : 0u, n_hi = wide16 ? (uint32_t)W2Cfg<16>::MAXN : (uint32_t)W2Cfg<8>::MAXN;
            size_t n_cand = 0;
            for (uint32_t x : pend.big_nodes) n_cand += (Pending::nodes_of(x) > n_lo && Pending::nodes_of(x) <= n_hi) ? 1 : 0;
            if (!wide_min || n_cand < (wide16 ? std::max<size_t>(1, wide_min / 16) : wide_min)) continue;   // (a large graph costs the dense-band kernels far more than a small one)
            std::vector<uint32_t> ids, id_nodes, keep, keep_ed, keep_nodes;
            for (size_t k = 0; k < pend.big.size(); ++k) {
                const uint32_t nn = Pending::nodes_of(pend.big_nodes[k]);
                if (nn > n_lo && nn <= n_hi) { ids.push_back(pend.big[k]); id_nodes.push_back(pend.big_nodes[k]); }
                else { keep.push_back(pend.big[k]); keep_ed.push_back(pend.big_ed[k]); keep_nodes.push_back(pend.big_nodes[k]); }
            }
            // longest read first, like the class lists
            std::vector<uint32_t> ord(ids.size());
            std::iota(ord.begin(), ord.end(), 0u);
            std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return dj[ids[x]].read_len > dj[ids[y]].read_len; });
            std::vector<uint32_t> up(ids.size() + 20, 0u);
            for (size_t k = 0; k < ids.size(); ++k) up[k] = ids[ord[k]];
            up[ids.size()] = (uint32_t)ids.size();          // n_items_dev
            int rc;
            if ((rc = d_wide.alloc(up.size() * 4)) != HP_OK) return rc;
            if ((rc = dev_put(d_wide.p, up.data(), up.size() * 4, s2)) != HP_OK) return rc;
            // scratch of its own (the session's; wider sets than the class regions of the context), its own tags
            const uint32_t wide_groups = (uint32_t)pend.n_cu * 28u;   // 7 workgroups of 4 groups per CU (LDS)
            const size_t group_dwords = wide16 ? w2_group_dwords<16, true>() : w2_group_dwords<8, true>();
            const size_t wset = (size_t)wide_groups * group_dwords * 4, whash = ((size_t)wide_groups << W2_HCAP_LOG2) * 8;
            const bool fresh = !d_wide_sets.p || d_wide_sets.bytes < wset || !d_wide_hash.p || d_wide_hash.bytes < whash;
            if ((rc = d_wide_sets.alloc(wset)) != HP_OK || (rc = d_wide_hash.alloc(whash)) != HP_OK) return rc;
            if (fresh || (uint64_t)wide_tag_next + n + 2 >= 0xFFFFFFF0ull || wide_last_gen != w2_gen()) {
                wide_last_gen = w2_gen();
                HP_HIP_CHECK(hipMemsetAsync(d_wide_sets.p, 0, wset, s2));
                HP_HIP_CHECK(hipMemsetAsync(d_wide_hash.p, 0, whash, s2));
                wide_tag_next = 0;
            }
            W2Batch B = pend.b2;
            B.gsets = d_wide_sets.as<uint32_t>(); B.htab = d_wide_hash.as<uint64_t>();
            B.set_stride = (uint32_t)group_dwords;
            B.tag_base = wide_tag_next; wide_tag_next += (uint32_t)n + 1u;
            B.order = d_wide.as<uint32_t>(); B.n_items = (uint32_t)ids.size(); B.n_items_dev = d_wide.as<uint32_t>() + ids.size();
            B.next = d_wide.as<uint32_t>() + ids.size() + 4;   // (zero)
            B.esc_role = 0u; B.esc_producers = 0u; B.esc_limit = 0u;
            uint32_t used = 0;
            const double tw0 = w2_now_ms();
            rc = wide16 ? w2_launch<16, 16, true>(B, B.n_items, pend.n_cu, wide_groups, s2, &used) : w2_launch<16, 8, true>(B, B.n_items, pend.n_cu, wide_groups, s2, &used);
            if (rc != HP_OK) return rc;
            pend.big.swap(keep); pend.big_ed.swap(keep_ed); pend.big_nodes.swap(keep_nodes);
            const size_t before = pend.big.size();
            if ((rc = collect(ids, id_nodes)) != HP_OK) return rc;
            if (trace || std::getenv("HP_DEBUG")) fprintf(stderr, "[hp] wfa2: launch with the wide slot tables (%d-word sets) over %zu leftovers, %u groups: %zu aligned, %zu left, %.1f ms\n", wide16 ? 16 : 8, ids.size(), used, ids.size() - (pend.big.size() - before), pend.big.size() - before, w2_now_ms() - tw0);
        }
    }
    tl_tail = tl_bound = w2_now_ms();
    n_late = pend.big.size();
    { const int rcd = dense_pass(); if (rcd != HP_OK) return rcd; }
    return HP_OK;
}

// The records the layout routed past the compact kernels: reference-window test (one wavefront per job, in place: it has to fit
// beside the launch set that has just been queued), then the dense-band pass. On the device's early worker thread.
void W2Session::early_pass() {
    const bool trace = std::getenv("HP_STREAM_TRACE") != nullptr;
    early.t_start = w2_now_ms();
    early.t_bound = early.t_start;
    const size_t n0 = early.big.size();
    int rc = HP_OK;
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed", device_id); rc = HP_ERR_HIP; }
    if (rc == HP_OK) rc = leftovers_out(early.big, early.big_ed, early.big_nodes, early.sw, early.dst, early.alleles, early.prune, early.max_ed, true, &early.t_bound, &early.n_settled, nullptr);
    early.t_done = w2_now_ms();
    if (trace)
        fprintf(stderr, "[hp] early: %zu records routed past the compact kernels; their pass started %.1f ms after the launch set was queued, the reference-window test had settled %zu after %.1f, the dense-band pass of the other %zu was done after %.1f (rc %d)\n",
                n0, early.t_start - early.t_post, early.n_settled, early.t_bound - early.t_post, n0 - early.n_settled, early.t_done - early.t_post, rc);
    {
        std::lock_guard<std::mutex> lk(em);
        early.rc = rc;
        if (rc != HP_OK) early.err = hp_last_error();
        early.done = true;
        ecv.notify_all();   // (under the lock: ~W2Session / wait_early may destroy the condition variable as soon as they see done)
    }
}

int W2Session::finish() {
    const int rc = finish_late();
    aligning_done();   // (every way out of a run ends here)
    std::string err_late;
    if (rc != HP_OK) err_late = hp_last_error();
    const int rce = wait_early();   // (always joined: it writes the caller's result arrays)
    if (rc != HP_OK) { set_error("%s", err_late.c_str()); return rc; }
    return rce;
}

int W2Session::finish_late() {
    if (async_inflight) {   // run() did not wait: the helper thread collects and runs late() in one task
        helper->wait();
        async_inflight = false;
        pend.on = false;
        if (pend.rc != HP_OK) { set_error("%s", pend.err.c_str()); return pend.rc; }
        g_last_kernel_ms = late_kernel_ms;
        return HP_OK;
    }
    if (!pend.on) return HP_OK;
    if (pend.posted) helper->wait(); else pend.rc = late();
    pend.on = false;
    if (pend.rc != HP_OK) { if (pend.posted) set_error("%s", pend.err.c_str()); return pend.rc; }
    g_last_kernel_ms = (pend.two_phase ? 0.0 : g_last_kernel_ms) + late_kernel_ms;
    return HP_OK;
}

int wfa_assign_batch_v2(const hp_wfa_job* jobs, size_t n, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out,
                        uint8_t* const* alleles, int device_id) {
    if (n == 0) return HP_OK;
    if (!jobs || !out) { set_error("null argument"); return HP_ERR_ARG; }
    if (max_ed > 60000) return wfa_assign_batch_v1(jobs, n, prune_distance, max_ed, out, alleles, device_id);   // (soft HP_WFA_UNSUPPORTED for every job)
    W2Session ses;
    int rc = ses.prepare(jobs, n, device_id);
    if (rc != HP_OK) return rc;
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] wfa2: layout + stage + upload %.2f ms\n", ses.last_prepare_ms); fflush(stderr); }
    return ses.run(prune_distance, max_ed, out, alleles);
}

// opaque handle for hp_block.hip
W2Session* w2_session_create() { return new W2Session(); }
void w2_session_destroy(W2Session* s) { delete s; }
int w2_session_prepare(W2Session* s, const hp_wfa_job* jobs, size_t n, int device_id) { return s->prepare(jobs, n, device_id); }
int w2_session_prepare_blocks(W2Session* s, const hp_block_input* in, size_t n_in, const W2JobIn* jobs, size_t n, int device_id) { return s->prepare_blocks(in, n_in, jobs, n, device_id); }
int w2_session_layout_blocks(W2Session* s, const hp_block_input* in, size_t n_in, const W2JobIn* jobs, size_t n) { return s->layout_blocks(in, n_in, jobs, n); }
int w2_session_upload_blocks(W2Session* s, int device_id) { return s->upload_blocks(device_id); }
void w2_session_prepare_stats(const W2Session* s, double prep[4]) { for (int i = 0; i < 4; ++i) prep[i] = s->prep_ms[i]; }
int w2_session_run(W2Session* s, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* const* alleles, int defer) { return s->run(prune_distance, max_ed, out, alleles, defer); }
int w2_session_finish(W2Session* s) { return s->finish(); }
int w2_session_collected(W2Session* s) { return s->wait_collected(); }
// jobs whose results finish() delivers (valid until the next run)
void w2_session_pending(const W2Session* s, const uint32_t** ids, size_t* n) { *ids = s->pend.on ? s->pend.ids.data() : nullptr; *n = s->pend.on ? s->pend.ids.size() : 0; }   // (after w2_session_collected)
double w2_session_span_ms(const W2Session* s) { return s->last_span_ms; }
void w2_session_work(const W2Session* s, uint64_t out[4]) { out[0] = s->work_jobs; out[1] = s->work_read_bytes; out[2] = s->work_node_bytes; out[3] = s->work_updates; }

}  // namespace hp
