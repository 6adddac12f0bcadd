// hp_common.h — shared host-side helpers for libhiphase_gpu.so (error slots, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <pthread.h>
#include <thread>
#include <vector>

#include "../../include/hiphase_gpu.h"

namespace hp {

void set_error(const char* fmt, ...);
extern thread_local double g_last_kernel_ms;  // see hp_last_kernel_ms()  // thread-local message returned by hp_last_error()

#define HP_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            hp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (_e == hipErrorOutOfMemory) ? HP_ERR_OOM : HP_ERR_HIP;                          \
        }                                                                                          \
    } while (0)

// Per-thread cache of device allocations: every entry point allocates a couple of dozen buffers per call, and
// hipMalloc / hipFree cost ~100 us each and synchronise the device. Blocks are handed back to the cache instead of
// being freed and re-used by later calls of the same thread on the same device (sizes are rounded up so that they
// match); the cache is bounded and released at thread exit. hp_common.h declares, hp_api.hip defines.
void* dev_cache_get(size_t bytes, size_t* got, int* dev);   // nullptr on failure (error set); *dev = the device it lives on
void dev_cache_put(void* p, size_t bytes, int dev);

// Host threads a parallel region of the library may use: min(want, the process's share of the host). The share is
// HP_HOST_THREADS if set, else hardware threads / processes per node (LOCAL_WORLD_SIZE as torchrun and friends export it: one
// process per GPU, eight of them on one host must not each assume the whole machine), never below 2.
unsigned host_threads(unsigned want);
// a pipeline stage (hp_stream.hip) runs beside two others: its parallel regions take this fraction of the share (per thread; 0 = all)
extern thread_local unsigned g_host_share_div;
extern std::atomic<int> g_pipelines;   // block pipelines alive in this process (hp_stream.hip): they share the host threads
// share (per cent) of the chip's wavefront slots the persistent graph-WFA launch set leaves empty for the kernels of other
// threads (hp_wfa2.hip run(); set by the alignment stage of a block stream, 0 elsewhere)
extern thread_local int g_wfa2_reserve_pct;

// The library's one-time runtime setup (hp_api.hip), before its first stream / allocation / hipSetDevice in the process:
// hipDeviceScheduleBlockingSync for every device whose primary context is NOT active yet. Every hipSetDevice of the library goes
// through hp_set_device. (Round 6: the flag used to be set by hp_device_count() only - i.e. whenever an entry point that asks for
// the device count was first called. Set on a device that already has streams, ROCm 7.2 does not refuse it (older runtimes
// returned hipErrorSetOnActiveProcess): the device switches from spinning on completion signals to waiting for their
// interrupt handlers, the signals of the streams that exist were created without interrupts,
// hsa_amd_signal_async_handler() fails on them ("failed to set the handler!", AMD_LOG_LEVEL=1) and whoever waits for such a
// command's completion through the handler path - hipHostFree / hipFree synchronising every stream - waits for ever. That was
// the "kernel that never ends" of rounds 4-5: no kernel at all)
void ensure_runtime_flags();
hipError_t hp_set_device(int device_id);

// number of compute units of a device, queried once (hipGetDeviceProperties costs milliseconds per call)
int device_cu_count(int device_id);

// CU partitions (block pipeline, hp_block.hip): while the graph-WFA of one chunk of blocks runs, the search of the chunk
// before it runs on compute units of its own. 0 = the whole device; 1 = the search partition (one CU in eight, the same
// number in every XCD); 2 = the other seven. The setting is per thread: a stream created while it is set is bound to
// that partition's CUs (and gets a hardware queue of its own), and the persistent kernels size their grids to it.
extern thread_local int g_cu_partition;
// priority: +1 the latency-critical kernels of the search, 0 normal, -1 the persistent throughput kernels of the alignment stage
hipError_t hp_stream_create(hipStream_t* s, int device_id, int priority = 0);
extern std::atomic<int> g_streams_created;
extern std::atomic<int> g_device_syncing_allocs;
int partition_cu_count(int device_id);   // CUs of the calling thread's current partition
// The calling thread's own stream on `device` in its current CU partition (created on first use, lives as long as the thread):
// what the small entry points launch on. Nothing in the library uses the NULL stream or hipDeviceSynchronize - a stage of a block
// stream must never wait for another stage's kernels.
hipStream_t thread_stream(int device_id);
extern thread_local bool g_thread_stream_high;   // set before the thread's first thread_stream(): a high-priority stream (a hardware queue of its own)

// Library threads carry names (`hp-...`, /proc/<pid>/task/*/comm): bench.py attributes the process's CPU time to them.
inline void name_thread(const char* name) { (void)pthread_setname_np(pthread_self(), name); }
inline void name_thread_after_creator(const char* creator, char suffix) {   // "<creator's name><suffix>"
    char b[16];
    std::snprintf(b, sizeof b, "%.13s%c", creator, suffix);
    name_thread(b);
}

// A helper thread that lives as long as its owner: its thread-local device-buffer cache (below) then survives from one
// task to the next (a fresh thread would hipMalloc every buffer again and hipFree it at exit, synchronising the device).
struct HelperThread {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> task;
    bool has_task = false, busy = false, quit = false;
    void start() {
        char who[16] = "hp";
        (void)pthread_getname_np(pthread_self(), who, sizeof who);
        const std::string creator = who;
        th = std::thread([this, creator]() {
            name_thread_after_creator(creator.c_str(), 'h');
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this]() { return has_task || quit; });
                if (quit) return;
                std::function<void()> t = std::move(task);
                has_task = false;
                lk.unlock();
                t();
                lk.lock();
                busy = false;
                cv.notify_all();
            }
        });
    }
    void post(std::function<void()> t) {
        std::unique_lock<std::mutex> lk(m);
        task = std::move(t); has_task = true; busy = true;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this]() { return !busy; });
    }
    ~HelperThread() {
        if (th.joinable()) {
            { std::unique_lock<std::mutex> lk(m); quit = true; cv.notify_all(); }
            th.join();
        }
    }
};

// Host worker threads shared by the stages that fan out over blocks / jobs (row assembly, outputs, result scatter, A*
// pack): a stage used to start and join its own std::threads, which cost more than the work on a 40 ms step. The pool is
// created on first use and never destroyed (its threads may be parked in a condition variable at exit). One parallel
// region at a time: a caller that finds the pool busy (another host thread's region, or a nested one) starts threads of
// its own as before. A worker's thread-local caches (device buffers, error slot) live as long as the process.
class WorkerPool {
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    std::function<void(unsigned)> job;
    unsigned want = 0, started = 0, done = 0;
    bool busy = false, quit = false;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [this]() { return started < want || quit; });
            if (quit) return;
            const unsigned id = started++;
            lk.unlock();
            job(id);
            lk.lock();
            if (++done == want) cv_done.notify_all();
        }
    }
    static WorkerPool*& tl_pool() { static thread_local WorkerPool* p = nullptr; return p; }
public:
    WorkerPool() = default;
    WorkerPool(const WorkerPool&) = delete;
    // (only pools owned by a block stream are ever destroyed; the process-wide one lives as long as the process)
    ~WorkerPool() {
        { std::unique_lock<std::mutex> lk(m); quit = true; cv.notify_all(); }
        for (auto& x : th) if (x.joinable()) x.join();
    }
    // The pool of the calling thread: a pipeline stage of a block stream (hp_stream.hip) brings its own, so that the stages'
    // parallel regions do not queue behind each other; everyone else shares the process-wide one.
    static WorkerPool& get() { if (tl_pool()) return *tl_pool(); static WorkerPool* p = new WorkerPool(); return *p; }
    static void set_thread_pool(WorkerPool* p) { tl_pool() = p; }
    // f(0) .. f(nt - 1), each on a thread of its own (the caller runs f(nt - 1)); returns when all are done
    template <class F> void run(unsigned nt, F&& f) {
        if (nt <= 1) { f(0u); return; }
        std::unique_lock<std::mutex> lk(m);
        if (busy) {
            lk.unlock();
            std::vector<std::thread> own;
            for (unsigned t = 0; t + 1 < nt; ++t) own.emplace_back([&f, t]() { f(t); });
            f(nt - 1);
            for (auto& x : own) x.join();
            return;
        }
        busy = true;
        if (th.size() + 1 < nt) {
            char who[16] = "hp";
            (void)pthread_getname_np(pthread_self(), who, sizeof who);
            const std::string creator = who;
            while (th.size() + 1 < nt) th.emplace_back([this, creator]() { name_thread_after_creator(creator.c_str(), 'w'); loop(); });
        }
        job = [&f](unsigned t) { f(t); };
        want = nt - 1; started = 0; done = 0;
        cv.notify_all();
        lk.unlock();
        f(nt - 1);
        lk.lock();
        cv_done.wait(lk, [this]() { return done == want; });
        want = 0; started = 0; done = 0;
        busy = false;
    }
};

// HP_SEQ_BAM4 -> one byte per base, htslib's table (what read.seq().as_bytes() returns): base k of the record sits in byte
// k / 2, high nibble first. Host-side decode for the few reads that leave the device paths (dense-band leftovers, local
// re-alignment); the bulk of the reads is expanded on the device (hp_wfa2_unpack_kernel).
inline void decode_bam4(const uint8_t* src, uint64_t first_base, uint64_t n, uint8_t* dst) {
    static const char tab[17] = "=ACMGRSVTWYHKDBN";
    // two bases per source byte through a 256-entry table (the high nibble is the first base = the low byte of the pair): the
    // fallbacks of a block set are a few hundred 15-kb reads, base by base that was 5 ms of a host thread
    static const struct Pairs { uint16_t v[256]; Pairs() { for (int b = 0; b < 256; ++b) v[b] = (uint16_t)((uint8_t)tab[b >> 4] | ((uint16_t)(uint8_t)tab[b & 15] << 8)); } } pairs;
    uint64_t k = 0;
    if ((first_base & 1u) && n) { dst[0] = (uint8_t)tab[src[first_base >> 1] & 15u]; k = 1; }
    const uint8_t* s = src + ((first_base + k) >> 1);
    for (; k + 2 <= n; k += 2, ++s) { const uint16_t v = pairs.v[*s]; __builtin_memcpy(dst + k, &v, 2); }
    if (k < n) dst[k] = (uint8_t)tab[*s >> 4];
}

// hp_local.hip: local_realignment (reference src/read_parsing.rs:121-503) for the records of several blocks in one go - every
// group with its own variant list; alleles / quals: n_reads x n_variants rows; stats may be NULL
struct LocalGroup {
    const hp_local_read* reads; size_t n_reads;
    const hp_local_variant* variants; size_t n_variants;
    uint8_t* alleles; uint8_t* quals; hp_read_stats* stats;
};
int local_realign_groups(LocalGroup* groups, size_t n_groups, int device_id);

// grow-only pinned staging (DMA straight from it; never value-initialised)
struct PinBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    int reserve(size_t n) {
        if (n <= cap) return HP_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = std::max<size_t>(2 * n + 4096, (size_t)4 << 20);   // (growing pinned memory synchronises the device too: leave room - twice the need, 4 MB at least)
        ensure_runtime_flags();
        g_device_syncing_allocs.fetch_add(1);
        if (hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault) != hipSuccess) {
            p = nullptr;
            set_error("hipHostMalloc(%zu) failed", want);
            return HP_ERR_OOM;
        }
        cap = want;
        return HP_OK;
    }
};

// Small host <-> device transfers of the side paths (job lists, statuses, result rows: kilobytes to a few megabytes), done by a
// copy KERNEL on the caller's stream between the device and the calling thread's pinned arena instead of hipMemcpy[Async]:
// inside a block stream the runtime's copies - its staging of pageable memory, the NULL stream, the copy engines working
// through the 48 MB pieces of the next set's reads - made a 1 ms step of the A* stage take 25-60 ms every few sets.
//   dev_put: h_src is copied into the arena at once (the caller may reuse it), the device copy is queued on st
//   dev_get: queued on st; h_dst is filled by dev_io_sync
//   dev_io_sync: waits for st, delivers the gets, empties the arena. (Everything a thread queued since its last dev_io_sync
//   must be on ONE stream.)
//   dev_copy: the copy kernel alone, between any two device-visible ranges (device memory, pinned host memory)
int dev_copy(void* dst, const void* src, size_t n, hipStream_t st);
int dev_put(void* d_dst, const void* h_src, size_t n, hipStream_t st);
int dev_get(void* h_dst, const void* d_src, size_t n, hipStream_t st);
int dev_io_sync(hipStream_t st);
//   dev_io_abort: waits for st and drops the pending gets undelivered (scope guards on error paths: after a successful dev_io_sync
//   there is nothing pending and it only waits)
void dev_io_abort(hipStream_t st);
// [p, p + n) lies in memory from hp_host_alloc (pinned, mapped: kernels read it in place); any such memory at all
bool host_range_of(const void* p, uintptr_t* lo, uintptr_t* hi);   // the hp_host_alloc range (slack included) that holds p
bool host_ranges_any();
extern std::atomic<uint64_t> g_in_place_bytes;
extern std::atomic<uint64_t> g_routed_records;   // hp_wfa_routed_records()

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int dev = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) dev_cache_put(p, bytes, dev);
        p = nullptr;
        bytes = 0;
    }
    // grows only: hipFree/hipMalloc synchronise the whole device, which would serialise the two solver streams
    int alloc(size_t n) {
        if (n == 0) n = 16;
        if (p && bytes >= n) return HP_OK;
        release();
        p = dev_cache_get(n, &bytes, &dev);
        if (!p) { bytes = 0; return HP_ERR_OOM; }
        return HP_OK;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace hp
