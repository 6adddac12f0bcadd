// hp_api.hip — misc entry points of the C ABI (errors, device discovery, version).
#include "hp_common.h"

#include <atomic>
#include <cstdio>
#include <sched.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>

namespace hp {
namespace {
// see hp_common.h: size classes = next power of two up to 1 MiB, then multiples of 1 MiB (so a re-run of a similar
// batch finds its blocks again); at most 2/9 of the device's memory stays cached (64 of 288 GiB), single blocks above an eighth of that never do. The cache is process-wide
// (one mutex; an entry point takes a couple of dozen blocks per call): a block set travels through the threads of a block
// stream - laid out by one, aligned by the next, solved by a third - and whoever lets a buffer go must hand it to whoever needs
// one next, or the first thread would hipMalloc (and synchronise the device) for every set. Never destroyed: threads may still
// hand blocks back while the process exits.
struct DevCache {
    std::mutex m;
    std::multimap<std::pair<int, size_t>, void*> free_;   // (device, bytes) -> block
    size_t cached = 0;
    // what may stay cached: a share of the device's memory (2/9 = 64 of an MI355X's 288 GiB; HP_DEV_CACHE_GB overrides), a single
    // block an eighth of that - a 64 GB part, or a device shared with another allocator, keeps its room. hp_trim_device_cache()
    // hands everything back.
    static size_t max_cached() {
        static const size_t v = [] {
            if (const char* e = std::getenv("HP_DEV_CACHE_GB")) return (size_t)std::max(0ll, std::atoll(e)) << 30;
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) return (size_t)64 << 30;
            return tot / 9 * 2;
        }();
        return v;
    }
    static size_t max_block() { return max_cached() / 8; }
    // size classes: powers of two and 1.5 x powers of two. A request takes any cached block up to twice its size: the buffers of a
    // stage differ a little from set to set (a few hundred leftovers more or less), and a miss is a hipMalloc - which waits for
    // every kernel on the device, the persistent ones of the neighbouring stage included (measured: stalls of 50-400 ms).
    // (round 5: nothing below 1 MB - the few dozen small tables of a side path then all come out of one class, whatever a set's
    // leftovers number, and a miss there is rare after a stream's first sets)
    static size_t round_up(size_t n) {
        size_t r = (size_t)1 << 20;
        while (r < n) { if (r + r / 2 >= n && r >= 4096) return r + r / 2; r <<= 1; }
        return r;
    }
};
DevCache& dev_cache() { static DevCache* c = new DevCache(); return *c; }
}  // namespace

// HP_DEV_CACHE_POISON=1 (debug): every block handed out - recycled or fresh - is filled with 0xA5 first, so that a kernel that reads a
// word before this run wrote it sees neither zeros (a fresh hipMalloc) nor a previous run's plausible values. The whole GPU suite
// must give the same results under it. The fill is a blocking hipMemset: a debugging switch, not a mode to time.
static bool dev_cache_poison() {
    static const bool v = [] { const char* e = std::getenv("HP_DEV_CACHE_POISON"); return e && e[0] != '0'; }();
    return v;
}
static void* poisoned(void* p, size_t n) {
    if (p && dev_cache_poison()) { (void)hipMemset(p, 0xA5, n); (void)hipStreamSynchronize(nullptr); }
    return p;
}

void* dev_cache_get(size_t bytes, size_t* got, int* dev_out) {
    const size_t want = DevCache::round_up(bytes);
    ensure_runtime_flags();
    int dev = 0;
    (void)hipGetDevice(&dev);
    *dev_out = dev;
    DevCache& c = dev_cache();
    {
        std::lock_guard<std::mutex> lk(c.m);
        auto it = c.free_.lower_bound({dev, want});
        if (it != c.free_.end() && it->first.first == dev && it->first.second <= 2 * want) {
            void* p = it->second;
            *got = it->first.second;
            c.cached -= it->first.second;
            c.free_.erase(it);
            return poisoned(p, *got);
        }
    }
    void* p = nullptr;
    g_device_syncing_allocs.fetch_add(1);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {   // give the cache back and try once more
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> lk(c.m);
            for (auto& kv : c.free_) drop.push_back(kv.second);
            c.free_.clear();
            c.cached = 0;
        }
        for (void* q : drop) (void)hipFree(q);
        if (!drop.empty()) e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return nullptr; }
    *got = want;
    return poisoned(p, want);
}

int device_cu_count(int device_id) {
    static std::atomic<int> cached[64];
    if (device_id < 0 || device_id >= 64) return 256;
    int v = cached[device_id].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cached[device_id].store(v, std::memory_order_relaxed);
    return v;
}

// A block stream keeps a dozen HIP streams busy at once (copies, three class launches, the search's two, the helpers'); the
// runtime multiplexes streams onto 4 hardware queues by default, and a kernel that lands behind a persistent graph-WFA kernel
// in a shared queue starts when that one ends. More hardware queues (read by the runtime when it initialises: this runs when
// the library is loaded, before its first HIP call; an explicit setting of the caller's wins).
// (24: a block stream at depth 5 has about twenty streams with work at the same time - six of the alignment stage, one per
// helper thread of the late results, the A* stream sets, the other stages' own)
static const int g_hw_queues_set = [] { return setenv("GPU_MAX_HW_QUEUES", "24", 0); }();
thread_local unsigned g_host_share_div = 0;
std::atomic<int> g_pipelines{0};
thread_local int g_wfa2_reserve_pct = 0;
unsigned host_threads(unsigned want) {
    static const unsigned share = [] {
        if (const char* e = std::getenv("HP_HOST_THREADS")) return (unsigned)std::max(1, std::atoi(e));
        // what this process may really use: the hardware threads, clipped by its affinity mask and by a cgroup CPU quota (a
        // container on a 256-thread host may be allowed 16 CPUs: 32 workers per stage would only take turns on them)
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        {
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) hw = std::min(hw, (unsigned)n); }
            if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
                char q[32] = {0}; long period = 0;
                if (std::fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) { const long quota = std::atol(q); if (quota > 0) hw = std::min(hw, (unsigned)std::max(1l, quota / period)); }
                std::fclose(f);
            } else if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                long quota = -1, period = 0;
                if (std::fscanf(g, "%ld", &quota) != 1) quota = -1;
                std::fclose(g);
                if (FILE* h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(h, "%ld", &period) != 1) period = 0; std::fclose(h); }
                if (quota > 0 && period > 0) hw = std::min(hw, (unsigned)std::max(1l, quota / period));
            }
        }
        unsigned procs = 1;
        if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) procs = (unsigned)std::max(1, std::atoi(e));
        // (... and at most 8 per process: measured on the streamed bench, 8 threads 1.78 M hets/s at 4.5 CPU-seconds per second, 16
        // threads 1.73 M at 7.0 - the stages' parallel regions are short, more workers only spin longer; 4: 1.44 M, 2: 1.32 M)
        return std::max(2u, std::min(8u, hw / procs));
    }();
    // (a stage thread of a block pipeline takes its stage's part of the process's share; several pipelines in one process - one per
    // device - split it between them)
    const unsigned mine = g_host_share_div ? std::max(2u, share / (g_host_share_div * (unsigned)std::max(1, g_pipelines.load(std::memory_order_relaxed)))) : share;
    return std::max(1u, std::min(want, mine));
}

thread_local int g_cu_partition = 0;

// Host threads that wait for the device sleep instead of spinning: a block stream has half a dozen threads waiting on
// streams at any time, and a process that is allowed 16 CPUs (a cgroup quota) is throttled - every thread of it frozen for
// the rest of the scheduler period - when waiters burn the quota that the staging and row-assembly threads need.
// Per device, once, and ONLY for a device whose primary context is not active yet (nothing in the process has created a stream
// or allocated on it): see hp_common.h for what happens otherwise. A host framework that got to the device first keeps its mode
// (HP_BLOCKING_SYNC=0: never set the flag).
static std::atomic<int> g_wait_mode{-1};   // hp_runtime_wait_mode()
// (HP_DEBUG_LATE_WAIT_MODE=1 restores rounds 1-5's behaviour for the reproducer in tests/test_process_order_gpu.py: the flag is set by
// hp_device_count() only, whatever the device's state. Never set it otherwise.)
static bool late_wait_mode() { static const bool v = [] { const char* e = std::getenv("HP_DEBUG_LATE_WAIT_MODE"); return e && e[0] == '1'; }(); return v; }
static void ensure_runtime_flags_impl(bool from_device_count) {
    if (late_wait_mode() && !from_device_count) return;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* e = std::getenv("HP_BLOCKING_SYNC");
        if (e && e[0] == '0') { g_wait_mode.store(0); return; }
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return; }
        int set = 0;
        int cur = 0;
        (void)hipGetDevice(&cur);
        bool moved = false;
        for (int d = 0; d < n; ++d) {
            unsigned int flags = 0;
            int active = 1;
            if (hipDevicePrimaryCtxGetState(d, &flags, &active) != hipSuccess) continue;
            if (active && !late_wait_mode()) {   // whoever initialised the device decided; it may well have decided the same
                if ((flags & hipDeviceScheduleMask) == hipDeviceScheduleBlockingSync) ++set;
                continue;
            }
            if (hipSetDevice(d) == hipSuccess) { moved = true; if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) == hipSuccess) ++set; }
        }
        g_wait_mode.store(n > 0 && set == n ? 1 : set > 0 ? 2 : 0);
        if (moved) (void)hipSetDevice(cur);
        (void)hipGetLastError();
    });
}
void ensure_runtime_flags() { ensure_runtime_flags_impl(false); }
void ensure_runtime_flags_from_device_count() { ensure_runtime_flags_impl(true); }
hipError_t hp_set_device(int device_id) {
    ensure_runtime_flags();
    return hipSetDevice(device_id);
}

// bit i of a CU mask: row a = i / 32, column b = i % 32. Whether the driver deals mask bits to the XCDs in blocks or
// round-robin is not documented; "(a + b) % 8 == 0" selects four CUs of every XCD either way (256 CUs, 8 XCDs).
static bool cu_in_search_partition(int i) {
    static const int mod = [] { const char* e = std::getenv("HP_SEARCH_CU_MOD"); const int v = e ? std::atoi(e) : 8; return (v == 2 || v == 4 || v == 8 || v == 16) ? v : 8; }();
    return ((i / 32 + i % 32) % mod) == 0;
}

int partition_cu_count(int device_id) {
    const int n = device_cu_count(device_id);
    if (g_cu_partition == 0) return n;
    int s = 0;
    for (int i = 0; i < n; ++i) s += cu_in_search_partition(i) ? 1 : 0;
    return g_cu_partition == 1 ? std::max(1, s) : std::max(1, n - s);
}

std::atomic<int> g_device_syncing_allocs{0};   // (diagnostics, HP_STREAM_TRACE: hipMalloc / hipHostMalloc calls - each waits for the whole device)
std::atomic<int> g_streams_created{0};   // (diagnostics: HP_STREAM_TRACE prints it - streams beyond GPU_MAX_HW_QUEUES share hardware queues)
hipError_t hp_stream_create(hipStream_t* s, int device_id, int priority) {
    ensure_runtime_flags();
    g_streams_created.fetch_add(1);
    if (g_cu_partition == 0) {
        if (priority == 0) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        int prio_lo = 0, prio_hi = 0;   // (numerically: the greatest priority is the lowest number)
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority > 0 ? prio_hi : prio_lo);
    }
    const int n = device_cu_count(device_id);
    std::vector<uint32_t> mask((size_t)(n + 31) / 32, 0u);
    for (int i = 0; i < n; ++i)
        if (cu_in_search_partition(i) == (g_cu_partition == 1)) mask[(size_t)i / 32] |= 1u << (i % 32);
    return hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
}

// ---- dev_put / dev_get (hp_common.h) ------------------------------------------------------------------------------------
// (single-wavefront workgroups: beside a launch set that holds 92 % of the wavefront slots a CU has one slot free, rarely four)
__global__ void __launch_bounds__(64) hp_copy_kernel(uint8_t* dst, const uint8_t* src, size_t n, uint32_t vec) {
    if (vec) {   // both 16-byte aligned
        const size_t stride = (size_t)gridDim.x * 64u * 16u;
        for (size_t i = ((size_t)blockIdx.x * 64u + threadIdx.x) * 16u; i + 16u <= n; i += stride) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
        if (blockIdx.x == 0 && threadIdx.x < (n & 15u)) dst[(n & ~(size_t)15) + threadIdx.x] = src[(n & ~(size_t)15) + threadIdx.x];
    } else {
        const size_t stride = (size_t)gridDim.x * 64u;
        for (size_t i = (size_t)blockIdx.x * 64u + threadIdx.x; i < n; i += stride) dst[i] = src[i];
    }
}
namespace {
struct IoArena {
    PinBuf buf;
    size_t used = 0, wanted = 0;   // wanted: what the transfers since the last sync would have needed (the arena grows after the sync)
    struct Get { void* dst; size_t off, n; };
    std::vector<Get> gets;
    // a slice of the arena, or nullptr when it is full while transfers are in flight (the caller then falls back on the runtime's copy)
    uint8_t* take(size_t n, size_t* off) {
        if (n > ((size_t)16 << 20)) return nullptr;   // bulk data: the copy engines' job
        const size_t need = (n + 63) & ~(size_t)63;
        wanted += need;
        if (used + need > buf.cap) {
            if (used != 0) return nullptr;
            if (buf.reserve(std::max<size_t>(need, (size_t)4 << 20)) != HP_OK) return nullptr;   // (grows only while nothing is in flight)
        }
        *off = used; used += need;
        return buf.p + *off;
    }
};
thread_local IoArena g_io;
void launch_copy(void* dst, const void* src, size_t n, hipStream_t st) {
    const uint32_t vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0 ? 1u : 0u;
    const size_t per_wg = vec ? 1024 : 256;   // one 16-byte (or four 1-byte) moves per lane and pass
    hipLaunchKernelGGL(hp_copy_kernel, dim3((unsigned)std::min<size_t>(2048, (n + per_wg - 1) / per_wg)), dim3(64), 0, st, static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), n, vec);
}
}  // namespace
int dev_copy(void* dst, const void* src, size_t n, hipStream_t st) {
    if (n == 0) return HP_OK;
    launch_copy(dst, src, n, st);
    HP_HIP_CHECK(hipGetLastError());
    return HP_OK;
}
int dev_put(void* d_dst, const void* h_src, size_t n, hipStream_t st) {
    if (n == 0) return HP_OK;
    size_t off = 0;
    uint8_t* a = g_io.take(n, &off);
    if (!a) { HP_HIP_CHECK(hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, st)); return HP_OK; }
    std::memcpy(a, h_src, n);
    launch_copy(d_dst, a, n, st);
    HP_HIP_CHECK(hipGetLastError());
    return HP_OK;
}
int dev_get(void* h_dst, const void* d_src, size_t n, hipStream_t st) {
    if (n == 0 || !h_dst) return HP_OK;
    size_t off = 0;
    uint8_t* a = g_io.take(n, &off);
    if (!a) { HP_HIP_CHECK(hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, st)); return HP_OK; }
    launch_copy(a, d_src, n, st);
    HP_HIP_CHECK(hipGetLastError());
    g_io.gets.push_back({h_dst, off, n});
    return HP_OK;
}
int dev_io_sync(hipStream_t st) {
    const hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) for (const IoArena::Get& g : g_io.gets) std::memcpy(g.dst, g_io.buf.p + g.off, g.n);
    g_io.gets.clear();
    g_io.used = 0;
    if (g_io.wanted > g_io.buf.cap && g_io.wanted <= ((size_t)64 << 20)) (void)g_io.buf.reserve(g_io.wanted);   // room for the same transfers next time
    g_io.wanted = 0;
    if (e != hipSuccess) { set_error("HIP error %s while waiting for a stream", hipGetErrorString(e)); return HP_ERR_HIP; }
    return HP_OK;
}

// the guards' way out (an error return between a dev_get and its dev_io_sync): waits for the stream - queued copies may still read
// or write the arena and the caller's buffers - and FORGETS the pending gets instead of delivering them: their destinations may
// be locals that were declared after the guard and are gone by the time it runs
void dev_io_abort(hipStream_t st) {
    (void)hipStreamSynchronize(st);
    g_io.gets.clear();
    g_io.used = 0;
    g_io.wanted = 0;
}

thread_local bool g_thread_stream_high = false;
hipStream_t thread_stream(int device_id) {
    struct Slot { int device = -1; hipStream_t s = nullptr; };
    struct Holder { Slot part[3]; ~Holder() { for (auto& x : part) if (x.s) (void)hipStreamDestroy(x.s); } };
    static thread_local Holder h;
    Slot& x = h.part[g_cu_partition];
    if (x.s && x.device == device_id) return x.s;
    if (x.s) { (void)hipStreamDestroy(x.s); x.s = nullptr; }
    // The kernels on a thread's own stream are the latency chains behind a launch set - the reference-window test, the dense-band
    // pass over a set's leftovers, the Levenshtein batch of the fallbacks - a few hundred workgroups that the rows stage of a block
    // stream WAITS for, beside the next sets' persistent alignment kernels that fill every compute unit: high priority, so that the
    // wavefront slots those kernels' retiring workgroups free go to the chain first (HP_CHAIN_PRIORITY=1; measured round 5: no effect on the first set's latency, 146-153 ms either way - off by default).
    static const int prio = [] { const char* e = std::getenv("HP_CHAIN_PRIORITY"); return e ? std::atoi(e) : 0; }();
    // (g_thread_stream_high: the device's early worker - hp_wfa2.hip - asks for a high-priority stream for the hardware QUEUE it comes
    // with: a process may hold more streams than the runtime has queues (a stream seven sets deep: 23 of GPU_MAX_HW_QUEUES = 24; more with other entries in use),
    // streams of one priority share them, and a chain's kernels behind a persistent class kernel in one queue wait for it to end -
    // measured: the reference-window test 50-60 instead of 7-10 ms, and the class kernels behind a dense-band pass 23-27 instead of 19)
    if (hp_stream_create(&x.s, device_id, g_thread_stream_high ? 1 : prio) != hipSuccess) { x.s = nullptr; return nullptr; }
    x.device = device_id;
    return x.s;
}

void dev_cache_put(void* p, size_t bytes, int dev) {   // dev: what dev_cache_get reported for this block
    DevCache& c = dev_cache();
    {
        std::lock_guard<std::mutex> lk(c.m);
        if (bytes <= DevCache::max_block() && c.cached + bytes <= DevCache::max_cached()) {
            c.free_.insert({{dev, bytes}, p});
            c.cached += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

extern "C" size_t hp_trim_device_cache(void) {   // frees every cached device block (of every device); returns the bytes handed back
    DevCache& c = dev_cache();
    std::vector<void*> drop;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(c.m);
        for (auto& kv : c.free_) drop.push_back(kv.second);
        bytes = c.cached;
        c.free_.clear();
        c.cached = 0;
    }
    for (void* q : drop) (void)hipFree(q);
    return bytes;
}

// ---- host memory the device reads in place (hp_host_alloc): a registry of the ranges, looked up by the layout stage ----------------
namespace {
struct HostRanges { std::mutex m; std::map<uintptr_t, size_t> r; };   // base -> bytes (slack included)
HostRanges& host_ranges() { static HostRanges* h = new HostRanges(); return *h; }
}  // namespace
extern "C" void* hp_host_alloc(size_t bytes) {
    void* p = nullptr;
    const size_t n = bytes + 64;
    ensure_runtime_flags();
    g_device_syncing_allocs.fetch_add(1);
    if (hipHostMalloc(&p, n, hipHostMallocPortable) != hipSuccess || !p) { set_error("hipHostMalloc(%zu) failed", n); return nullptr; }
    HostRanges& H = host_ranges();
    std::lock_guard<std::mutex> lk(H.m);
    H.r[(uintptr_t)p] = n;
    return p;
}
extern "C" void hp_host_free(void* p) {
    if (!p) return;
    { HostRanges& H = host_ranges(); std::lock_guard<std::mutex> lk(H.m); H.r.erase((uintptr_t)p); }
    (void)hipHostFree(p);
}
bool host_range_of(const void* p, uintptr_t* lo, uintptr_t* hi) {   // the hp_host_alloc range p lies in
    HostRanges& H = host_ranges();
    std::lock_guard<std::mutex> lk(H.m);
    if (H.r.empty()) return false;
    auto it = H.r.upper_bound((uintptr_t)p);
    if (it == H.r.begin()) return false;
    --it;
    if ((uintptr_t)p >= it->first + it->second) return false;
    *lo = it->first; *hi = it->first + it->second;
    return true;
}
std::atomic<uint64_t> g_in_place_bytes{0};
extern "C" uint64_t hp_host_in_place_bytes(void) { return g_in_place_bytes.load(); }
std::atomic<uint64_t> g_routed_records{0};
extern "C" uint64_t hp_wfa_routed_records(void) { return g_routed_records.load(); }
bool host_ranges_any() { HostRanges& H = host_ranges(); std::lock_guard<std::mutex> lk(H.m); return !H.r.empty(); }

static std::atomic<int> g_coalesce{-1};   // -1: ask the environment once
bool coalescing_enabled() {
    int v = g_coalesce.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = std::getenv("HP_COALESCE");
        v = (e && e[0] == '0') ? 0 : 1;
        g_coalesce.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

static thread_local std::string g_err;
thread_local double g_last_kernel_ms = 0.0;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
}  // namespace hp

extern "C" {

const char* hp_last_error(void) { return hp::g_err.c_str(); }
const char* hp_version(void) { return "hiphase_gpu 0.2.0 (gfx950)"; }
int hp_set_coalescing(int on) {
    const int prev = hp::coalescing_enabled() ? 1 : 0;
    hp::g_coalesce.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}
double hp_last_kernel_ms(void) { return hp::g_last_kernel_ms; }
// how the library's host threads wait for the device: -1 not decided yet (no entry point has touched a device), 1 blocking
// (hipDeviceScheduleBlockingSync set on every device, all of them untouched when the library got there), 2 on some, 0 on none (the
// process had used the devices already - they keep the mode they were initialised in - or HP_BLOCKING_SYNC=0)
int hp_runtime_wait_mode(void) { return hp::g_wait_mode.load(); }

int hp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    hp::ensure_runtime_flags_from_device_count();
    return n;
}

// One process per GPU under torch.distributed / torchrun: LOCAL_RANK selects the device unless
// HP_DEVICE overrides it.
int hp_default_device(void) {
    const char* e = std::getenv("HP_DEVICE");
    if (!e) e = std::getenv("LOCAL_RANK");
    int d = e ? std::atoi(e) : 0;
    int n = hp_device_count();
    if (n > 0 && d >= n) d %= n;
    return d < 0 ? 0 : d;
}

}  // extern "C"
