// hp_astar.hip — host side of the A* solver: packs hp_block_view matrices into the bit-sliced HBM layout,
// owns the device pools/scratch, launches hp_astar_kernel and implements the hp_astar_* / hp_batch_* C ABI.
//
// Boundary: replaces reference src/astar_phaser.rs:426-429 `astar_solver(...)` as called from
// reference src/phaser.rs:541-543. See include/hiphase_gpu.h for the contract.
#include "hp_astar_kernel.hip"
#include "hp_combine.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

namespace hp {

namespace {

struct HostPack {
    uint64_t qual_limit = 1ull << 35;  // total quality mass a block may have: the cost field of the packed keys (2^29 with the sub-solver's wide-index keys)
    std::vector<BlockDesc> desc;
    std::vector<uint32_t> vlo, vhi;
    std::vector<uint8_t> vflags;
    std::vector<uint32_t> rstart, rend, rword;
    uint64_t n_words = 0;              // plane words of all rows (built on the device, hp_pack_words_kernel)
    std::vector<uint64_t> rcell;       // packed row -> offset of its first cell in the block's caller arrays
    std::vector<uint8_t> raw_alleles;  // the caller's alleles_2bit / quals of every block, concatenated (byte-aligned)
    std::vector<uint8_t> raw_quals;
    std::vector<PackRaw> raw;          // per block: where its bytes start
    std::vector<uint64_t> work;  // LPT estimate per block
    std::vector<uint32_t> row_block;   // packed row -> block
    std::vector<uint32_t> row_orig;    // packed row -> caller's row index inside its block
    std::vector<uint64_t> caller_row_off;  // per block: offset of its rows in the caller's concatenated order
    uint64_t caller_rows = 0;
    uint64_t chunk_total = 0;
    uint64_t h_total = 0;
    uint64_t cell_total = 0;   // entries of the per-position cell tables (device-built)
    uint32_t max_n = 0;
};

inline uint8_t cell_allele(const hp_block_view* v, uint64_t cell) {
    return (v->alleles_2bit[cell >> 2] >> (2 * (cell & 3))) & 3;
}

// Packs one block: sorts rows by start, drops inert rows, builds per-variant candidate ranges and the
// bit-sliced plane words. Returns HP_OK or an HP_ERR_* code (message in hp_last_error()).
int pack_block(const hp_block_view* v, HostPack& hpk) {
    const uint32_t N = v->n_variants, R = v->n_reads;
    if (N == 0) { set_error("block with 0 variants (phaser.rs:415-434 short-circuits those before the solver)"); return HP_ERR_ARG; }
    if (N >= (1u << 24)) { set_error("N=%u >= 2^24 exceeds the packed priority-key limit", N); return HP_ERR_UNSUPPORTED; }
    {   // test hook: blocks of exactly this many variants count as beyond the limits (the real ones need > 10^8 cells in one block)
        const char* e = std::getenv("HP_TEST_UNSUPPORTED_N");
        const long test_n = e ? std::atol(e) : -1l;
        if (test_n >= 0 && (long)N == test_n) { set_error("N=%u: HP_TEST_UNSUPPORTED_N", N); return HP_ERR_UNSUPPORTED; }
    }
    if (R && (!v->read_start || !v->read_end || !v->row_off || !v->alleles_2bit || !v->quals)) {
        set_error("null array in hp_block_view"); return HP_ERR_ARG;
    }
    if (!v->var_flags) { set_error("null var_flags"); return HP_ERR_ARG; }
    std::vector<uint32_t> idx;
    idx.reserve(R);
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t s = v->read_start[r], e = v->read_end[r];
        if (s > e || e > N || v->row_off[r + 1] < v->row_off[r] || v->row_off[r + 1] - v->row_off[r] != (uint64_t)(e - s)) {
            set_error("row %u: inconsistent region [%u,%u) / row_off", r, s, e); return HP_ERR_ARG;
        }
        if (e > s) idx.push_back(r);
    }
    // rows ordered by (start, end, caller index): a counting sort over the N possible starts, then the (short) runs
    // of equal start are ordered by end; rows of equal (start, end) keep the caller's order
    {
        std::vector<uint32_t> first(N + 2, 0);
        for (uint32_t r : idx) first[v->read_start[r] + 1]++;
        for (uint32_t p = 0; p <= N; ++p) first[p + 1] += first[p];
        std::vector<uint32_t> sorted(idx.size());
        {
            std::vector<uint32_t> cur(first.begin(), first.end() - 1);
            for (uint32_t r : idx) sorted[cur[v->read_start[r]]++] = r;   // idx is ascending in r: stable
        }
        for (uint32_t p = 0; p <= N; ++p) {
            const uint32_t a = first[p], b = first[p + 1];
            if (b - a > 16)
                std::stable_sort(sorted.begin() + a, sorted.begin() + b, [&](uint32_t x, uint32_t y) { return v->read_end[x] < v->read_end[y]; });
            else
                for (uint32_t i = a + 1; i < b; ++i) {   // stable insertion sort of a short run
                    const uint32_t x = sorted[i], ex = v->read_end[x];
                    uint32_t j = i;
                    while (j > a && v->read_end[sorted[j - 1]] > ex) { sorted[j] = sorted[j - 1]; --j; }
                    sorted[j] = x;
                }
        }
        idx.swap(sorted);
    }
    BlockDesc d{};
    d.n_vars = N;
    d.n_reads = (uint32_t)idx.size();
    d.var_off = hpk.vlo.size();
    d.read_off = hpk.rstart.size();
    d.word_off = hpk.n_words;
    d.h_off = hpk.h_total;
    hpk.h_total += (uint64_t)N + 1;
    d.chunk_off = hpk.chunk_total;
    hpk.chunk_total += ((uint64_t)N + 31) / 32;
    hpk.caller_row_off.push_back(hpk.caller_rows);
    hpk.caller_rows += R;
    const uint32_t blk_index = (uint32_t)hpk.desc.size();

    const size_t v0 = hpk.vlo.size();
    hpk.vlo.resize(v0 + N, 0xFFFFFFFFu);
    hpk.vhi.resize(v0 + N, 0);
    hpk.vflags.insert(hpk.vflags.end(), v->var_flags, v->var_flags + N);
    for (uint32_t p = 0; p < N; ++p) hpk.vflags[v0 + p] &= (uint8_t)(HP_VAR_IGNORED | HP_VAR_SNV);   // bit 2 is VAR_NOFAST (device-only)

    // The rows' cells stay in the caller's layout: they are uploaded as they are and the bit-sliced plane words are
    // built on the device (hp_pack_words_kernel). The host keeps the cheap parts: the quality sums behind the packed
    // limits and the ignored-variant invariant (only rows crossing an ignored variant are looked at).
    uint64_t n_words = 0, cells = 0, max_row_qual = 0, total_qual = 0;
    uint32_t max_row_len = 0;
    const uint64_t n_cells_blk = R ? v->row_off[R] : 0;
    hpk.raw.push_back(PackRaw{hpk.raw_alleles.size(), hpk.raw_quals.size()});
    if (n_cells_blk) {
        hpk.raw_alleles.insert(hpk.raw_alleles.end(), v->alleles_2bit, v->alleles_2bit + (n_cells_blk + 3) / 4);
        hpk.raw_quals.insert(hpk.raw_quals.end(), v->quals, v->quals + n_cells_blk);
    }
    std::vector<uint32_t> ignored;
    for (uint32_t p = 0; p < N; ++p) if (v->var_flags[p] & HP_VAR_IGNORED) ignored.push_back(p);
    const size_t row0 = hpk.rstart.size();
    hpk.rstart.resize(row0 + idx.size()); hpk.rend.resize(row0 + idx.size()); hpk.row_block.resize(row0 + idx.size());
    hpk.row_orig.resize(row0 + idx.size()); hpk.rword.resize(row0 + idx.size()); hpk.rcell.resize(row0 + idx.size());
    uint32_t *p_rstart = hpk.rstart.data() + row0, *p_rend = hpk.rend.data() + row0, *p_row_block = hpk.row_block.data() + row0,
             *p_row_orig = hpk.row_orig.data() + row0, *p_rword = hpk.rword.data() + row0;
    uint64_t* p_rcell = hpk.rcell.data() + row0;
    for (uint32_t i = 0; i < idx.size(); ++i) {
        const uint32_t r = idx[i];
        const uint32_t s = v->read_start[r], e = v->read_end[r];
        const uint32_t k0 = s >> 5, k1 = (e - 1) >> 5;
        if (n_words > 0xFFFFFFF0ull) { set_error("block too large (plane words)"); return HP_ERR_UNSUPPORTED; }
        if (v->row_off[r] + (e - s) > n_cells_blk) { set_error("row %u: row_off outside the cell arrays", r); return HP_ERR_ARG; }
        p_rstart[i] = s;
        p_rend[i] = e;
        p_row_block[i] = blk_index;
        p_row_orig[i] = r;
        p_rword[i] = (uint32_t)n_words;
        p_rcell[i] = v->row_off[r];
        const uint64_t ro = v->row_off[r];
        if (!ignored.empty())
        for (auto it = std::lower_bound(ignored.begin(), ignored.end(), s); it != ignored.end() && *it < e; ++it) {
            const uint8_t a = cell_allele(v, ro + (*it - s));
            if (a != HP_ALLELE_NOOVERLAP) {
                set_error("row %u has allele %u at ignored variant %u (astar_phaser.rs:435-442 assert)", r, a, *it);
                return HP_ERR_INVARIANT;
            }
        }
        max_row_len = std::max(max_row_len, e - s);
        n_words += k1 - k0 + 1;
        cells += e - s;
    }
    // candidate rows of variant p: [vlo[p], vhi[p]) with vhi[p] = number of rows with start <= p and vlo[p] = the
    // first row that covers p (vhi[p] if none). Rows are sorted by start, so both only move forward.
    {
        uint32_t j = 0, lo = 0;
        for (uint32_t p = 0; p < N; ++p) {
            while (j < idx.size() && v->read_start[idx[j]] <= p) ++j;
            while (lo < j && v->read_end[idx[lo]] <= p) ++lo;
            hpk.vhi[v0 + p] = j;
            hpk.vlo[v0 + p] = lo;
        }
    }
    uint32_t max_cov = 0;
    for (uint32_t p = 0; p < N; ++p) max_cov = std::max(max_cov, hpk.vhi[v0 + p] - hpk.vlo[v0 + p]);
    // quality mass behind the packed limits: qualities are bytes, so 255 x length bounds a row and 255 x cells the block;
    // the exact sums are only needed when those coarse bounds do not already clear the limits
    max_row_qual = 255ull * max_row_len;
    total_qual = 255ull * cells;
    if (max_row_qual * (uint64_t)std::max(max_cov, 1u) >= (1ull << 32) || total_qual >= hpk.qual_limit) {
        max_row_qual = 0; total_qual = 0;
        for (uint32_t r : idx) {
            const uint8_t* q = v->quals + v->row_off[r];
            uint64_t acc = 0;
            for (uint32_t c = 0; c < v->read_end[r] - v->read_start[r]; ++c) acc += q[c];
            max_row_qual = std::max(max_row_qual, acc);
            total_qual += acc;
        }
    }
    if (max_row_qual * (uint64_t)std::max(max_cov, 1u) >= (1ull << 32)) {
        set_error("row quality mass x coverage overflows the u32 score accumulators"); return HP_ERR_UNSUPPORTED;
    }
    if (total_qual >= hpk.qual_limit) { set_error("total quality mass of the block exceeds the packed key range (2^%d)", hpk.qual_limit == (1ull << 35) ? 35 : 29); return HP_ERR_UNSUPPORTED; }
    if (idx.size() >= (1u << 28)) { set_error("more than 2^28 rows in one block"); return HP_ERR_UNSUPPORTED; }
    d.max_cov = max_cov;
    d.n_words = (uint32_t)n_words;
    hpk.n_words += n_words;
    // per-position cell table (incremental scoring in the sub-solver); variants where two covering rows collide on
    // (row index mod 64) are flagged for the plane-word path. HP_NO_CTAB=1 switches the table off (A/B testing).
    const bool no_ctab = std::getenv("HP_NO_CTAB") != nullptr;
    if (!no_ctab && max_row_len <= CELL_T_MAX) {   // the cell table stores (p - row start) in 14 bits
        // entries per variant (hp_astar_dev.h CELL_*): 64, unless that leaves more than 2 % of the variants with two
        // covering rows on one entry (high coverage), then 128
        auto collisions = [&](uint32_t ents, bool mark) {
            uint32_t n_bad = 0;
            for (uint32_t p = 0; p < N; ++p) {
                const uint32_t lo = hpk.vlo[v0 + p], hi = hpk.vhi[v0 + p];
                if (hi - lo <= ents) continue;
                uint64_t seen[2] = {0, 0};
                for (uint32_t i = lo; i < hi; ++i) {
                    if (v->read_end[idx[i]] <= p) continue;
                    const uint32_t e = i & (ents - 1u);
                    const uint64_t bit = 1ull << (e & 63u);
                    if (seen[e >> 6] & bit) { n_bad += 1; if (mark) hpk.vflags[v0 + p] |= VAR_NOFAST; break; }
                    seen[e >> 6] |= bit;
                }
            }
            return n_bad;
        };
        uint32_t ents = 64;
        if (max_cov > 64) {   // (a handful of colliding variants in a small block is not worth the wider kernel variant)
            const uint32_t n_bad = collisions(64, false);
            if ((uint64_t)n_bad * 50 > N && n_bad >= 32) ents = 128;
        }
        if (max_cov > ents) collisions(ents, true);
        d.ctab_shift = ents == 128 ? 7u : 6u;
        d.cell_off = hpk.cell_total;
        hpk.cell_total += (uint64_t)N * ents;
    } else {
        d.cell_off = ~0ull;
        d.ctab_shift = 6;
    }
    hpk.desc.push_back(d);
    hpk.work.push_back(cells * 8 + N);
    hpk.max_n = std::max(hpk.max_n, N);
    return HP_OK;
}

inline double wall_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

template <class T> int upload(DevBuf& buf, const std::vector<T>& v, hipStream_t s) {
    int rc = buf.alloc(v.size() * sizeof(T));
    if (rc != HP_OK) return rc;
    return v.empty() ? HP_OK : dev_put(buf.p, v.data(), v.size() * sizeof(T), s);   // (hp_common.h: not the copy engines, which are busy with the next set's reads; the caller ends with dev_io_sync on s, or on a stream that waits for s)
}

}  // namespace

}  // namespace hp

using namespace hp;

// Streams and events of a batch come from a per-thread pool (creating two streams and four events costs milliseconds,
// which matters when a caller solves one small block per call).
struct StreamSet {
    int device = -1, partition = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr;
    bool create(int dev) {
        device = dev; partition = g_cu_partition;
        // the segment stream carries the critical path: a high-priority stream also gets a hardware queue of its own
        // (streams of equal priority may share one when the host process has created many, e.g. under PyTorch)
        return hp_stream_create(&stream, dev, 1) == hipSuccess && hp_stream_create(&stream2, dev, 1) == hipSuccess &&
               hipEventCreate(&ev0) == hipSuccess && hipEventCreate(&ev1) == hipSuccess &&
               hipEventCreate(&ev_fork) == hipSuccess && hipEventCreate(&ev_join) == hipSuccess;
    }
    void destroy() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
        *this = StreamSet{};
    }
};
// process-wide (a batch may be created on one thread and solved / destroyed on another: the stages of a block stream); never
// destroyed
struct StreamPool {
    std::mutex m;
    std::vector<StreamSet> free_;
    bool get(int dev, StreamSet& out) {
        {
            std::lock_guard<std::mutex> lk(m);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].device == dev && free_[i].partition == g_cu_partition) { out = free_[i]; free_.erase(free_.begin() + i); return true; }
        }
        return out.create(dev);
    }
    void put(StreamSet& s) {
        {
            std::lock_guard<std::mutex> lk(m);
            if (free_.size() < 32) { free_.push_back(s); s = StreamSet{}; return; }
        }
        s.destroy();
    }
};
static StreamPool& stream_pool() { static StreamPool* p = new StreamPool(); return *p; }
#define g_stream_pool (stream_pool())

struct hp_batch {
    int device = 0;
    int partition = 0;   // CU partition its streams were created in (hp_common.h)
    hipStream_t stream = nullptr;
    size_t n_blocks = 0;
    hp_astar_params params{};
    SolveParams prm{};
    std::vector<BlockDesc> desc;
    std::vector<uint32_t> order;  // LPT
    uint64_t sum_n = 0, sum_h = 0;
    uint32_t max_n = 0;
    uint32_t tiles = 1;     // 2 when some block's cell table has 128 entries per variant (selects the TILES kernel variant)
    int n_cu = 256;
    // device inputs
    DevBuf d_desc, d_order, d_vlo, d_vhi, d_vflags, d_rstart, d_rend, d_rword, d_words, d_ctab;
    // device outputs
    DevBuf d_H, d_h1, d_h2, d_stats, d_counters, d_status, d_hapw;
    // post-processing (phaser.rs:350-388, :714-750)
    DevBuf d_row_block, d_haplotag, d_first_het, d_js, d_je, d_junc_block, d_junc_off, d_span;
    std::vector<uint32_t> row_orig, row_block_h;
    std::vector<uint64_t> caller_row_off;
    // host tables that are uploaded with hipMemcpyAsync: they live as long as the batch, so no copy can outlive its source
    std::vector<SegDesc> h_segs;
    std::vector<uint32_t> h_seg_order, h_sb_first, h_sb_n, h_sb_id, h_junc_block, h_blk_seg;   // h_blk_seg: per block, [first segment | how many]
    std::vector<uint64_t> h_junc_off;
    uint64_t caller_rows = 0, n_rows_packed = 0, n_junctures = 0;
    bool solved = false;
    // segment-parallel heuristic (large blocks on an otherwise idle GPU)
    DevBuf d_segs, d_seg_order, d_seg_out, d_seg_off, d_seg_retry, d_sb_first, d_sb_n, d_sb_id, s_seg_pool, d_blk_seg;
    uint32_t last_n_segs = 0;
    // second stream + scratch: segmented blocks run beside the sequential pass of all the others
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    DevBuf g_main_pool, g_main_heap, g_sub_pool, g_sub_heap, g_tracker, d_order2, d_order1;
    uint32_t g_slots = 0, g_cap = 0;
    // scratch (sized on first solve, kept)
    DevBuf s_sub_pool, s_main_pool, s_sub_heap, s_main_heap, s_tracker;
    uint32_t scratch_slots = 0, scratch_cap_main = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    ~hp_batch() {
        (void)hp_set_device(device);
        // the device buffers go back to the per-thread cache (hp_common.h), not to hipFree: nothing may still use them
        if (stream2) (void)hipStreamSynchronize(stream2);
        if (stream) (void)hipStreamSynchronize(stream);
        StreamSet ss;
        ss.device = device; ss.partition = partition; ss.stream = stream; ss.stream2 = stream2; ss.ev0 = ev0; ss.ev1 = ev1; ss.ev_fork = ev_fork; ss.ev_join = ev_join;
        if (stream && stream2 && ev0 && ev1 && ev_fork && ev_join) g_stream_pool.put(ss); else ss.destroy();
    }
};

namespace {

constexpr uint32_t LDS_SUB_HEAP_MAX_BYTES = 24 * 1024;  // keep >= 6 waves per CU resident (160 KiB LDS)

// Node capacity of a launch's FIRST main search. 6 N + 2048 covers clean data (<= 4 N + 1 nodes); wrong-haplotype cells make the
// search jump and its frontier grow (15 % of them at 60x: 25-30 nodes a variant), and an attempt that runs out is run again from
// its start. Scratch that is never touched costs nothing but address space, so a launch gets up to 32 N + 2048 while its slots
// together stay within 2 GB (104 bytes a node: family record, heap key, a quarter of a chunk record). HP_ASTAR_CAP0=f (test hook):
// f N + 64, to reach the retries.
uint32_t first_pass_cap(uint32_t max_n, size_t n_items, uint32_t n_cu) {
    if (const char* e = std::getenv("HP_ASTAR_CAP0")) return (uint32_t)std::max(1, std::atoi(e)) * max_n + 64;
    const uint64_t lo = 6ull * max_n + 2048, hi = 32ull * max_n + 2048;
    const uint64_t slots = std::max<uint64_t>(1, std::min<uint64_t>(n_items, (uint64_t)n_cu * 24));
    const uint64_t by_mem = (2ull << 30) / slots / 104;
    return (uint32_t)std::max(lo, std::min(hi, by_mem));
}

// Plans and launches the segment-parallel heuristic for blocks that would otherwise be the critical path
// (heavy-tailed block sizes, or few blocks): see hp_astar_dev.h. Blocks whose seams verify get status ST_H_READY
// and their final H[]; everything else is left for the sequential path of hp_astar_kernel.
int launch_segments(hp_batch* b, hipStream_t st, std::vector<uint32_t>& seg_blocks) {
    b->last_n_segs = 0;
    seg_blocks.clear();
    if (std::getenv("HP_NO_SEGMENTS") || !b->prm.sub_heap_in_lds || b->prm.max_seg > SEG_STATE) return HP_OK;
    // HP_SEG_CRING=1: the sub-solver's window of the cell table staged in LDS (hp_astar_kernel.hip, CR) instead of read where it lies
    // (launches whose blocks need two tiles per variant - coverage 45 and up - never stage: their ring would be 32 KB). Read per call:
    // the parity tests switch it inside one process. OFF by default, measured (round 6, MI355X, three runs a side): one C2 block 505.2
    // against 506.5 ms - the rows it replaces were L1 / L2 hits, the chain is bound by its instructions - and inside a block stream the
    // A* kernels' chain 20.0-21.3 against 17.6-19.4 ms per set: a resident graph-WFA launch set holds 154 of a compute unit's 160 KB of
    // LDS, a 19 KB segment workgroup waits for one of its workgroups to retire where a 2.8 KB one moves in beside them.
    const bool cring_env = [] { const char* e = std::getenv("HP_SEG_CRING"); return e && e[0] == '1'; }();
    const bool cring = cring_env && b->tiles != 2 && b->d_ctab.p != nullptr;
    const size_t heap_bytes = LDS_HEAP_OFF + (size_t)b->prm.jcap_sub * 64 * sizeof(uint64_t);
    const size_t lds_bytes = heap_bytes + (cring ? (size_t)64 * 64 * 4 : 0);
    const int occ = 6;
    const uint32_t per_cu = (uint32_t)std::min<size_t>(4 * occ, (160 * 1024) / lds_bytes);
    const uint64_t max_slots = (uint64_t)b->n_cu * std::max(per_cu, 1u);
    uint64_t total = 0;
    for (auto& d : b->desc) total += d.n_vars;
    const char* tenv = std::getenv("HP_SEG_TARGET");
    // segments of 32 owned variants (+ the warm-up) unless that makes more segments than resident slots: up to two
    // single-wave workgroups per SIMD run at the speed of one (profiles/round2/issue_ceiling.txt), so the shorter chain is free
    uint64_t target = tenv ? (uint64_t)std::atoll(tenv) : std::max<uint64_t>(32, total / max_slots);
    target = std::max<uint64_t>(32, (target + 31) / 32 * 32);
    // Two rounds: a short warm-up first (48 variants; 64 closed every seam of the synthetic mixes; the chain forgets its start
    // after a few dozen variants), then only the segments below a seam that stayed open are solved again with the long
    // one (160: scripts/spec_converge.py found 120 sufficient at 1 % and 15 % error). The seam check decides, so both
    // lengths only matter for speed. HP_SEG_WARM / HP_SEG_WARM2 override them; WARM >= WARM2 means one round.
    const char* wenv = std::getenv("HP_SEG_WARM");
    const char* wenv2 = std::getenv("HP_SEG_WARM2");
    const uint32_t warm = wenv ? (uint32_t)std::atoi(wenv) : 48;   // (round 5: 48 closes nearly every seam of the bench's mix - A* kernels 17.3 against 19.0 ms per set in the stream with 64; at 40 and 32 so many segments are solved again with the long warm-up that they take 30)
    const uint32_t warm2 = wenv2 ? (uint32_t)std::atoi(wenv2) : 160;
    const bool two_rounds = warm < warm2;
    std::vector<SegDesc>& segs = b->h_segs;
    std::vector<uint32_t>&sb_first = b->h_sb_first, &sb_n = b->h_sb_n, &sb_id = b->h_sb_id;
    (void)hipStreamSynchronize(st);   // a previous solve's uploads from these tables are long done; make it certain before they change
    segs.clear(); sb_first.clear(); sb_n.clear(); sb_id.clear();
    for (uint32_t i = 0; i < b->desc.size(); ++i) {
        const uint32_t N = b->desc[i].n_vars;
        if ((uint64_t)N < 2 * target) continue;
        const uint32_t ns = (uint32_t)(N / target);   // last (top) segment takes the remainder: length in [target, 2*target)
        sb_first.push_back((uint32_t)segs.size());
        sb_n.push_back(ns);
        sb_id.push_back(i);
        for (uint32_t k = 0; k < ns; ++k) {
            SegDesc sd;
            sd.blk = i;
            sd.a = (uint32_t)(k * target);
            sd.b = (k + 1 == ns) ? N : (uint32_t)((k + 1) * target);
            sd.v0 = (k + 1 == ns) ? N : std::min<uint32_t>(N, sd.b + std::max(warm, 1u));
            segs.push_back(sd);
        }
    }
    if (segs.empty()) return HP_OK;
    std::vector<uint32_t>& order = b->h_seg_order;
    order.resize(segs.size());
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return segs[x].v0 - segs[x].a > segs[y].v0 - segs[y].a; });
    const uint32_t slots = (uint32_t)std::min<uint64_t>(segs.size(), max_slots);
    SolveParams prm = b->prm;
    prm.seg_profile = 0;
    const size_t sub_pool_bytes = sub_pool_bytes_per_slot(prm);
    int rc;
    if (b->s_seg_pool.bytes < (size_t)slots * sub_pool_bytes && (rc = b->s_seg_pool.alloc((size_t)slots * sub_pool_bytes)) != HP_OK) return rc;
    // per block (what hp_astar_kernel looks up for a block whose seams stayed open): first segment, then the counts
    std::vector<uint32_t>& blk_seg = b->h_blk_seg;
    blk_seg.assign(2 * (size_t)b->n_blocks, 0u);
    for (size_t t = 0; t < sb_id.size(); ++t) { blk_seg[sb_id[t]] = sb_first[t]; blk_seg[b->n_blocks + sb_id[t]] = sb_n[t]; }
    if ((rc = upload(b->d_blk_seg, blk_seg, st)) != HP_OK) return rc;
    if ((rc = upload(b->d_segs, segs, st)) || (rc = upload(b->d_seg_order, order, st)) || (rc = upload(b->d_sb_first, sb_first, st)) ||
        (rc = upload(b->d_sb_n, sb_n, st)) || (rc = upload(b->d_sb_id, sb_id, st)))
        return rc;
    if ((rc = b->d_seg_out.alloc(segs.size() * sizeof(SegOut))) || (rc = b->d_seg_off.alloc(segs.size() * 8)) || (rc = b->d_seg_retry.alloc(segs.size()))) return rc;
    HP_HIP_CHECK(hipMemsetAsync(b->d_seg_off.p, 0, segs.size() * 8, st));
    HP_HIP_CHECK(hipMemsetAsync(b->d_seg_retry.p, 0, segs.size(), st));
    SegBatchDev S{};
    BatchDev& B = S.B;
    B.desc = b->d_desc.as<BlockDesc>();
    B.vlo = b->d_vlo.as<uint32_t>(); B.vhi = b->d_vhi.as<uint32_t>(); B.vflags = b->d_vflags.as<uint8_t>();
    B.rstart = b->d_rstart.as<uint32_t>(); B.rend = b->d_rend.as<uint32_t>(); B.rword = b->d_rword.as<uint32_t>();
    B.words = b->d_words.as<uint32_t>();
    B.ctab = b->d_ctab.as<uint32_t>();
    B.H = b->d_H.as<uint64_t>();
    B.sub_pool = b->s_seg_pool.as<unsigned char>();
    B.prm = prm;
    S.segs = b->d_segs.as<SegDesc>(); S.seg_order = b->d_seg_order.as<uint32_t>(); S.n_segs = (uint32_t)segs.size();
    S.out = b->d_seg_out.as<SegOut>();
    S.run_flag = nullptr; S.warm = 0; S.cring_off = (uint32_t)heap_bytes;
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] segment-parallel heuristic: %zu segments of ~%llu hets (+%u warm-up, open seams again with +%u) over %zu blocks, slots=%u\n", segs.size(), (unsigned long long)target, warm, two_rounds ? warm2 : warm, sb_id.size(), slots); fflush(stderr); }
    StitchDev T{};
    T.desc = B.desc; T.segs = S.segs; T.out = S.out;
    T.blk_first_seg = b->d_sb_first.as<uint32_t>(); T.blk_n_seg = b->d_sb_n.as<uint32_t>(); T.blk_id = b->d_sb_id.as<uint32_t>();
    T.n_seg_blocks = (uint32_t)sb_id.size(); T.H = B.H; T.seg_offset = b->d_seg_off.as<uint64_t>();
    T.status = b->d_status.as<int32_t>(); T.counters = b->d_counters.as<hp_work_counters>();
    T.retry = b->d_seg_retry.as<uint8_t>();
    for (int round = 0; round < (two_rounds ? 2 : 1); ++round) {
        if (round == 1) { S.run_flag = T.retry; S.warm = warm2; }
        T.final_round = (round == 1 || !two_rounds) ? 1u : 0u;
        if (b->tiles == 2) hipLaunchKernelGGL((hp_heur_seg_kernel<true, 6, 2, false>), dim3(slots), dim3(64), lds_bytes, st, S);
        else if (cring) hipLaunchKernelGGL((hp_heur_seg_kernel<true, 6, 1, true>), dim3(slots), dim3(64), lds_bytes, st, S);
        else hipLaunchKernelGGL((hp_heur_seg_kernel<true, 6, 1, false>), dim3(slots), dim3(64), lds_bytes, st, S);
        hipLaunchKernelGGL(hp_heur_stitch_kernel, dim3(T.n_seg_blocks), dim3(64), 0, st, T);
    }
    ApplyDev A{};
    A.segs = S.segs; A.seg_offset = T.seg_offset; A.desc = B.desc; A.n_segs = S.n_segs; A.H = B.H;
    hipLaunchKernelGGL(hp_heur_apply_kernel, dim3(S.n_segs), dim3(64), 0, st, A);   // (single-wavefront workgroups, here and below: beside a graph-WFA launch set a CU has one wavefront slot free, rarely four - a 256-thread workgroup of the row kernel waited 18 ms for one)
    HP_HIP_CHECK(hipGetLastError());
    if (std::getenv("HP_DEBUG")) {   // how the seams went (costs a wait: debug only)
        std::vector<int32_t> stt(b->n_blocks);
        std::vector<uint8_t> rf(segs.size());
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(stt.data(), b->d_status.p, stt.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(rf.data(), b->d_seg_retry.p, rf.size(), hipMemcpyDeviceToHost);
        size_t ready = 0, again = 0;
        for (uint32_t i : sb_id) ready += stt[i] == ST_H_READY;
        for (uint8_t f : rf) again += f;
        fprintf(stderr, "[hp] segment-parallel heuristic: %zu of %zu blocks stitched, %zu segments solved again with the long warm-up\n", ready, sb_id.size(), again);
        fflush(stderr);
    }
    b->last_n_segs = S.n_segs;
    seg_blocks = sb_id;
    return HP_OK;
}

int launch_pass(hp_batch* b, hipStream_t st, const std::vector<uint32_t>& items, uint32_t cap_main,
                DevBuf& main_pool, DevBuf& main_heap, DevBuf& sub_pool, DevBuf& sub_heap, DevBuf& tracker,
                uint32_t& have_slots, uint32_t& have_cap, DevBuf& d_items, bool seg_pass = false) {
    SolveParams prm = b->prm;
    prm.cap_main = cap_main;
    prm.jcap_main = (cap_main + 63) / 64 + 1;
    uint32_t max_n = 0;
    for (uint32_t i : items) max_n = std::max(max_n, b->desc[i].n_vars);
    prm.max_n_vars = max_n;
    prm.seg_profile = std::getenv("HP_SEG_PROFILE") ? 1u : 0u;   // per-segment s_memtime profile of the sub-solver loop
    const bool verbose = std::getenv("HP_DEBUG") != nullptr;
    // (at least 1 KB behind the rings: the main search keeps its heaps' roots where the sub-solver's heap lay)
    const size_t lds_bytes = LDS_HEAP_OFF + std::max<size_t>(1024, prm.sub_heap_in_lds ? (size_t)prm.jcap_sub * 64 * sizeof(uint64_t) : 0);
    // resident waves per CU limited by LDS; one wave per workgroup
    const int occ = prm.sub_heap_in_lds ? 6 : 4;
    uint32_t per_cu = (uint32_t)std::min<size_t>(4 * occ, (160 * 1024) / std::max<size_t>(lds_bytes, 1));
    if (per_cu == 0) per_cu = 1;
    prm.cap_chunk_main = cap_main / 4 + 64;
    const size_t main_pool_bytes = (size_t)cap_main * sizeof(FamRec) + (size_t)prm.cap_chunk_main * sizeof(ChunkRec);
    const size_t sub_pool_bytes = sub_pool_bytes_per_slot(prm);
    const size_t per_slot = main_pool_bytes + (size_t)prm.jcap_main * 64 * sizeof(Key) + sub_pool_bytes + ((size_t)max_n + 1) * 4 +
                            (prm.sub_heap_in_lds ? 0 : (size_t)prm.jcap_sub * 64 * sizeof(uint64_t));
    size_t free_b = 0, total_b = 0;
    HP_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    size_t budget = free_b / 2 + main_pool.bytes + main_heap.bytes + sub_pool.bytes + tracker.bytes;
    uint32_t slots = (uint32_t)std::min<size_t>({(size_t)items.size(), (size_t)b->n_cu * per_cu, std::max<size_t>(1, budget / per_slot)});
    if (slots == 0) slots = 1;
    if (per_slot > budget) { set_error("a single block needs %zu bytes of solver scratch; only %zu available", per_slot, budget); return HP_ERR_OOM; }
    if (have_slots < slots || have_cap != cap_main || tracker.bytes < (size_t)slots * ((size_t)max_n + 1) * 4) {
        int rc;
        if ((rc = main_pool.alloc((size_t)slots * main_pool_bytes)) != HP_OK) return rc;
        if ((rc = main_heap.alloc((size_t)slots * prm.jcap_main * 64 * sizeof(Key))) != HP_OK) return rc;
        if ((rc = sub_pool.alloc((size_t)slots * sub_pool_bytes)) != HP_OK) return rc;
        if ((rc = tracker.alloc((size_t)slots * ((size_t)max_n + 1) * 4)) != HP_OK) return rc;
        if (!prm.sub_heap_in_lds && (rc = sub_heap.alloc((size_t)slots * prm.jcap_sub * 64 * sizeof(uint64_t))) != HP_OK) return rc;
        have_slots = slots;
        have_cap = cap_main;
    }
    slots = std::min(slots, have_slots);
    int rc = upload(d_items, items, st);
    if (rc != HP_OK) return rc;

    BatchDev B{};
    B.desc = b->d_desc.as<BlockDesc>();
    B.order = d_items.as<uint32_t>();
    B.n_items = (uint32_t)items.size();
    B.vlo = b->d_vlo.as<uint32_t>(); B.vhi = b->d_vhi.as<uint32_t>(); B.vflags = b->d_vflags.as<uint8_t>();
    B.rstart = b->d_rstart.as<uint32_t>(); B.rend = b->d_rend.as<uint32_t>(); B.rword = b->d_rword.as<uint32_t>();
    B.words = b->d_words.as<uint32_t>();
    B.ctab = b->d_ctab.as<uint32_t>();
    B.H = b->d_H.as<uint64_t>(); B.h1 = b->d_h1.as<uint8_t>(); B.h2 = b->d_h2.as<uint8_t>(); B.hapw = b->d_hapw.as<Win>();
    B.stats = b->d_stats.as<hp_phase_stats>(); B.counters = b->d_counters.as<hp_work_counters>();
    B.status = b->d_status.as<int32_t>();
    B.sub_pool = sub_pool.as<unsigned char>(); B.main_pool = main_pool.as<unsigned char>();
    B.sub_heap_g = sub_heap.as<uint64_t>(); B.main_heap = main_heap.as<Key>(); B.tracker = tracker.as<uint32_t>();
    // (only the launch BEHIND the segment kernels, on their stream: the tables are uploaded on that stream, and no other launch has a
    // block with segments - the ordinary pass on the other stream read them before they had arrived, found by HP_DEV_CACHE_POISON=1.
    // HP_SEG_NO_TAKEOVER=1: an unaccepted block walks its whole chain, as until round 6 - A/B switch)
    if (seg_pass && b->last_n_segs && !std::getenv("HP_SEG_NO_TAKEOVER")) {
        B.segs = b->d_segs.as<SegDesc>(); B.seg_out = b->d_seg_out.as<SegOut>();
        B.blk_seg_first = b->d_blk_seg.as<uint32_t>(); B.blk_seg_n = b->d_blk_seg.as<uint32_t>() + b->n_blocks;
    }
    B.prm = prm;
    if (verbose) {
        int occ_blocks = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_blocks, hp_astar_kernel<true, 6, false, 1>, 64, lds_bytes);
        fprintf(stderr, "[hp] hipOccupancyMaxActiveBlocksPerMultiprocessor(hp_astar_kernel, 64, %zu) = %d\n", lds_bytes, occ_blocks);
    }
    if (verbose) { fprintf(stderr, "[hp] launch items=%zu slots=%u cap_main=%u cap_sub=%u jcap_sub=%u lds=%zu\n", items.size(), slots, cap_main, prm.cap_sub, prm.jcap_sub, lds_bytes); fflush(stderr); }
    if (!prm.sub_heap_in_lds) hipLaunchKernelGGL((hp_astar_kernel<false, 4, false, 2>), dim3(slots), dim3(64), lds_bytes, st, B);
    else if (b->tiles == 2) hipLaunchKernelGGL((hp_astar_kernel<true, 6, false, 2>), dim3(slots), dim3(64), lds_bytes, st, B);
    else if (prm.seg_profile) hipLaunchKernelGGL((hp_astar_kernel<true, 6, true, 1>), dim3(slots), dim3(64), lds_bytes, st, B);
    else hipLaunchKernelGGL((hp_astar_kernel<true, 6, false, 1>), dim3(slots), dim3(64), lds_bytes, st, B);
    HP_HIP_CHECK(hipGetLastError());
    return HP_OK;
}

}  // namespace

extern "C" {

hp_batch* hp_batch_create(size_t n_blocks, const hp_block_view* blks, const hp_astar_params* p, int device_id, int* status) {
    bool device_touched = false;
    // on a failure after work was queued, wait for it: the buffers go back to the per-thread cache on return
    auto fail = [&](int code) { if (device_touched) (void)hipDeviceSynchronize(); if (status) *status = code; return (hp_batch*)nullptr; };
    if (!blks || !p || n_blocks == 0 || n_blocks > 0x7FFFFFFFull) { set_error("bad arguments to hp_batch_create"); return fail(HP_ERR_ARG); }
    const uint64_t max_seg = p->max_segment_size ? p->max_segment_size : 40;
    if (max_seg < 2 || max_seg > 62) { set_error("max_segment_size %llu outside [2,62]", (unsigned long long)max_seg); return fail(HP_ERR_UNSUPPORTED); }
    if (p->min_queue_size > (1u << 26) || p->queue_increment > (1u << 20)) { set_error("queue parameters too large"); return fail(HP_ERR_UNSUPPORTED); }

    // Validation + packing is independent per block: host threads (HP_PACK_THREADS, default min(8, cores)) each
    // pack a contiguous range of blocks into their own arrays; the parts are uploaded side by side and only the
    // small per-block tables are merged.
    const double t_pack0 = wall_ms();
    // (8: measured 1.9 ms for the default bench's 452 blocks; 4: 2.0, 16: 2.9, 32: 4.8 - every part costs nine uploads)
    unsigned nt = host_threads(8u);
    if (const char* e = std::getenv("HP_PACK_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
    nt = (unsigned)std::min<size_t>(nt, n_blocks / 8 + 1);
    std::vector<HostPack> parts(nt);
    // the sub-solver's key: 14 index bits and 36 cost bits, or - for queue sizes whose sub-problems may visit more than 4 095
    // nodes (--phase-min-queue-size above 39 730 at the default increment) - 20 index bits and 30 cost bits
    const bool wide_sub_keys = 4ull * ((uint64_t)(p->min_queue_size / 10) + (uint64_t)p->queue_increment * max_seg) + 1ull >= (1ull << 14);
    if (wide_sub_keys) for (auto& part : parts) part.qual_limit = 1ull << 29;
    // contiguous ranges of blocks with about the same number of cells each (block sizes are heavy-tailed: equal COUNTS left
    // one thread with the 2 000-variant block and its neighbours)
    std::vector<size_t> cut(nt + 1, n_blocks);
    {
        std::vector<uint64_t> cum(n_blocks + 1, 0);
        for (size_t i = 0; i < n_blocks; ++i) {
            const hp_block_view& v = blks[i];
            const uint64_t c = (v.row_off && v.n_reads && v.n_reads <= (1u << 28)) ? v.row_off[v.n_reads] : 0;
            cum[i + 1] = cum[i] + std::min<uint64_t>(c, 1ull << 40) + 64u * v.n_reads + 256u;
        }
        cut[0] = 0;
        for (unsigned t = 1; t < nt; ++t)
            cut[t] = std::max<size_t>(cut[t - 1], (size_t)(std::lower_bound(cum.begin(), cum.end(), cum[n_blocks] / nt * t) - cum.begin()));
        for (unsigned t = 1; t < nt; ++t) cut[t] = std::min(cut[t], n_blocks);
    }
    {
        std::vector<int> rcs(nt, HP_OK);
        std::vector<std::string> errs(nt);
        auto work = [&](unsigned t) {
            {   // exact capacities up front: no reallocation copies while packing (rows with an empty region are dropped,
                // so the row tables may end up a little shorter; malformed views are rejected by pack_block)
                uint64_t rows = 0, cells = 0, vars = 0, abytes = 0;
                for (size_t i = cut[t]; i < cut[t + 1]; ++i) {
                    const hp_block_view& v = blks[i];
                    if (!v.row_off || v.n_reads > (1u << 28) || v.n_variants > (1u << 24)) { rows = 0; cells = 0; vars = 0; abytes = 0; break; }
                    const uint64_t c = v.n_reads ? v.row_off[v.n_reads] : 0;
                    if (c > (1ull << 40)) { rows = 0; cells = 0; vars = 0; abytes = 0; break; }
                    rows += v.n_reads; cells += c; vars += v.n_variants; abytes += (c + 3) / 4;
                }
                HostPack& q = parts[t];
                try {
                    q.vlo.reserve(vars); q.vhi.reserve(vars); q.vflags.reserve(vars);
                    q.rstart.reserve(rows); q.rend.reserve(rows); q.rword.reserve(rows); q.rcell.reserve(rows);
                    q.row_block.reserve(rows); q.row_orig.reserve(rows);
                    q.raw_alleles.reserve(abytes); q.raw_quals.reserve(cells);
                } catch (...) {}   // a hint only: a view with absurd sizes is reported by pack_block, not here
            }
            for (size_t i = cut[t]; i < cut[t + 1]; ++i) {
                const int rc = pack_block(&blks[i], parts[t]);
                if (rc != HP_OK) {
                    rcs[t] = rc;
                    errs[t] = "block " + std::to_string(i) + ": " + hp_last_error();
                    return;
                }
            }
        };
        WorkerPool::get().run(nt, work);
        for (unsigned t = 0; t < nt; ++t) if (rcs[t] != HP_OK) { set_error("%s", errs[t].c_str()); return fail(rcs[t]); }
    }
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] host pack of %zu blocks on %u threads: %.1f ms\n", n_blocks, nt, wall_ms() - t_pack0); fflush(stderr); }
    HostPack hpk;   // merged view: per-block tables and totals only (the large arrays stay in `parts`)
    struct PartBase { uint64_t var, read, word, h, chunk, cell, rows, ral, rq; uint32_t blk; };
    std::vector<PartBase> base(nt);
    {
        PartBase acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (unsigned t = 0; t < nt; ++t) {
            HostPack& q = parts[t];
            base[t] = acc;
            for (BlockDesc d : q.desc) {
                d.var_off += acc.var; d.read_off += acc.read; d.word_off += acc.word; d.h_off += acc.h; d.chunk_off += acc.chunk;
                if (d.cell_off != ~0ull) d.cell_off += acc.cell;
                hpk.desc.push_back(d);
            }
            for (const PackRaw& pr : q.raw) hpk.raw.push_back(PackRaw{pr.allele_off + acc.ral, pr.qual_off + acc.rq});
            for (uint64_t o : q.caller_row_off) hpk.caller_row_off.push_back(o + acc.rows);
            hpk.work.insert(hpk.work.end(), q.work.begin(), q.work.end());
            hpk.max_n = std::max(hpk.max_n, q.max_n);
            acc.var += q.vlo.size(); acc.read += q.rstart.size(); acc.word += q.n_words;
            acc.ral += q.raw_alleles.size(); acc.rq += q.raw_quals.size();
            acc.h += q.h_total; acc.chunk += q.chunk_total; acc.cell += q.cell_total; acc.rows += q.caller_rows;
            acc.blk += (uint32_t)q.desc.size();
        }
        hpk.h_total = acc.h; hpk.chunk_total = acc.chunk; hpk.cell_total = acc.cell; hpk.caller_rows = acc.rows;
        // the two per-row tables the batch keeps on the host (tens of millions of rows): merged by the same threads
        hpk.row_block.resize(acc.read);
        hpk.row_orig.resize(acc.read);
        auto merge_rows = [&](unsigned t) {
            const HostPack& q = parts[t];
            uint32_t* rb = hpk.row_block.data() + base[t].read;
            for (size_t i = 0; i < q.row_block.size(); ++i) rb[i] = q.row_block[i] + base[t].blk;
            if (!q.row_orig.empty()) std::memcpy(hpk.row_orig.data() + base[t].read, q.row_orig.data(), q.row_orig.size() * sizeof(q.row_orig[0]));
        };
        if (nt == 1) merge_rows(0);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(merge_rows, t);
            for (auto& x : th) x.join();
        }
    }
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp]   merged the per-thread parts at %.1f ms\n", wall_ms() - t_pack0); fflush(stderr); }
    const uint64_t tot_vars = base[nt - 1].var + parts[nt - 1].vlo.size(), tot_rows = base[nt - 1].read + parts[nt - 1].rstart.size();
    const uint64_t tot_words = base[nt - 1].word + parts[nt - 1].n_words;
    const uint64_t tot_ral = base[nt - 1].ral + parts[nt - 1].raw_alleles.size(), tot_rq = base[nt - 1].rq + parts[nt - 1].raw_quals.size();
    if (tot_words > 0xFFFFFFFFFFull) { set_error("batch too large"); return fail(HP_ERR_UNSUPPORTED); }
    // host-side validation/packing is done; from here on a GPU is mandatory
    if (device_id < 0) device_id = hp_default_device();
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed - no usable GPU; there is no CPU fallback", device_id); return fail(HP_ERR_HIP); }
    device_touched = true;
    std::unique_ptr<hp_batch> b(new hp_batch());
    b->device = device_id;
    b->n_blocks = n_blocks;
    b->params = *p;
    b->desc = hpk.desc;
    b->max_n = hpk.max_n;
    for (const BlockDesc& d : hpk.desc) if (d.ctab_shift == 7) b->tiles = 2;
    b->sum_h = hpk.h_total;
    b->sum_n = tot_vars;
    b->row_orig = std::move(hpk.row_orig);
    b->row_block_h = std::move(hpk.row_block);
    b->caller_row_off = std::move(hpk.caller_row_off);
    b->caller_rows = hpk.caller_rows;
    b->n_rows_packed = b->row_block_h.size();
    {   // the post-processing kernels index rows and junctures of the whole batch with 32 bits
        uint64_t nj = 0;
        for (size_t i = 0; i < n_blocks; ++i) nj += blks[i].n_variants ? blks[i].n_variants - 1 : 0;
        if (b->n_rows_packed >= 0xFFFFFFF0ull || nj >= 0xFFFFFFF0ull) { set_error("batch with %llu rows / %llu junctures exceeds 2^32: split it", (unsigned long long)b->n_rows_packed, (unsigned long long)nj); return fail(HP_ERR_UNSUPPORTED); }
    }
    b->n_cu = partition_cu_count(device_id);
    {
        StreamSet ss;
        const bool ok = g_stream_pool.get(device_id, ss);
        b->partition = ss.partition; b->stream = ss.stream; b->stream2 = ss.stream2; b->ev0 = ss.ev0; b->ev1 = ss.ev1; b->ev_fork = ss.ev_fork; b->ev_join = ss.ev_join;
        if (!ok) { set_error("creating the streams/events of a batch failed"); return fail(HP_ERR_HIP); }
    }

    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp]   batch object filled at %.1f ms\n", wall_ms() - t_pack0); fflush(stderr); }
    SolveParams& prm = b->prm;
    prm.minq_main = (uint32_t)p->min_queue_size;
    prm.minq_sub = (uint32_t)(p->min_queue_size / 10);
    prm.qinc = (uint32_t)p->queue_increment;
    prm.max_seg = (uint32_t)max_seg;
    const uint64_t max_visits = (uint64_t)prm.minq_sub + (uint64_t)prm.qinc * max_seg;
    prm.cap_sub = (uint32_t)(4 * max_visits + 1);   // root + at most 4 children per visit
    prm.jcap_sub = (uint32_t)((max_visits + 63) / 64);   // one key per family; <= 1 key per visit is dealt round-robin to the lanes (SubHeap::deal)
    prm.cap_chunk_sub = (uint32_t)max_visits + 8;   // at most one ChunkRec per expansion
    prm.sub_heap_in_lds = ((size_t)prm.jcap_sub * 64 * sizeof(uint64_t) <= LDS_SUB_HEAP_MAX_BYTES) ? 1 : 0;
    prm.save_state = (b->tiles == 2) ? 1u : 0u;   // must match the TILES template argument of the launches (see subsolve)
    prm.sub_idx_bits = wide_sub_keys ? 20u : 14u;
    if (4ull * max_visits + 1ull >= (1ull << 20)) { set_error("min_queue_size/10 + queue_increment*max_segment_size = %llu visits exceeds the packed sub-key limit (262 143)", (unsigned long long)max_visits); return fail(HP_ERR_UNSUPPORTED); }

    b->order.resize(n_blocks);
    std::iota(b->order.begin(), b->order.end(), 0u);
    std::stable_sort(b->order.begin(), b->order.end(), [&](uint32_t a, uint32_t c) { return hpk.work[a] > hpk.work[c]; });

    hipStream_t s = b->stream;
    int rc;
    if ((rc = upload(b->d_desc, hpk.desc, s)) != HP_OK) return fail(rc);
    if ((rc = b->d_vlo.alloc(tot_vars * 4)) || (rc = b->d_vhi.alloc(tot_vars * 4)) || (rc = b->d_vflags.alloc(tot_vars)) ||
        (rc = b->d_rstart.alloc(tot_rows * 4)) || (rc = b->d_rend.alloc(tot_rows * 4)) || (rc = b->d_rword.alloc(tot_rows * 4)) ||
        (rc = b->d_words.alloc(tot_words * WORD_DWORDS * 4)))
        return fail(rc);
    DevBuf d_rcell, d_ral, d_rq, d_raw;   // inputs of the device-side packing only: released when this function returns
    if ((rc = d_rcell.alloc(tot_rows * 8)) || (rc = d_ral.alloc(tot_ral + 16)) || (rc = d_rq.alloc(tot_rq + 16)) ||
        (rc = upload(d_raw, hpk.raw, s)))
        return fail(rc);
    {
        // the parts go into ONE pinned staging buffer at their final offsets (host threads), then nine uploads: a pageable
        // source costs a synchronous bounce per call, and there were nine per part
        static thread_local PinBuf pin;
        const uint64_t sz[9] = {tot_vars * 4, tot_vars * 4, tot_vars, tot_rows * 4, tot_rows * 4, tot_rows * 4, tot_rows * 8, tot_ral, tot_rq};
        uint64_t off9[10]; off9[0] = 0;
        for (int k = 0; k < 9; ++k) off9[k + 1] = (off9[k] + sz[k] + 63) / 64 * 64;
        void* dst9[9] = {b->d_vlo.p, b->d_vhi.p, b->d_vflags.p, b->d_rstart.p, b->d_rend.p, b->d_rword.p, d_rcell.p, d_ral.p, d_rq.p};
        if (off9[9] > (256ull << 20)) {
            // a batch of gigabytes (thousands of large blocks): straight from the parts, no second copy in host memory
            for (unsigned t = 0; t < nt; ++t) {
                const HostPack& q = parts[t];
                const PartBase& o = base[t];
                const void* src9[9] = {q.vlo.data(), q.vhi.data(), q.vflags.data(), q.rstart.data(), q.rend.data(), q.rword.data(), q.rcell.data(), q.raw_alleles.data(), q.raw_quals.data()};
                const uint64_t at9[9] = {o.var * 4, o.var * 4, o.var, o.read * 4, o.read * 4, o.read * 4, o.read * 8, o.ral, o.rq};
                const size_t n9[9] = {q.vlo.size() * 4, q.vhi.size() * 4, q.vflags.size(), q.rstart.size() * 4, q.rend.size() * 4, q.rword.size() * 4, q.rcell.size() * 8, q.raw_alleles.size(), q.raw_quals.size()};
                for (int k = 0; k < 9; ++k)
                    if (n9[k] && hipMemcpyAsync(static_cast<unsigned char*>(dst9[k]) + at9[k], src9[k], n9[k], hipMemcpyHostToDevice, s) != hipSuccess) {
                        set_error("upload failed: %s", hipGetErrorString(hipGetLastError()));
                        return fail(HP_ERR_HIP);
                    }
            }
        } else {
        if ((rc = pin.reserve((size_t)off9[9] + 64)) != HP_OK) return fail(rc);
        uint8_t* P = pin.p;
        WorkerPool::get().run(nt, [&](unsigned t) {
            const HostPack& q = parts[t];
            const PartBase& o = base[t];
            auto put = [&](int k, uint64_t off_bytes, const void* src, size_t bytes) { if (bytes) std::memcpy(P + off9[k] + off_bytes, src, bytes); };
            put(0, o.var * 4, q.vlo.data(), q.vlo.size() * 4); put(1, o.var * 4, q.vhi.data(), q.vhi.size() * 4);
            put(2, o.var, q.vflags.data(), q.vflags.size()); put(3, o.read * 4, q.rstart.data(), q.rstart.size() * 4);
            put(4, o.read * 4, q.rend.data(), q.rend.size() * 4); put(5, o.read * 4, q.rword.data(), q.rword.size() * 4);
            put(6, o.read * 8, q.rcell.data(), q.rcell.size() * 8); put(7, o.ral, q.raw_alleles.data(), q.raw_alleles.size());
            put(8, o.rq, q.raw_quals.data(), q.raw_quals.size());
        });
        for (int k = 0; k < 9; ++k)
            if (sz[k] && dev_copy(dst9[k], P + off9[k], sz[k], s) != HP_OK) return fail(HP_ERR_HIP);   // (P is pinned: the copy kernel reads it in place)
        }
    }
    if ((rc = upload(b->d_row_block, b->row_block_h, s)) != HP_OK) return fail(rc);
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp]   uploads queued at %.1f ms\n", wall_ms() - t_pack0); fflush(stderr); }
    // the bit-sliced plane words are built on the device from the caller's cells
    if (!b->row_block_h.empty()) {
        PackDev P{};
        P.desc = b->d_desc.as<BlockDesc>(); P.raw = d_raw.as<PackRaw>(); P.row_block = b->d_row_block.as<uint32_t>();
        P.rstart = b->d_rstart.as<uint32_t>(); P.rend = b->d_rend.as<uint32_t>(); P.rword = b->d_rword.as<uint32_t>();
        P.rcell = d_rcell.as<uint64_t>(); P.alleles = d_ral.as<uint8_t>(); P.quals = d_rq.as<uint8_t>();
        P.words = b->d_words.as<uint32_t>(); P.n_rows = b->row_block_h.size();
        hipLaunchKernelGGL(hp_pack_words_kernel, dim3((unsigned)((P.n_rows + 63) / 64)), dim3(64), 0, s, P);
        if (hipGetLastError() != hipSuccess) { set_error("hp_pack_words_kernel launch failed"); return fail(HP_ERR_HIP); }
    }
    // per-position cell tables are derived on the device from the rows just uploaded
    if ((rc = b->d_ctab.alloc(hpk.cell_total * sizeof(uint32_t) + 16)) != HP_OK) return fail(rc);
    if (hpk.cell_total && !b->row_block_h.empty()) {
        if (hipMemsetAsync(b->d_ctab.p, 0, hpk.cell_total * sizeof(uint32_t), s) != hipSuccess) { set_error("hipMemsetAsync failed"); return fail(HP_ERR_HIP); }
        CtabDev T{};
        T.desc = b->d_desc.as<BlockDesc>(); T.row_block = b->d_row_block.as<uint32_t>();
        T.rstart = b->d_rstart.as<uint32_t>(); T.rend = b->d_rend.as<uint32_t>(); T.rword = b->d_rword.as<uint32_t>();
        T.words = b->d_words.as<uint32_t>(); T.ctab = b->d_ctab.as<uint32_t>(); T.n_rows = b->row_block_h.size();
        T.vflags = b->d_vflags.as<uint8_t>();
        hipLaunchKernelGGL(hp_build_ctab_kernel, dim3((unsigned)T.n_rows), dim3(64), 0, s, T);
        if (hipGetLastError() != hipSuccess) { set_error("hp_build_ctab_kernel launch failed"); return fail(HP_ERR_HIP); }
    }
    if ((rc = b->d_hapw.alloc(hpk.chunk_total * sizeof(Win) + 16)) != HP_OK) return fail(rc);
    if ((rc = b->d_H.alloc(b->sum_h * 8)) != HP_OK) return fail(rc);
    if ((rc = b->d_h1.alloc(b->sum_n)) != HP_OK) return fail(rc);
    if ((rc = b->d_h2.alloc(b->sum_n)) != HP_OK) return fail(rc);
    if ((rc = b->d_stats.alloc(n_blocks * sizeof(hp_phase_stats))) != HP_OK) return fail(rc);
    if ((rc = b->d_counters.alloc(n_blocks * sizeof(hp_work_counters))) != HP_OK) return fail(rc);
    if ((rc = b->d_status.alloc(n_blocks * sizeof(int32_t))) != HP_OK) return fail(rc);
    if (dev_io_sync(s) != HP_OK) { set_error("upload failed"); return fail(HP_ERR_HIP); }
    if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] hp_batch_create total %.1f ms\n", wall_ms() - t_pack0); fflush(stderr); }
    if (status) *status = HP_OK;
    return b.release();
}

// Device -> host for the batch's small result arrays: dev_get (hp_common.h), not hipMemcpy - see there.
namespace {
struct Fetch { void* dst; const void* src; size_t n; };
int fetch_all(hp_batch*, hipStream_t st, std::initializer_list<Fetch> fs) {
    for (const Fetch& f : fs) { const int rc = dev_get(f.dst, f.src, f.n, st); if (rc != HP_OK) { dev_io_abort(st); return rc; } }
    return dev_io_sync(st);
}
}  // namespace

int hp_batch_solve(hp_batch* b, void* stream, float* kernel_ms) {
    if (!b) { set_error("null batch"); return HP_ERR_ARG; }
    HP_HIP_CHECK(hp_set_device(b->device));
    hipStream_t st = stream ? (hipStream_t)stream : b->stream;
    std::vector<int32_t> status(b->n_blocks, ST_PENDING);
    { const int rc0 = dev_put(b->d_status.p, status.data(), status.size() * 4, st); if (rc0 != HP_OK) return rc0; }
    HP_HIP_CHECK(hipEventRecord(b->ev0, st));

    // pass 0: every block, scratch sized generously above the clean-data bound (<= 4N+1 nodes); blocks whose
    // frontier outgrows it (noisy data) keep their finished heuristic and only their main search is re-run
    // with 4x the capacity until it fits (or memory runs out).
    // (round 6: the first attempt's scratch is sized from a memory budget, not from the clean-data bound alone - first_pass_cap)
    uint32_t cap_main = first_pass_cap(b->max_n, b->order.size(), b->n_cu);   // node_index < 2^38 is guaranteed by cap64 <= 0xF0000000 below
    std::vector<uint32_t> blk_cap(b->n_blocks, cap_main);   // the capacity every block's latest attempt ran with
    // Large blocks that would be the critical path get a segment-parallel heuristic on a second stream, beside the
    // ordinary pass over all other blocks; their (short) main search follows on that stream.
    HP_HIP_CHECK(hipEventRecord(b->ev_fork, st));
    HP_HIP_CHECK(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
    std::vector<uint32_t> seg_blocks;
    int rc = launch_segments(b, b->stream2, seg_blocks);
    if (rc != HP_OK) return rc;
    std::vector<uint32_t> items_seq = b->order, items_seg;
    if (!seg_blocks.empty()) {
        std::vector<uint8_t> is_seg(b->n_blocks, 0);
        for (uint32_t i : seg_blocks) is_seg[i] = 1;
        items_seq.clear();
        for (uint32_t i : b->order) (is_seg[i] ? items_seg : items_seq).push_back(i);
        uint32_t seg_max_n = 0;
        for (uint32_t i : items_seg) seg_max_n = std::max(seg_max_n, b->desc[i].n_vars);
        const uint32_t cap_seg = first_pass_cap(seg_max_n, items_seg.size(), b->n_cu);
        for (uint32_t i : items_seg) blk_cap[i] = cap_seg;
        rc = launch_pass(b, b->stream2, items_seg, cap_seg, b->g_main_pool, b->g_main_heap, b->g_sub_pool, b->g_sub_heap,
                         b->g_tracker, b->g_slots, b->g_cap, b->d_order2, true);
        if (rc != HP_OK) return rc;
        HP_HIP_CHECK(hipEventRecord(b->ev_join, b->stream2));
        uint32_t seq_max_n = 0;
        for (uint32_t i : items_seq) seq_max_n = std::max(seq_max_n, b->desc[i].n_vars);
        cap_main = first_pass_cap(seq_max_n, items_seq.size(), b->n_cu);
        for (uint32_t i : items_seq) blk_cap[i] = cap_main;
    }
    if (!items_seq.empty()) {
        rc = launch_pass(b, st, items_seq, cap_main, b->s_main_pool, b->s_main_heap, b->s_sub_pool, b->s_sub_heap,
                         b->s_tracker, b->scratch_slots, b->scratch_cap_main, b->d_order1);
        if (rc != HP_OK) return rc;
    }
    if (!seg_blocks.empty()) HP_HIP_CHECK(hipStreamWaitEvent(st, b->ev_join, 0));
    HP_HIP_CHECK(hipEventRecord(b->ev1, st));
    HP_HIP_CHECK(hipStreamSynchronize(st));
    float ms_total = 0.f;
    HP_HIP_CHECK(hipEventElapsedTime(&ms_total, b->ev0, b->ev1));
    if ((rc = fetch_all(b, st, {{status.data(), b->d_status.p, status.size() * 4}})) != HP_OK) return rc;

    std::vector<uint32_t> retry;
    for (uint32_t i : b->order) if (status[i] == ST_OVERFLOW_MAIN) retry.push_back(i);
    DevBuf r_main_pool, r_main_heap, r_sub_pool, r_sub_heap, r_tracker, r_items;
    uint32_t r_slots = 0, r_cap = 0;
    std::vector<hp_work_counters> ovf_ctr;
    while (!retry.empty()) {
        // An attempt that ran out of scratch left how far it had come (hp_work_counters.reserved[2] = variants reached): the next
        // attempt gets what the whole block needs at that rate, with half as much again on top, and never less than twice what
        // failed. (Until round 6: 4x per attempt - a 2 500-het block at 60x with 15 % wrong cells ran its main search three times,
        // 15 + 50 + 51 ms.)
        ovf_ctr.resize(b->n_blocks);
        if ((rc = fetch_all(b, st, {{ovf_ctr.data(), b->d_counters.p, ovf_ctr.size() * sizeof(hp_work_counters)}})) != HP_OK) return rc;
        uint64_t cap64 = 0;
        for (uint32_t i : retry) {
            const uint64_t n = b->desc[i].n_vars, reached = std::min<uint64_t>(std::max<uint64_t>(ovf_ctr[i].reserved[2], 1), n);
            const uint64_t est = (uint64_t)blk_cap[i] * n / reached * 3 / 2 + 2048;
            cap64 = std::max<uint64_t>(cap64, std::min<uint64_t>(std::max<uint64_t>(est, 2ull * blk_cap[i]), 64ull * blk_cap[i]));
        }
        if (cap64 > 0xF0000000ull) { set_error("search frontier exceeds 2^32 nodes"); return HP_ERR_OOM; }
        for (uint32_t i : retry) blk_cap[i] = (uint32_t)cap64;
        if (std::getenv("HP_DEBUG")) { fprintf(stderr, "[hp] %zu blocks ran out of main-search scratch: again with %llu nodes\n", retry.size(), (unsigned long long)cap64); fflush(stderr); }
        HP_HIP_CHECK(hipEventRecord(b->ev0, st));
        rc = launch_pass(b, st, retry, (uint32_t)cap64, r_main_pool, r_main_heap, r_sub_pool, r_sub_heap, r_tracker, r_slots,
                         r_cap, r_items);
        if (rc != HP_OK) return rc;
        HP_HIP_CHECK(hipEventRecord(b->ev1, st));
        HP_HIP_CHECK(hipStreamSynchronize(st));
        float ms = 0.f;
        HP_HIP_CHECK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
        ms_total += ms;
        if ((rc = fetch_all(b, st, {{status.data(), b->d_status.p, status.size() * 4}})) != HP_OK) return rc;
        std::vector<uint32_t> again;
        for (uint32_t i : retry) if (status[i] == ST_OVERFLOW_MAIN) again.push_back(i);
        retry.swap(again);
    }
    if (std::getenv("HP_DEBUG")) {   // the blocks a solve's time is made of: shader-clock ticks of the heuristic chain and of the main search
        std::vector<hp_work_counters> ctr(b->n_blocks);
        if (fetch_all(b, st, {{ctr.data(), b->d_counters.p, ctr.size() * sizeof(hp_work_counters)}}) == HP_OK) {
            std::vector<uint32_t> by(b->n_blocks);
            std::iota(by.begin(), by.end(), 0u);
            std::sort(by.begin(), by.end(), [&](uint32_t x, uint32_t y) { return ctr[x].reserved[0] + ctr[x].reserved[1] > ctr[y].reserved[0] + ctr[y].reserved[1]; });
            for (size_t k = 0; k < std::min<size_t>(4, by.size()); ++k) {
                const hp_work_counters& c = ctr[by[k]];
                fprintf(stderr, "[hp]   block %u: %u hets x %u reads, heuristic chain %.1f M ticks, main search %.1f M ticks (%llu pops), %llu sub-solver pops\n", by[k], b->desc[by[k]].n_vars,
                        b->desc[by[k]].n_reads, c.reserved[0] / 1e6, c.reserved[1] / 1e6, (unsigned long long)c.main_pops, (unsigned long long)c.sub_pops);
            }
            fflush(stderr);
        }
    }
    if (kernel_ms) *kernel_ms = ms_total;
    g_last_kernel_ms = ms_total;
    b->solved = true;
    for (size_t i = 0; i < status.size(); ++i) {
        if (status[i] != ST_OK) b->solved = false;
        if (status[i] == ST_INVARIANT) {
            set_error("block %zu: solver invariant violated (the reference would panic/assert, astar_phaser.rs:268,284,360,529,631)", i);
            return HP_ERR_INVARIANT;
        }
        if (status[i] != ST_OK) { set_error("block %zu: unexpected device status %d", i, status[i]); return HP_ERR_HIP; }
    }
    return HP_OK;
}

int hp_batch_results(hp_batch* b, uint8_t* h1, uint8_t* h2, hp_phase_stats* stats, hp_work_counters* counters, uint64_t* heuristics) {
    if (!b) { set_error("null batch"); return HP_ERR_ARG; }
    HP_HIP_CHECK(hp_set_device(b->device));
    return fetch_all(b, b->stream, {{h1, b->d_h1.p, b->sum_n}, {h2, b->d_h2.p, b->sum_n}, {stats, b->d_stats.p, b->n_blocks * sizeof(hp_phase_stats)},
                                    {counters, b->d_counters.p, b->n_blocks * sizeof(hp_work_counters)}, {heuristics, b->d_H.p, b->sum_h * 8}});
}

int hp_batch_postprocess(hp_batch* b, uint64_t* span_counts, uint8_t* haplotag, uint32_t* first_het) {
    if (!b) { set_error("null batch"); return HP_ERR_ARG; }
    if (!b->solved) { set_error("hp_batch_postprocess needs a successful hp_batch_solve first"); return HP_ERR_ARG; }
    HP_HIP_CHECK(hp_set_device(b->device));
    hipStream_t st = b->stream;
    int rc;
    if (!b->d_js.p) {
        std::vector<uint32_t>& junc_block = b->h_junc_block;
        std::vector<uint64_t>& junc_off = b->h_junc_off;
        junc_block.clear();
        junc_off.assign(b->n_blocks, 0);
        uint64_t nj = 0;
        for (size_t i = 0; i < b->n_blocks; ++i) {
            junc_off[i] = nj;
            const uint32_t n = b->desc[i].n_vars;
            for (uint32_t j = 0; j + 1 < n; ++j) junc_block.push_back((uint32_t)i);
            nj += n > 0 ? n - 1 : 0;
        }
        b->n_junctures = nj;
        if ((rc = upload(b->d_junc_block, junc_block, st)) != HP_OK) return rc;
        if ((rc = upload(b->d_junc_off, junc_off, st)) != HP_OK) return rc;
        if ((rc = b->d_haplotag.alloc(b->n_rows_packed + 16)) || (rc = b->d_first_het.alloc(b->n_rows_packed * 4 + 16)) ||
            (rc = b->d_js.alloc(b->n_rows_packed * 4 + 16)) || (rc = b->d_je.alloc(b->n_rows_packed * 4 + 16)) ||
            (rc = b->d_span.alloc(nj * 8 + 16)))
            return rc;
    }
    PostDev P{};
    P.desc = b->d_desc.as<BlockDesc>(); P.row_block = b->d_row_block.as<uint32_t>();
    P.rstart = b->d_rstart.as<uint32_t>(); P.rend = b->d_rend.as<uint32_t>(); P.rword = b->d_rword.as<uint32_t>();
    P.words = b->d_words.as<uint32_t>(); P.vlo = b->d_vlo.as<uint32_t>(); P.vhi = b->d_vhi.as<uint32_t>();
    P.hapw = b->d_hapw.as<Win>(); P.n_rows_total = (uint32_t)b->n_rows_packed; P.n_junctures_total = b->n_junctures;
    P.haplotag = b->d_haplotag.as<uint8_t>(); P.first_het = b->d_first_het.as<uint32_t>();
    P.js = b->d_js.as<uint32_t>(); P.je = b->d_je.as<uint32_t>();
    P.junc_block = b->d_junc_block.as<uint32_t>(); P.junc_off = b->d_junc_off.as<uint64_t>(); P.span_counts = b->d_span.as<uint64_t>();
    HP_HIP_CHECK(hipEventRecord(b->ev0, st));
    if (b->n_rows_packed)
        hipLaunchKernelGGL(hp_post_rows_kernel, dim3((unsigned)((b->n_rows_packed + 63) / 64)), dim3(64), 0, st, P);
    if (b->n_junctures)
        hipLaunchKernelGGL(hp_post_spans_kernel, dim3((unsigned)((b->n_junctures + 63) / 64)), dim3(64), 0, st, P);
    HP_HIP_CHECK(hipGetLastError());
    HP_HIP_CHECK(hipEventRecord(b->ev1, st));
    HP_HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0.f;
    HP_HIP_CHECK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    g_last_kernel_ms = ms;
    const bool tags = haplotag || first_het;
    std::vector<uint8_t> ht(tags ? b->n_rows_packed + 1 : 1);
    std::vector<uint32_t> fh(tags ? b->n_rows_packed + 1 : 1);
    if ((rc = fetch_all(b, st, {{span_counts, b->d_span.p, b->n_junctures * 8}, {tags ? ht.data() : nullptr, b->d_haplotag.p, b->n_rows_packed},
                                {tags ? fh.data() : nullptr, b->d_first_het.p, b->n_rows_packed * 4}})) != HP_OK) return rc;
    if (tags) {
        // back to the caller's row order; inert rows (start == end) are untagged
        if (haplotag) std::memset(haplotag, 2, b->caller_rows);
        if (first_het) for (uint64_t i = 0; i < b->caller_rows; ++i) first_het[i] = 0xFFFFFFFFu;
        for (uint64_t r = 0; r < b->n_rows_packed; ++r) {
            const uint64_t dst = b->caller_row_off[b->row_block_h[r]] + b->row_orig[r];
            if (haplotag) haplotag[dst] = ht[r];
            if (first_het) first_het[dst] = fh[r];
        }
    }
    return HP_OK;
}

void hp_batch_destroy(hp_batch* b) { delete b; }

static int solve_on_device(size_t n, const hp_block_view* blks, const hp_astar_params* p, uint8_t* const* h1,
                           uint8_t* const* h2, hp_phase_stats* out, int device_id) {
    int status = HP_OK;
    const bool verbose = std::getenv("HP_DEBUG") != nullptr;
    auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t0 = now();
    hp_batch* b = hp_batch_create(n, blks, p, device_id, &status);
    if (!b) return status;
    std::unique_ptr<hp_batch> guard(b);
    const double t1 = now();
    int rc = hp_batch_solve(b, nullptr, nullptr);
    if (rc != HP_OK) return rc;
    const double t2 = now();
    std::vector<uint8_t> a1(b->sum_n), a2(b->sum_n);
    rc = hp_batch_results(b, a1.data(), a2.data(), out, nullptr, nullptr);
    if (rc != HP_OK) return rc;
    if (verbose) { fprintf(stderr, "[hp] solve_on_device: create %.2f ms, solve %.2f ms (kernel %.2f), results %.2f ms\n", t1 - t0, t2 - t1, g_last_kernel_ms, now() - t2); fflush(stderr); }
    for (size_t i = 0; i < n; ++i) {
        const BlockDesc& d = b->desc[i];
        if (h1 && h1[i]) std::memcpy(h1[i], a1.data() + d.var_off, d.n_vars);
        if (h2 && h2[i]) std::memcpy(h2[i], a2.data() + d.var_off, d.n_vars);
    }
    return HP_OK;
}

int hp_astar_solve_batch(size_t n_blocks, const hp_block_view* blks, const hp_astar_params* p, uint8_t* const* h1,
                         uint8_t* const* h2, hp_phase_stats* out, int device_id) {
    if (n_blocks == 0) return HP_OK;
    if (!blks || !p) { set_error("null argument"); return HP_ERR_ARG; }
    if (device_id >= 0) return solve_on_device(n_blocks, blks, p, h1, h2, out, device_id);

    // device_id == -1: host-side work queue over every visible GPU: blocks sorted by
    // estimated work (LPT), cut into chunks, one worker thread per device pulls chunks. No collective.
    const int real_dev = hp_device_count();
    if (real_dev <= 0) { set_error("no HIP device visible; there is no CPU fallback"); return HP_ERR_HIP; }
    // HP_QUEUE_WORKERS=n (test hook): run the multi-device queue with n workers even on a 1-GPU box; worker w
    // uses device w % real_dev
    const char* wenv = std::getenv("HP_QUEUE_WORKERS");
    const int ndev = wenv ? std::max(1, std::atoi(wenv)) : real_dev;
    if (ndev == 1) return solve_on_device(n_blocks, blks, p, h1, h2, out, 0);
    std::vector<uint32_t> order(n_blocks);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t c) {
        const uint64_t wa = blks[a].row_off ? blks[a].row_off[blks[a].n_reads] : 0, wc = blks[c].row_off ? blks[c].row_off[blks[c].n_reads] : 0;
        return wa > wc;
    });
    // interleave so that every chunk carries a similar mix of large and small blocks
    const size_t n_chunks = std::min<size_t>(n_blocks, (size_t)ndev * 4);
    std::vector<std::vector<uint32_t>> chunks(n_chunks);
    for (size_t i = 0; i < n_blocks; ++i) chunks[i % n_chunks].push_back(order[i]);
    std::atomic<size_t> next{0};
    std::atomic<int> first_err{HP_OK};
    std::vector<std::string> errs(ndev);
    std::vector<std::thread> workers;
    for (int dev = 0; dev < ndev; ++dev) {
        workers.emplace_back([&, dev]() {
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= n_chunks || first_err.load() != HP_OK) break;
                const auto& ids = chunks[c];
                std::vector<hp_block_view> v(ids.size());
                std::vector<uint8_t*> p1(ids.size()), p2(ids.size());
                std::vector<hp_phase_stats> st(ids.size());
                for (size_t k = 0; k < ids.size(); ++k) { v[k] = blks[ids[k]]; p1[k] = h1 ? h1[ids[k]] : nullptr; p2[k] = h2 ? h2[ids[k]] : nullptr; }
                int rc = solve_on_device(ids.size(), v.data(), p, p1.data(), p2.data(), st.data(), dev % real_dev);
                if (rc != HP_OK) { int exp = HP_OK; if (first_err.compare_exchange_strong(exp, rc)) errs[dev] = hp_last_error(); break; }
                if (out) for (size_t k = 0; k < ids.size(); ++k) out[ids[k]] = st[k];
            }
        });
    }
    for (auto& t : workers) t.join();
    if (first_err.load() != HP_OK) {
        for (auto& e : errs) if (!e.empty()) { set_error("%s", e.c_str()); break; }
        return first_err.load();
    }
    return HP_OK;
}

// hp_astar_solve calls that are in flight together (HiPhase's thread pool, main.rs:385-408) become one resident batch:
// see hp_combine.h. Requests are grouped by (device, solver parameters); a group that fails as a batch (one block outside
// the packed-key limits fails hp_batch_create for all) is re-run block by block so that every caller gets its own status.
namespace {
struct SolveReq {
    const hp_block_view* blk; hp_astar_params prm; int device;
    uint8_t* h1; uint8_t* h2; hp_phase_stats st{};
    int rc = HP_OK; std::string err; bool done = false;
};
void run_solve_batch(std::vector<SolveReq*>& batch);
// never destroyed: its service thread may outlive every static destructor
hp::Combiner<SolveReq>& g_solve_combiner() { static auto* c = new hp::Combiner<SolveReq>(run_solve_batch); return *c; }
void run_solve_batch(std::vector<SolveReq*>& batch) {
    std::vector<char> taken(batch.size(), 0);
    for (size_t i = 0; i < batch.size(); ++i) {
        if (taken[i]) continue;
        std::vector<SolveReq*> grp;
        for (size_t j = i; j < batch.size(); ++j)
            if (!taken[j] && batch[j]->device == batch[i]->device && batch[j]->prm.min_queue_size == batch[i]->prm.min_queue_size &&
                batch[j]->prm.queue_increment == batch[i]->prm.queue_increment && batch[j]->prm.max_segment_size == batch[i]->prm.max_segment_size) {
                taken[j] = 1; grp.push_back(batch[j]);
            }
        std::vector<hp_block_view> views(grp.size());
        std::vector<uint8_t*> a1(grp.size()), a2(grp.size());
        std::vector<hp_phase_stats> st(grp.size());
        for (size_t k = 0; k < grp.size(); ++k) { views[k] = *grp[k]->blk; a1[k] = grp[k]->h1; a2[k] = grp[k]->h2; }
        int rc = solve_on_device(grp.size(), views.data(), &grp[0]->prm, a1.data(), a2.data(), st.data(), grp[0]->device);
        if (rc == HP_OK) { for (size_t k = 0; k < grp.size(); ++k) { grp[k]->st = st[k]; grp[k]->rc = HP_OK; } continue; }
        if (grp.size() == 1) { grp[0]->rc = rc; grp[0]->err = hp_last_error(); continue; }
        for (SolveReq* r : grp) {   // find out whose block it was
            uint8_t* b1[1] = {r->h1};
            uint8_t* b2[1] = {r->h2};
            r->rc = solve_on_device(1, r->blk, &r->prm, b1, b2, &r->st, r->device);
            if (r->rc != HP_OK) r->err = hp_last_error();
        }
    }
}
}  // namespace

int hp_astar_solve(const hp_block_view* blk, const hp_astar_params* p, uint8_t* h1, uint8_t* h2, hp_phase_stats* out) {
    if (!blk || !p) { set_error("null argument"); return HP_ERR_ARG; }
    if (!hp::Combiner<SolveReq>::enabled()) {
        uint8_t* a1[1] = {h1};
        uint8_t* a2[1] = {h2};
        hp_phase_stats st{};
        int rc = solve_on_device(1, blk, p, a1, a2, &st, hp_default_device());
        if (rc == HP_OK && out) *out = st;
        return rc;
    }
    SolveReq r{blk, *p, hp_default_device(), h1, h2};
    g_solve_combiner().submit(&r);
    if (r.rc != HP_OK) set_error("%s", r.err.c_str());
    else if (out) *out = r.st;
    return r.rc;
}

}  // extern "C"
