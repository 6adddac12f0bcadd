// hp_block.hip — whole phase blocks behind ONE entry: phaser::solve_block from the decoded records on
// (reference src/phaser.rs:513-630), for any number of blocks at once.
//
//   hp_solve_blocks / hp_blockset_* =
//     read_parsing::load_full_read_segments (read_parsing.rs:520-637)   [or load_read_segments, :47-113]
//        global_realignment per record (:652-867)      -> ONE graph-WFA device batch over the records of ALL blocks
//        Err(MaxEditDistance) -> local_realignment      -> hp_local_realign_batch for the failed records (:564-575)
//        the order-dependent `global_disabled` switch   -> replayed per block in BAM order (:556-600)
//        quality assignment (:803-835), ReadSegment::new, collapse per read name, min_matched_alleles split (:611-629)
//     astar_phaser::astar_solver (phaser.rs:541-543)    -> hp_batch_* (one resident batch of all blocks)
//     get_solution_span_counts / haplotag_reads (:546, :614-630) -> hp_batch_postprocess on the resident matrix
//
// The quality assignment, the per-read-name collapse and the fallback replay used to exist three times ABOVE the C
// ABI (Python mirror, C++ mirror, tests); they now live here once, behind it. They are host logic over a few bytes per
// record (the reference's own statements, cited inline); everything that scales with bases or with the search runs in
// the kernels. No CPU fallback: without a device the first device stage fails with HP_ERR_HIP.
#include "hp_common.h"
#include "hp_combine.h"
#include "hp_wfa2_host.h"
#include "hp_block.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <deque>
#include <ctime>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace hp {

namespace {

double blk_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// read_parsing.rs:18-22: base qualities per type; global re-alignment doubles them (:815)
int base_quality(uint32_t variant_type) {
    switch (variant_type) {
        case 0: return 80;                       // Snv
        case 1: case 2: case 3: return 10;       // Insertion, Deletion, Indel
        case 4: case 5: return 20;               // SvInsertion, SvDeletion
        case 9: return 40;                       // TandemRepeat
        default: return -1;                      // panic!("No implementation for matching ...") (:829)
    }
}

// ReadSegment::new (read_segments.rs:40-62) from a window [w0, w0 + n) of the block-length vectors (everything outside
// the window is NoOverlap with quality 0)
Segment segment_new(Arena& ar, const uint8_t* alleles, const uint8_t* quals, uint32_t w0, uint32_t n, uint32_t n_hets) {
    Segment s;
    uint32_t first = n, last = n;
    for (uint32_t i = 0; i < n; ++i) if (alleles[i] < HP_ALLELE_AMBIGUOUS) { first = i; break; }
    for (uint32_t i = n; i-- > 0;) if (alleles[i] < HP_ALLELE_AMBIGUOUS) { last = i + 1; break; }
    if (first == n) { s.start = s.end = n_hets; return s; }   // no set allele: len..len (:58-61)
    s.start = w0 + first; s.end = w0 + last;
    uint8_t* a = ar.get(last - first);
    uint8_t* q = ar.get(last - first);
    std::memcpy(a, alleles + first, last - first);
    std::memcpy(q, quals + first, last - first);
    s.alleles = a; s.quals = q;
    return s;
}
inline uint8_t seg_allele(const Segment& s, uint32_t i) { return (i >= s.start && i < s.end) ? s.alleles[i - s.start] : (uint8_t)HP_ALLELE_NOOVERLAP; }
inline uint8_t seg_qual(const Segment& s, uint32_t i) { return (i >= s.start && i < s.end) ? s.quals[i - s.start] : (uint8_t)0; }
// ReadSegment::collapse (read_segments.rs:71-121). Returns false when `assert!(quals[i] > 0)` (:105) would fire.
bool segment_collapse(Arena& ar, const std::vector<const Segment*>& rs, uint32_t n_hets, Segment& out) {
    if (rs.size() == 1) { out = *rs[0]; return true; }
    uint32_t min_start = rs[0]->start, max_end = rs[0]->end;
    for (auto* r : rs) { min_start = std::min(min_start, r->start); max_end = std::max(max_end, r->end); }
    std::vector<uint8_t> alleles(max_end > min_start ? max_end - min_start : 0, (uint8_t)HP_ALLELE_NOOVERLAP), quals(alleles.size(), 0);
    for (auto* r : rs)
        for (uint32_t i = min_start; i < max_end; ++i) {
            const uint8_t a = seg_allele(*r, i), q = seg_qual(*r, i);
            uint8_t& ca = alleles[i - min_start];
            uint8_t& cq = quals[i - min_start];
            if (a == HP_ALLELE_NOOVERLAP) continue;
            if (ca == HP_ALLELE_NOOVERLAP) { ca = a; cq = q; }
            else if (ca == HP_ALLELE_AMBIGUOUS) {}
            else if (ca == a) { cq = std::max(cq, q); if (cq == 0) return false; }
            else { ca = HP_ALLELE_AMBIGUOUS; cq = 0; }
        }
    out = segment_new(ar, alleles.data(), quals.data(), min_start, (uint32_t)alleles.size(), n_hets);
    return true;
}
// joint_stats += read_stats (writers/phase_stats.rs:107-121) for the fields the per-record counts feed
void add_read_stats(hp_read_stats& acc, const hp_read_stats& r) {
    acc.num_alleles += r.num_alleles;
    for (int t = 0; t < HP_N_VARIANT_TYPES; ++t) {
        acc.exact_matches[t] += r.exact_matches[t]; acc.inexact_matches[t] += r.inexact_matches[t]; acc.failed_matches[t] += r.failed_matches[t];
        acc.allele0_matches[t] += r.allele0_matches[t]; acc.allele1_matches[t] += r.allele1_matches[t];
    }
}
uint32_t seg_num_set(const Segment& s) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < s.end - s.start; ++i) n += s.alleles[i] < HP_ALLELE_AMBIGUOUS;
    return n;
}

}  // namespace
}  // namespace hp

using namespace hp;

namespace {

// indices [first, last) of the variants with lo <= position <= hi (read_parsing.rs:688-700, 721-730); the reference
// asserts that they are contiguous (:715), which position-sorted input guarantees
bool overlap_range(const hp_wfa_variant* v, uint32_t n, bool sorted, int64_t lo, int64_t hi, uint32_t& first, uint32_t& last, bool& contiguous) {
    contiguous = true;
    if (sorted) {
        const hp_wfa_variant* a = std::lower_bound(v, v + n, lo, [](const hp_wfa_variant& x, int64_t p) { return x.position < p; });
        const hp_wfa_variant* b = std::upper_bound(v, v + n, hi, [](int64_t p, const hp_wfa_variant& x) { return p < x.position; });
        first = (uint32_t)(a - v); last = (uint32_t)(b - v);
        return last > first;
    }
    bool any = false;
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (lo <= v[i].position && v[i].position <= hi) { if (!any) { first = i; any = true; } last = i + 1; ++cnt; }
    if (any && cnt != last - first) contiguous = false;
    return any;
}

}  // namespace

int hp::blockset_layout(hp_blockset* bs, size_t n_blocks, const hp_block_input* in, const hp_block_params* p, int device_id) {
    if (!in || !p) { set_error("null argument"); return HP_ERR_ARG; }
    const double t0 = blk_now_ms();
    bs->n_blocks = n_blocks; bs->in = in; bs->prm = *p;
    bs->device = device_id < 0 ? hp_default_device() : device_id;
    // a set that failed between its alignment stage and blockset_rows' join may have left its late pass running: it writes
    // bs->wfa_out / bs->alleles and reads the previous caller's inputs - join it before any of that is reassigned
    if (bs->wfa) (void)w2_session_finish(bs->wfa);
    if (bs->small_inflight && bs->small_helper) { bs->small_helper->wait(); bs->small_inflight = false; }   // (likewise a small set's pass on the slot's helper thread)
    bs->wfa_ready = false;
    bs->prep[0] = bs->prep[1] = bs->prep[2] = bs->prep[3] = 0.0;
    if (bs->meta.size() < n_blocks) bs->meta.resize(n_blocks);
    if (bs->st.size() < n_blocks) bs->st.resize(n_blocks);
    bs->job_first.assign(n_blocks + 1, 0);
    const char* mj = std::getenv("HP_WFA2_MIN_JOBS");
    // (sets below this take the latency path - the dense-band kernel, one wavefront per read. 1 024 since round 4: a merged set of
    // the per-block dispatcher is a few thousand records, its noisy reads cost the dense-band kernel a 500-edit band each, and the
    // compact road settles them by the reference-window test instead: 64 blocking callers 77 -> 84 k hets/s, 2 560 asynchronous
    // ones 626 -> 748 k. hp_wfa_assign_batch keeps its own 4 608.)
    const size_t min_jobs = mj ? (size_t)std::strtoull(mj, nullptr, 10) : 1024;
    // per block (host threads over blocks): validation, and for every record its overlaps (read_parsing.rs:688-730)
    std::vector<uint32_t> njobs(n_blocks, 0);
    {
        std::atomic<size_t> next{0};
        std::atomic<int> first_rc{HP_OK};
        unsigned nt = host_threads(16u);
        if (const char* e = std::getenv("HP_BLOCK_HOST_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
        nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, n_blocks / 4));
        std::vector<std::string> errs(nt);
        auto fail = [&](unsigned t, int rc) { int exp = HP_OK; if (first_rc.compare_exchange_strong(exp, rc)) errs[t] = hp_last_error(); };
        WorkerPool::get().run(nt, [&](unsigned t) {
            for (;;) {
                const size_t b = next.fetch_add(1);
                if (b >= n_blocks || first_rc.load() != HP_OK) return;
                const hp_block_input& B = in[b];
                if (B.n_hets == 0) { set_error("block %zu has no variants (phaser.rs:415-434 short-circuits those before this path)", b); return fail(t, HP_ERR_ARG); }
                if (!B.hets || !B.het_types || (B.n_homs && !B.homs) || (B.n_records && !B.records) || (p->global_realignment && !B.reference)) {
                    set_error("block %zu: null array", b); return fail(t, HP_ERR_ARG);
                }
                if (B.seq_format != HP_SEQ_ASCII && B.seq_format != HP_SEQ_BAM4) { set_error("block %zu: unknown seq_format %u", b, B.seq_format); return fail(t, HP_ERR_ARG); }
                for (uint32_t i = 0; i < B.n_hets; ++i)
                    if (B.het_types[i] > 10) { set_error("block %zu: invalid variant type", b); return fail(t, HP_ERR_ARG); }
                std::vector<RecMeta>& meta = bs->meta[b];
                meta.assign(B.n_records, RecMeta{});
                bool hs = true;
                for (uint32_t i = 1; i < B.n_hets; ++i) hs = hs && B.hets[i - 1].position <= B.hets[i].position;
                uint32_t nj = 0;
                for (uint32_t r = 0; r < B.n_records; ++r) {
                    const hp_block_record& rec = B.records[r];
                    if (rec.qname_id >= B.n_qnames) { set_error("block %zu record %u: qname_id out of range", b, r); return fail(t, HP_ERR_ARG); }
                    if (rec.max_position < rec.min_position) { set_error("block %zu record %u: assert!(max_position >= min_position) (read_parsing.rs:685)", b, r); return fail(t, HP_ERR_INVARIANT); }
                    if (!p->global_realignment) continue;
                    uint32_t f = 0, l = 0;
                    bool contig = true;
                    if (!overlap_range(B.hets, B.n_hets, hs, rec.min_position, rec.max_position, f, l, contig)) continue;   // :703-712: skipped
                    if (!contig) { set_error("block %zu: assert_eq!(num_overlaps, last_overlap - first_overlap) (read_parsing.rs:715)", b); return fail(t, HP_ERR_INVARIANT); }
                    if (rec.min_position < (int64_t)B.ref_base) { set_error("block %zu record %u: alignment starts before the reference buffer", b, r); return fail(t, HP_ERR_ARG); }
                    meta[r].job = (int64_t)nj++; meta[r].first = f; meta[r].last = l;   // (job: block-local until the offsets are known)
                }
                njobs[b] = nj;
            }
        });
        if (first_rc.load() != HP_OK) {
            for (auto& e : errs) if (!e.empty()) { set_error("%s", e.c_str()); break; }
            return first_rc.load();
        }
    }
    for (size_t b = 0; b < n_blocks; ++b) bs->job_first[b + 1] = bs->job_first[b] + njobs[b];
    const size_t n_jobs = bs->job_first[n_blocks];
    bs->jobs.resize(n_jobs);
    bs->job_alloff.resize(n_jobs + 1);
    {   // the jobs (host threads over blocks): the hom overlaps of a record are only looked up for records that have het overlaps
        std::atomic<size_t> next{0};
        unsigned nt = host_threads(16u);
        if (const char* e = std::getenv("HP_BLOCK_HOST_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
        nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, n_blocks / 4));
        WorkerPool::get().run(nt, [&](unsigned) {
            for (;;) {
                const size_t b = next.fetch_add(1);
                if (b >= n_blocks) return;
                const hp_block_input& B = in[b];
                bool ms = true;
                for (uint32_t i = 1; i < B.n_homs; ++i) ms = ms && B.homs[i - 1].position <= B.homs[i].position;
                const uint32_t j0 = bs->job_first[b];
                for (uint32_t r = 0; r < B.n_records; ++r) {
                    RecMeta& m = bs->meta[b][r];
                    if (m.job < 0) continue;
                    const hp_block_record& rec = B.records[r];
                    uint32_t hf = 0, hl = 0;
                    bool contig = true;
                    const bool homs = overlap_range(B.homs, B.n_homs, ms, rec.min_position, rec.max_position, hf, hl, contig);
                    m.job += j0;
                    W2JobIn& j = bs->jobs[(size_t)m.job];
                    j.block = (uint32_t)b; j.rec = r;
                    j.het_first = m.first; j.n_hets = m.last - m.first;
                    j.hom_first = homs ? hf : 0; j.n_homs = homs ? hl - hf : 0;   // first_hom_overlap.unwrap_or(0) with an empty range
                }
            }
        });
    }
    uint64_t al_total = 0;
    for (size_t k = 0; k < n_jobs; ++k) { bs->job_alloff[k] = al_total; al_total += bs->jobs[k].n_hets; }
    bs->job_alloff[n_jobs] = al_total;
    bs->alleles.assign((size_t)al_total + 1, (uint8_t)HP_ALLELE_NOOVERLAP);
    bs->allele_ptrs.resize(n_jobs);
    for (size_t k = 0; k < n_jobs; ++k) bs->allele_ptrs[k] = bs->alleles.data() + bs->job_alloff[k];
    bs->wfa_out.resize(n_jobs);
    bs->upload_min_jobs = min_jobs;
    // large sets (the compact road): the host-only half of the sequence layout - offsets, the runs the copy engines read in place,
    // the length order - belongs to this stage too: in front of the upload it was 4-5 ms of every set's turn on the PCIe link
    bs->wfa_laid = false;
    if (n_jobs && n_jobs >= min_jobs) {
        if (!bs->wfa) bs->wfa = w2_session_create();
        const int rc = w2_session_layout_blocks(bs->wfa, in, n_blocks, bs->jobs.data(), n_jobs);
        if (rc != HP_OK) return rc;
        bs->wfa_laid = true;
    }
    bs->prep[0] = blk_now_ms() - t0;
    return HP_OK;
}

// second half of the first stage: the sequences laid out and uploaded (resident) - staging copy + PCIe, then they stay in HBM
int hp::blockset_upload(hp_blockset* bs) {
    const double t0 = blk_now_ms();
    const size_t n_jobs = bs->jobs.size(), min_jobs = bs->upload_min_jobs, n_blocks = bs->n_blocks;
    const hp_block_input* in = bs->in;
    // large sets: lay the sequences out and upload them now (resident); small ones take the latency path at solve time
    if (n_jobs && n_jobs >= min_jobs) {
        int rc;
        if (bs->wfa_laid) rc = w2_session_upload_blocks(bs->wfa, bs->device);
        else {
            if (!bs->wfa) bs->wfa = w2_session_create();
            rc = w2_session_prepare_blocks(bs->wfa, in, n_blocks, bs->jobs.data(), n_jobs, bs->device);
        }
        bs->wfa_laid = false;
        if (rc != HP_OK) return rc;
        bs->wfa_ready = true;
        double pr[4];
        w2_session_prepare_stats(bs->wfa, pr);
        bs->prep[1] = pr[1]; bs->prep[3] = pr[3];   // (the layout half's time is in prep[0] already: blockset_layout)
    }
    bs->prep[2] = bs->prep[0] + (blk_now_ms() - t0);
    return HP_OK;
}

int hp::blockset_init(hp_blockset* bs, size_t n_blocks, const hp_block_input* in, const hp_block_params* p, int device_id) {
    const int rc = blockset_layout(bs, n_blocks, in, p, device_id);
    return rc != HP_OK ? rc : blockset_upload(bs);
}

namespace {

// load_full_read_segments' tail for one block (read_parsing.rs:546-629) once every record's WFA outcome is known:
// fallback to local re-alignment, the global_disabled switch in BAM order, qualities, ReadSegment::new, collapse, split
// the records of block b whose local re-alignment is known to be needed before the replay: every record in local mode
// (read_parsing.rs:47-113), the records whose graph-WFA ran into max_edit_distance otherwise (:564-575). Resets the block's state.
int local_needs(hp_blockset* bs, size_t b) {
    const hp_block_input& B = bs->in[b];
    BlockState& S = bs->st[b];
    S.reset();
    const uint32_t R = B.n_records;
    S.local_slot.assign(R, -1);
    if (!bs->prm.global_realignment) { S.loc_need.resize(R); for (uint32_t i = 0; i < R; ++i) S.loc_need[i] = i; }
    else {
        const std::vector<RecMeta>& meta = bs->meta[b];
        for (uint32_t i = 0; i < R; ++i) {
            if (meta[i].job < 0) continue;
            const int32_t st = bs->wfa_out[(size_t)meta[i].job].status;
            if (st == HP_WFA_MAX_ED) S.loc_need.push_back(i);
            else if (st == HP_WFA_UNSUPPORTED) S.wfa_unsupported = true;   // (the reference has no such limit: the caller solves this block itself)
        }
        if (S.wfa_unsupported) { S.loc_need.clear(); return HP_OK; }
    }
    if (S.loc_need.empty()) return HP_OK;
    if (!B.local_hets) { set_error("block %zu: a record needs local re-alignment (read_parsing.rs:121-503) but local_hets is NULL", b); return HP_ERR_ARG; }
    S.loc_reads.resize(S.loc_need.size());
    for (size_t k = 0; k < S.loc_need.size(); ++k) {
        if (!B.records[S.loc_need[k]].local) { set_error("block %zu record %u needs local re-alignment but has no CIGAR view", b, S.loc_need[k]); return HP_ERR_ARG; }
        S.loc_reads[k] = *B.records[S.loc_need[k]].local;
    }
    const size_t N = B.n_hets;
    S.loc_alleles.resize(S.loc_need.size() * N); S.loc_quals.resize(S.loc_need.size() * N); S.loc_stats.resize(S.loc_need.size());
    for (size_t k = 0; k < S.loc_need.size(); ++k) S.local_slot[S.loc_need[k]] = (int64_t)k;
    return HP_OK;
}

// local re-alignment for the listed blocks' needs: ONE device launch over all of them (a block with a handful of fallbacks used
// to make a launch of its own - a few hundred small launches per set, each waiting its turn on the device)
int local_prepass(hp_blockset* bs, const std::vector<size_t>& blocks) {
    std::vector<LocalGroup> groups;
    for (size_t b : blocks) {
        BlockState& S = bs->st[b];
        if (S.loc_need.empty()) continue;
        groups.push_back(LocalGroup{S.loc_reads.data(), S.loc_reads.size(), bs->in[b].local_hets, bs->in[b].n_hets, S.loc_alleles.data(), S.loc_quals.data(), S.loc_stats.data()});
    }
    return groups.empty() ? HP_OK : local_realign_groups(groups.data(), groups.size(), bs->device);
}

int assemble_block(hp_blockset* bs, size_t b) {
    const hp_blockset& CH = *bs;
    const hp_block_input& B = bs->in[b];
    const hp_block_params& P = bs->prm;
    BlockState& S = bs->st[b];
    const uint32_t N = B.n_hets, R = B.n_records;
    const std::vector<RecMeta>& meta = bs->meta[b];
    if (S.wfa_unsupported) {   // nothing is assembled: an empty matrix stands in, the block is left out of the A* batch (blockset_rows)
        S.var_flags.assign(N, 0); S.row_off.assign(1, 0); S.alleles_2bit.assign(1, 0); S.quals.assign(1, 0);
        return HP_OK;
    }
    // local re-alignment rows: the pre-pass (local_needs + local_prepass) has the ones known up front; the `global_disabled`
    // switch (below) may ask for more (a record's local result does not depend on any other record)
    std::vector<int64_t>& local_slot = S.local_slot;
    std::vector<uint8_t>& loc_alleles = S.loc_alleles;
    std::vector<uint8_t>& loc_quals = S.loc_quals;
    std::vector<hp_read_stats>& loc_stats = S.loc_stats;
    auto solve_local = [&](const std::vector<uint32_t>& idx) -> int {
        std::vector<uint32_t> need;
        for (uint32_t i : idx) if (local_slot[i] < 0) need.push_back(i);
        if (need.empty()) return HP_OK;
        if (!B.local_hets) { set_error("block %zu: a record needs local re-alignment (read_parsing.rs:121-503) but local_hets is NULL", b); return HP_ERR_ARG; }
        std::vector<hp_local_read> reads(need.size());
        for (size_t k = 0; k < need.size(); ++k) {
            if (!B.records[need[k]].local) { set_error("block %zu record %u needs local re-alignment but has no CIGAR view", b, need[k]); return HP_ERR_ARG; }
            reads[k] = *B.records[need[k]].local;
        }
        const size_t base = loc_stats.size();
        loc_alleles.resize((base + need.size()) * (size_t)N);
        loc_quals.resize((base + need.size()) * (size_t)N);
        loc_stats.resize(base + need.size());
        const int rc = hp_local_realign_batch(reads.data(), reads.size(), B.local_hets, N, loc_alleles.data() + base * N, loc_quals.data() + base * N,
                                              loc_stats.data() + base, bs->device);
        if (rc != HP_OK) return rc;
        for (size_t k = 0; k < need.size(); ++k) local_slot[need[k]] = (int64_t)(base + k);
        return HP_OK;
    };
    int rc;
    // per read name: the segments of its records, in BAM order
    // (a linked list per read name through the records' own segments: almost every read name has one record, and a
    // vector per name would cost an allocation per read)
    std::vector<Segment> recseg(R);
    std::vector<uint32_t> qfirst(B.n_qnames, UINT32_MAX), qlast(B.n_qnames, UINT32_MAX), qcount(B.n_qnames, 0), next_rec(R, UINT32_MAX);
    std::vector<uint32_t> order;           // first-seen read names
    order.reserve(B.n_qnames);
    std::vector<uint8_t> seen(B.n_qnames, 0);
    bool global_disabled = false;
    double num_global_failures = 0.0, total_parsed = 0.0;
    for (uint32_t idx = 0; idx < R; ++idx) {
        const RecMeta& m = meta[idx];
        Segment seg;
        uint64_t wfa_score = 0;
        bool skipped = false;
        double local_aligned = 0.0;
        if (!P.global_realignment) {
            const size_t s = (size_t)local_slot[idx];
            skipped = loc_stats[s].skipped_reads == 1;
            add_read_stats(S.rs, loc_stats[s]);   // read_parsing.rs:88 (skipped or not)
            if (!skipped) seg = segment_new(S.arena, loc_alleles.data() + s * N, loc_quals.data() + s * N, 0, N, N);
            local_aligned = 1.0;
        } else {
            if (m.job < 0) { S.skipped_reads += 1; continue; }   // no overlaps: flagged skipped (read_parsing.rs:703-712, :602-605)
            const hp_wfa_result& w = CH.wfa_out[(size_t)m.job];
            if (global_disabled || w.status == HP_WFA_MAX_ED) {
                if (local_slot[idx] < 0) {   // the switch just flipped: everything from here on is local (read_parsing.rs:556-559)
                    std::vector<uint32_t> rest;
                    for (uint32_t i = idx; i < R; ++i) if (meta[i].job >= 0) rest.push_back(i);
                    if ((rc = solve_local(rest)) != HP_OK) return rc;
                }
                const size_t s = (size_t)local_slot[idx];
                skipped = loc_stats[s].skipped_reads == 1;
                add_read_stats(S.rs, loc_stats[s]);   // read_parsing.rs:607 with local_realignment's stats
                if (!skipped) seg = segment_new(S.arena, loc_alleles.data() + s * N, loc_quals.data() + s * N, 0, N, N);
                wfa_score = P.max_edit_distance;   // :559 / :573: the distance carried by the error is max_edit_distance
                local_aligned = 1.0;
            } else {
                const uint32_t n = m.last - m.first;
                const uint8_t* a = CH.alleles.data() + CH.job_alloff[(size_t)m.job];
                // ReadSegment::new on the window [first, last): clip to the set alleles, then read_parsing.rs:803-835 for the
                // qualities of what is left (2 x base quality for 0/1 alleles, else 0)
                uint32_t f0 = n, l0 = n;
                // the record's ReadStats (read_parsing.rs:805-850): per het of the overlap range - Ambiguous: failed; 0 / 1: inexact
                // (`exact_allele` is false upstream) + allele0 / allele1 + num_alleles
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t vt = B.het_types[m.first + i];
                    if (a[i] == HP_ALLELE_AMBIGUOUS) S.rs.failed_matches[vt] += 1;
                    else if (a[i] < HP_ALLELE_AMBIGUOUS) {
                        S.rs.inexact_matches[vt] += 1;
                        if (a[i] == HP_ALLELE_REFERENCE) S.rs.allele0_matches[vt] += 1; else S.rs.allele1_matches[vt] += 1;
                        S.rs.num_alleles += 1;
                    }
                }
                for (uint32_t i = 0; i < n; ++i) if (a[i] < HP_ALLELE_AMBIGUOUS) { f0 = i; break; }
                for (uint32_t i = n; i-- > 0;) if (a[i] < HP_ALLELE_AMBIGUOUS) { l0 = i + 1; break; }
                if (f0 == n) { seg.start = seg.end = N; }
                else {
                    seg.start = m.first + f0; seg.end = m.first + l0;
                    uint8_t* sa = S.arena.get(l0 - f0);
                    uint8_t* sq = S.arena.get(l0 - f0);
                    std::memcpy(sa, a + f0, l0 - f0);
                    for (uint32_t i = f0; i < l0; ++i) {
                        sq[i - f0] = 0;
                        if (a[i] < HP_ALLELE_AMBIGUOUS) {
                            const int q = base_quality(B.het_types[m.first + i]);
                            if (q < 0) { set_error("block %zu: no base quality for variant type %u (read_parsing.rs:829 panics)", b, (unsigned)B.het_types[m.first + i]); return HP_ERR_INVARIANT; }
                            sq[i - f0] = (uint8_t)(2 * q);
                        }
                    }
                    seg.alleles = sa; seg.quals = sq;
                }
                // (a het outside [f0, l0) is NoOverlap / Ambiguous with quality 0 either way; the reference panics on an
                // unknown type only for a 0/1 allele, :803-829, which all lie inside the clip)
                wfa_score = w.score;
            }
        }
        if (skipped) { S.skipped_reads += 1; continue; }
        S.local_aligned += (uint64_t)local_aligned;
        S.global_aligned += 1 - (uint64_t)local_aligned;
        const uint32_t q = B.records[idx].qname_id;
        if (!seen[q]) { seen[q] = 1; order.push_back(q); qfirst[q] = idx; } else next_rec[qlast[q]] = idx;
        qlast[q] = idx; qcount[q]++;
        recseg[idx] = seg;
        if (P.global_realignment) {
            S.edit_distances.push_back(wfa_score);
            num_global_failures += local_aligned;
            total_parsed += 1.0;
            if (!global_disabled && num_global_failures >= (double)P.global_failure_minimum && num_global_failures / total_parsed >= P.global_failure_ratio)
                global_disabled = true;   // read_parsing.rs:597-600
        }
    }
    // collapse per read name + the min_matched_alleles split (read_parsing.rs:611-629 / :95-113)
    S.segs.reserve(order.size()); S.seg_qname.reserve(order.size()); S.seg_solver.reserve(order.size()); S.solver_rows.reserve(order.size());
    std::vector<const Segment*> grp;
    for (uint32_t q : order) {
        Segment col;
        if (qcount[q] == 1) col = recseg[qfirst[q]];   // collapse of one segment is that segment (read_segments.rs:72-75)
        else {
            grp.clear();
            for (uint32_t i = qfirst[q]; i != UINT32_MAX; i = next_rec[i]) grp.push_back(&recseg[i]);
            if (!segment_collapse(S.arena, grp, N, col)) { set_error("block %zu: assert!(quals[i] > 0) (read_segments.rs:105)", b); return HP_ERR_INVARIANT; }
        }
        const uint32_t num_set = seg_num_set(col);
        const bool solver = num_set >= P.min_matched_alleles;
        if (solver) S.num_reads += qcount[q]; else S.skipped_reads += qcount[q];
        if (!solver && num_set == 0) continue;
        if (solver) S.solver_rows.push_back((uint32_t)S.segs.size());
        S.segs.push_back(col);
        S.seg_qname.push_back(q);
        S.seg_solver.push_back(solver ? 1 : 0);
    }
    // the solver matrix (phaser.rs:514-533) as hp_block_view
    S.var_flags.resize(N);
    for (uint32_t i = 0; i < N; ++i)
        S.var_flags[i] = (uint8_t)(((B.hets[i].flags & 1u) ? HP_VAR_IGNORED : 0) | (B.het_types[i] == 0 ? HP_VAR_SNV : 0));
    S.row_off.push_back(0);
    uint64_t cells = 0;
    for (uint32_t k : S.solver_rows) cells += S.segs[k].end - S.segs[k].start;
    S.alleles_2bit.assign((size_t)(cells + 3) / 4 + 1, 0);
    S.quals.reserve((size_t)cells + 1);
    uint64_t c = 0;
    for (uint32_t k : S.solver_rows) {
        const Segment& s = S.segs[k];
        S.read_start.push_back(s.start);
        S.read_end.push_back(s.end);
        const uint32_t len = s.end - s.start;
        for (uint32_t i = 0; i < len; ++i, ++c) S.alleles_2bit[(size_t)(c >> 2)] |= (uint8_t)(s.alleles[i] << (2 * (c & 3)));
        S.quals.insert(S.quals.end(), s.quals, s.quals + len);
        S.row_off.push_back(c);
    }
    if (S.quals.empty()) S.quals.push_back(0);
    return HP_OK;
}

}  // namespace

extern "C" hp_blockset* hp_blockset_create(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, int device_id, int* status) {
    auto bs = std::unique_ptr<hp_blockset>(new hp_blockset());
    const int rc = blockset_init(bs.get(), n_blocks, in, p, device_id);
    if (status) *status = rc;
    if (rc != HP_OK) return nullptr;
    return bs.release();
}

extern "C" void hp_blockset_destroy(hp_blockset* bs) { delete bs; }

// graph-WFA for every record with overlaps (one device batch); runs on the calling thread
int hp::blockset_wfa(hp_blockset* bs) {
    hp_blockset& ch = *bs;
    const double t0 = blk_now_ms();
    ch.ms[6] = 0.0;
    if (!ch.jobs.empty()) {
        int rc;
        // (the few reads the compact kernel hands back are still in the dense-band pass when this returns: blockset_rows)
        if (ch.wfa_ready) rc = w2_session_run(ch.wfa, bs->prm.wfa_prune_distance, bs->prm.max_edit_distance, ch.wfa_out.data(), ch.allele_ptrs.data(), 2);
        else {
            // a small set takes the latency path (dense-band kernel, one wavefront per read): the jobs as hp_wfa_assign_batch takes
            // them, BAM 4-bit reads decoded on the host (a few thousand reads at most)
            std::vector<hp_wfa_job>& jobs = ch.small_jobs;
            std::vector<std::vector<uint8_t>>& ascii = ch.small_ascii;
            jobs.assign(ch.jobs.size(), hp_wfa_job{});
            ascii.clear();
            for (size_t k = 0; k < ch.jobs.size(); ++k) {
                const W2JobIn& ji = ch.jobs[k];
                const hp_block_input& B = bs->in[ji.block];
                const hp_block_record& rec = B.records[ji.rec];
                hp_wfa_job& j = jobs[k];
                j.reference = B.reference; j.ref_base = B.ref_base;
                j.ref_start = (uint64_t)rec.min_position; j.ref_end = (uint64_t)rec.max_position + 1;   // read_parsing.rs:772-773
                j.hets = B.hets + ji.het_first; j.n_hets = ji.n_hets;
                j.homs = ji.n_homs ? B.homs + ji.hom_first : nullptr; j.n_homs = ji.n_homs;
                j.read_len = rec.read_len;
                if (B.seq_format == HP_SEQ_BAM4) {
                    ascii.emplace_back((size_t)rec.read_len + 1);
                    decode_bam4(rec.read_align, rec.read_offset, rec.read_len, ascii.back().data());
                    j.read = ascii.back().data();
                } else j.read = rec.read_align + rec.read_offset;
            }
            // (straight to the dense-band implementation: the public entry would queue behind the call combiner)
            if (ch.small_async) {   // inside a pipeline: on the slot's helper thread, joined by the rows stage (hp_block.h)
                if (!ch.small_helper) { ch.small_helper.reset(new HelperThread()); ch.small_helper->start(); }
                ch.small_rc = HP_OK; ch.small_err.clear(); ch.small_kernel_ms = 0.0;
                ch.small_inflight = true;
                hp_blockset* self = bs;
                const int part = g_cu_partition;
                ch.small_helper->post([self, part]() {
                    g_cu_partition = part == 2 ? 0 : part;
                    self->small_rc = wfa_assign_batch_v1(self->small_jobs.data(), self->small_jobs.size(), self->prm.wfa_prune_distance, self->prm.max_edit_distance,
                                                         self->wfa_out.data(), self->allele_ptrs.data(), self->device);
                    if (self->small_rc != HP_OK) self->small_err = hp_last_error();
                    self->small_kernel_ms = g_last_kernel_ms;
                });
                rc = HP_OK;
            } else
                rc = wfa_assign_batch_v1(jobs.data(), jobs.size(), bs->prm.wfa_prune_distance, bs->prm.max_edit_distance, ch.wfa_out.data(),
                                         ch.allele_ptrs.data(), bs->device);
        }
        if (rc != HP_OK) return rc;
        // the three class instantiations of hp_wfa2_kernel run concurrently: their span is the kernel time of the stage
        ch.ms[6] = ch.wfa_ready ? 0.0 : g_last_kernel_ms;   // (resident session: known once its second collection is done, blockset_rows)
    }
    ch.ms[0] = blk_now_ms() - t0;
    return HP_OK;
}

// after the WFA: fallback / replay / rows / collapse (host threads over blocks). May run on another thread than blockset_wfa did.
int hp::blockset_rows(hp_blockset* bs) {
    hp_blockset& ch = *bs;
    const bool has_wfa = ch.wfa_ready;
    const double t1 = blk_now_ms();
    int rc = HP_OK;
    // the deferred late pass of the set's alignment stage (helper thread of the session) writes ch.wfa_out / ch.alleles and reads
    // the caller's inputs: it is joined on EVERY way out of this function - an error return must not leave it running under a
    // slot that is recycled, a merged set that is re-run request by request, or inputs the caller frees after the error
    struct LateJoin { hp_blockset* s; bool on; ~LateJoin() { if (on && s->wfa) (void)w2_session_finish(s->wfa); } } late_join{bs, has_wfa};
    if ((rc = ch.small_join()) != HP_OK) return rc;   // (a small set's alignment ran on the slot's helper thread: blockset_wfa)
    {
        unsigned nt = host_threads(32u);   // measured: 16 -> 32 threads 6.0 -> 4.1 ms, 64 no better
        if (const char* e = std::getenv("HP_BLOCK_HOST_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
        nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, bs->n_blocks));
        std::vector<size_t> order(bs->n_blocks);
        for (size_t b = 0; b < bs->n_blocks; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return bs->in[x].n_records > bs->in[y].n_records; });
        // blocks that hold a read whose alignment is still in the dense-band pass go last; whoever reaches the first of
        // them waits for that pass (it has had the other blocks' assembly to finish in)
        size_t n_free = order.size();
        // the set's alignment stage did not wait for its own results (w2_session_run, defer = 2): they are handed over here
        if (has_wfa && (rc = w2_session_collected(ch.wfa)) != HP_OK) return rc;
        ch.rows_ms[4] = blk_now_ms() - t1;
        if (has_wfa) {
            const uint32_t* ids = nullptr; size_t n_ids = 0;
            w2_session_pending(ch.wfa, &ids, &n_ids);
            if (n_ids) {
                std::vector<uint8_t> held(bs->n_blocks, 0);
                for (size_t k = 0; k < n_ids; ++k) held[ch.jobs[ids[k]].block] = 1;
                std::stable_partition(order.begin(), order.end(), [&](size_t b) { return !held[b]; });
                n_free = 0;
                while (n_free < order.size() && !held[order[n_free]]) ++n_free;
            }
        }
        const bool dbg = std::getenv("HP_DEBUG") != nullptr;
        std::vector<std::string> errs(std::max(1u, nt));
        std::atomic<int> first_rc{HP_OK};
        // one phase = the blocks order[lo, hi): which of their records need local re-alignment (host threads), ONE local
        // re-alignment launch for all of them, then replay / qualities / collapse / solver rows per block (host threads)
        auto phase = [&](size_t lo, size_t hi) -> int {
            if (lo >= hi) return HP_OK;
            const double ta = blk_now_ms();
            for (int pass = 0; pass < 2; ++pass) {
                std::atomic<size_t> next{lo};
                WorkerPool::get().run(std::max(1u, (unsigned)std::min<size_t>(nt, hi - lo)), [&](unsigned t) {
                    for (;;) {
                        const size_t k = next.fetch_add(1);
                        if (k >= hi || first_rc.load() != HP_OK) return;
                        const int r = pass == 0 ? local_needs(bs, order[k]) : assemble_block(bs, order[k]);
                        if (r != HP_OK) { int exp = HP_OK; if (first_rc.compare_exchange_strong(exp, r)) errs[t] = hp_last_error(); return; }
                    }
                });
                if (first_rc.load() != HP_OK) {
                    for (auto& e : errs) if (!e.empty()) { set_error("%s", e.c_str()); break; }
                    return first_rc.load();
                }
                if (pass == 0) {
                    const std::vector<size_t> blocks(order.begin() + (ptrdiff_t)lo, order.begin() + (ptrdiff_t)hi);
                    const double tl = blk_now_ms();
                    const int r = local_prepass(bs, blocks);
                    ch.rows_ms[lo == 0 ? 3 : 2] = blk_now_ms() - tl;
                    if (r != HP_OK) return r;
                }
            }
            if (dbg) fprintf(stderr, "[hp] rows: %zu blocks on %u threads in %.2f ms\n", hi - lo, nt, blk_now_ms() - ta);
            return HP_OK;
        };
        // the blocks that wait for nothing first; then the ones that hold a read the second collection / the dense-band pass
        // delivers (that pass has had the first phase to finish in)
        ch.rows_ms[2] = ch.rows_ms[3] = 0.0;
        if ((rc = phase(0, n_free)) != HP_OK) return rc;
        const double tw = blk_now_ms();
        ch.rows_ms[0] = tw - t1;
        if (has_wfa && (rc = w2_session_finish(ch.wfa)) != HP_OK) return rc;
        ch.late_wait_ms = blk_now_ms() - tw;
        const double th = blk_now_ms();
        if ((rc = phase(n_free, order.size())) != HP_OK) return rc;
        ch.rows_ms[1] = blk_now_ms() - th;
        if (has_wfa) ch.ms[6] = w2_session_span_ms(ch.wfa);
    }
    ch.ms[1] = blk_now_ms() - t1;
    return HP_OK;
}

// after the rows: the A* batch of all blocks, packed (host threads) and uploaded
int hp::blockset_pack(hp_blockset* bs) {
    hp_blockset& ch = *bs;
    if (ch.batch) { hp_batch_destroy(ch.batch); ch.batch = nullptr; }
    const double t2 = blk_now_ms();
    // ---- A* over the set's blocks ----
    const size_t nb = bs->n_blocks;
    std::vector<hp_block_view>& views = ch.views;
    views.assign(nb, hp_block_view{});
    for (size_t k = 0; k < nb; ++k) {
        const size_t b = k;
        const BlockState& S = bs->st[b];
        hp_block_view v{};
        v.n_variants = bs->in[b].n_hets;
        v.n_reads = (uint32_t)S.read_start.size();
        v.read_start = S.read_start.data(); v.read_end = S.read_end.data(); v.row_off = S.row_off.data();
        v.alleles_2bit = S.alleles_2bit.data(); v.quals = S.quals.data(); v.var_flags = S.var_flags.data();
        views[k] = v;
    }
    hp_astar_params ap = bs->prm.astar;
    int st = HP_OK;
    // kept[j] = index (into views) of the j-th block of the A* batch. A block outside the solver's packed-key
    // limits must not fail the others - nor the call when it is alone: it is left out of the batch and handed back
    // with the soft status HP_BLOCK_UNSUPPORTED (segments filled; h1 / h2 / stats / spans / tags untouched), whether it came
    // alone, with others, or merged with other callers' blocks
    std::vector<size_t>& kept = ch.kept;
    kept.resize(nb);
    for (size_t k = 0; k < nb; ++k) kept[k] = k;
    std::vector<char>& unsupported = ch.unsupported;
    unsupported.assign(nb, 0);
    bool any_wfa_unsupported = false;
    for (size_t k = 0; k < nb; ++k) if (bs->st[k].wfa_unsupported) { unsupported[k] = 1; any_wfa_unsupported = true; }
    hp_batch* batch = any_wfa_unsupported ? nullptr : hp_batch_create(nb, views.data(), &ap, bs->device, &st);
    if (any_wfa_unsupported) st = HP_ERR_UNSUPPORTED;   // (the same road as a block beyond the solver's limits: the others are solved)
    if (!batch && st == HP_ERR_UNSUPPORTED) {
        kept.clear();
        for (size_t k = 0; k < nb; ++k) {   // a create on its own tells which
            if (unsupported[k]) continue;
            int s1 = HP_OK;
            hp_batch* one = hp_batch_create(1, &views[k], &ap, bs->device, &s1);
            if (one) { hp_batch_destroy(one); kept.push_back(k); continue; }
            if (s1 != HP_ERR_UNSUPPORTED) return s1;
            unsupported[k] = 1;
        }
        if (!kept.empty() && kept.size() < nb) {
            std::vector<hp_block_view> kv(kept.size());
            for (size_t j = 0; j < kept.size(); ++j) kv[j] = views[kept[j]];
            batch = hp_batch_create(kept.size(), kv.data(), &ap, bs->device, &st);
            if (!batch) return st != HP_OK ? st : HP_ERR_HIP;
        } else if (!kept.empty()) return HP_ERR_UNSUPPORTED;   // every block packs alone but not together: the batch limits (split the call)
    } else if (!batch) return st != HP_OK ? st : HP_ERR_HIP;
    ch.batch = batch;
    const double t3 = blk_now_ms();
    ch.ms[2] = t3 - t2;
    return HP_OK;
}

// after the WFA, second half: A* over the packed batch, span counts and haplotags on the resident matrix, outputs into the
// caller's buffers. May run on yet another thread (the fourth stage of a block stream).
int hp::blockset_solve(hp_blockset* bs, hp_block_output* out) {
    hp_blockset& ch = *bs;
    const bool has_wfa = ch.wfa_ready;
    const size_t nb = bs->n_blocks;
    std::vector<hp_block_view>& views = ch.views;
    std::vector<size_t>& kept = ch.kept;
    std::vector<char>& unsupported = ch.unsupported;
    hp_batch* batch = ch.batch;
    struct BatchGuard { hp_blockset* s; ~BatchGuard() { if (s->batch) { hp_batch_destroy(s->batch); s->batch = nullptr; } } } guard{bs};
    int rc = HP_OK;
    const size_t nk = kept.size();
    const double t3 = blk_now_ms();
    float kms = 0.f;
    uint64_t sum_n = 0, sum_rows = 0, sum_j = 0;
    for (size_t j = 0; j < nk; ++j) { const size_t k = kept[j]; sum_n += bs->in[k].n_hets; sum_rows += views[k].n_reads; sum_j += bs->in[k].n_hets - 1; }
    std::vector<uint8_t> h1((size_t)sum_n + 1), h2((size_t)sum_n + 1);
    std::vector<hp_phase_stats> stats(nk + 1);
    std::vector<hp_work_counters> ctr(nk);
    if (batch) {
        if ((rc = hp_batch_solve(batch, nullptr, &kms)) != HP_OK) return rc;
        if ((rc = hp_batch_results(batch, h1.data(), h2.data(), stats.data(), ctr.data(), nullptr)) != HP_OK) return rc;
    }
    for (int i = 0; i < 8; ++i) ch.work[i] = 0;
    if (has_wfa) w2_session_work(ch.wfa, ch.work);
    for (auto& c : ctr) { ch.work[4] += c.cells; ch.work[5] += c.evals; }
    ch.work[6] = sum_n; ch.work[7] = sum_rows;
    const double t4 = blk_now_ms();
    // ---- span counts and haplotags on the resident matrix ----
    std::vector<uint64_t> spans((size_t)sum_j + 1);
    std::vector<uint8_t> tag((size_t)sum_rows + 1);
    std::vector<uint32_t> fh((size_t)sum_rows + 1);
    if (batch && (rc = hp_batch_postprocess(batch, spans.data(), tag.data(), fh.data())) != HP_OK) return rc;
    // ---- outputs (blocks are independent: host threads over blocks) ----
    std::vector<uint64_t> on_of(nb + 1, 0), orow_of(nb + 1, 0), oj_of(nb + 1, 0), slot_of(nb, 0);   // offsets in the batch's results
    {
        uint64_t on = 0, orow = 0, oj = 0;
        for (size_t j = 0; j < nk; ++j) {
            const size_t kb = kept[j];
            const uint32_t N = bs->in[kb].n_hets;
            on_of[kb] = on; orow_of[kb] = orow; oj_of[kb] = oj; slot_of[kb] = j;
            on += N; orow += views[kb].n_reads; oj += N - 1;
        }
    }
    std::atomic<int64_t> cap_fail{-1};
    auto emit_block = [&](size_t kb) {
        const uint64_t on = on_of[kb], orow = orow_of[kb], oj = oj_of[kb];
        const size_t b = kb;
        const hp_block_input& B = bs->in[b];
        const BlockState& S = bs->st[b];
        hp_block_output& O = out[b];
        const bool unsup = unsupported[kb] != 0;
        O.status = unsup ? HP_BLOCK_UNSUPPORTED : HP_OK;
        const uint32_t N = B.n_hets;
        const uint8_t* H1 = h1.data() + on;
        const uint8_t* H2 = h2.data() + on;
        if (!unsup) {   // (an unsupported block's h1 / h2 / stats / spans stay as the caller left them: hiphase_gpu.h)
            if (O.h1) std::memcpy(O.h1, H1, N);
            if (O.h2) std::memcpy(O.h2, H2, N);
            O.stats = stats[slot_of[kb]];
            if (O.span_counts && N > 1) std::memcpy(O.span_counts, spans.data() + oj, (size_t)(N - 1) * 8);
        }
        O.n_segments = (uint32_t)S.segs.size();
        O.n_solver = (uint32_t)S.solver_rows.size();
        O.num_reads = S.num_reads; O.skipped_reads = S.skipped_reads; O.global_aligned = S.global_aligned; O.local_aligned = S.local_aligned;
        O.num_alleles = S.rs.num_alleles;
        for (int t = 0; t < HP_N_VARIANT_TYPES; ++t) {
            O.exact_matches[t] = S.rs.exact_matches[t]; O.inexact_matches[t] = S.rs.inexact_matches[t]; O.failed_matches[t] = S.rs.failed_matches[t];
            O.allele0_matches[t] = S.rs.allele0_matches[t]; O.allele1_matches[t] = S.rs.allele1_matches[t];
        }
        O.n_edit_distances = S.edit_distances.size();
        if (O.edit_distances && !S.edit_distances.empty()) std::memcpy(O.edit_distances, S.edit_distances.data(), S.edit_distances.size() * 8);
        uint64_t cells = 0;
        uint32_t srow = 0;
        for (size_t k = 0; k < S.segs.size(); ++k) {
            const Segment& s = S.segs[k];
            if (O.seg_qname) O.seg_qname[k] = S.seg_qname[k];
            if (O.seg_start) O.seg_start[k] = s.start;
            if (O.seg_end) O.seg_end[k] = s.end;
            if (O.seg_solver) O.seg_solver[k] = S.seg_solver[k];
            uint8_t ht = 2;
            uint32_t first = UINT32_MAX;
            if (unsup) {}
            else if (S.seg_solver[k]) { ht = tag[(size_t)(orow + srow)]; first = fh[(size_t)(orow + srow)]; ++srow; }
            else {
                // a segment outside the solver matrix (fewer than min_matched_alleles set alleles) is tagged against the
                // solution the way haplotag_reads tags any segment (phaser.rs:620-630, :714-750): a handful of cells
                uint64_t s1 = 0, s2 = 0;
                for (uint32_t i = s.start; i < s.end; ++i) {
                    const uint8_t a = s.alleles[i - s.start], q = s.quals[i - s.start];
                    if (H1[i] < HP_ALLELE_AMBIGUOUS && a != H1[i]) s1 += q;
                    if (H2[i] < HP_ALLELE_AMBIGUOUS && a != H2[i]) s2 += q;
                }
                if (s1 != s2) {
                    ht = s1 < s2 ? 0 : 1;
                    uint32_t f = s.start;
                    while (f < s.end && (H1[f] == H2[f] || s.alleles[f - s.start] >= HP_ALLELE_AMBIGUOUS)) ++f;
                    first = f < s.end ? f : UINT32_MAX;
                }
            }
            if (O.seg_haplotag) O.seg_haplotag[k] = ht;
            if (O.seg_first_het) O.seg_first_het[k] = first;
            if (O.seg_row_off) O.seg_row_off[k] = cells;
            const uint64_t len = s.end - s.start;
            if (O.seg_alleles || O.seg_quals) {
                if (cells + len > O.seg_cell_cap) { cap_fail.store((int64_t)b); return; }
                if (O.seg_alleles && len) std::memcpy(O.seg_alleles + cells, s.alleles, (size_t)len);
                if (O.seg_quals && len) std::memcpy(O.seg_quals + cells, s.quals, (size_t)len);
            }
            cells += len;
        }
        if (O.seg_row_off) O.seg_row_off[S.segs.size()] = cells;
    };
    {
        unsigned nt = host_threads(16u);
        if (const char* e = std::getenv("HP_BLOCK_HOST_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
        nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, nb / 8));
        std::atomic<size_t> next{0};
        WorkerPool::get().run(std::max(1u, nt), [&](unsigned) { for (;;) { const size_t kb = next.fetch_add(1); if (kb >= nb) return; emit_block(kb); } });
        if (cap_fail.load() >= 0) { set_error("block %lld: seg_cell_cap too small", (long long)cap_fail.load()); return HP_ERR_ARG; }
    }
    const double t5 = blk_now_ms();
    ch.ms[3] = t4 - t3; ch.ms[4] = t5 - t4; ch.ms[7] = kms;
    return HP_OK;
}

extern "C" int hp_blockset_solve(hp_blockset* bs, hp_block_output* out, double* stage_ms) {
    if (!bs || !out) { set_error("null argument"); return HP_ERR_ARG; }
    const double t0 = blk_now_ms();
    int rc = blockset_wfa(bs);
    if (rc == HP_OK) rc = blockset_rows(bs);
    if (rc == HP_OK) rc = blockset_pack(bs);
    if (rc == HP_OK) rc = blockset_solve(bs, out);
    if (rc != HP_OK) return rc;
    if (stage_ms) {
        for (int i = 0; i < 8; ++i) stage_ms[i] = bs->ms[i];
        stage_ms[5] = blk_now_ms() - t0;   // wall time of the call
    }
    return HP_OK;
}

extern "C" int hp_blockset_work(const hp_blockset* bs, uint64_t out[8]) {
    if (!bs || !out) { set_error("null argument"); return HP_ERR_ARG; }
    for (int i = 0; i < 8; ++i) out[i] = bs->work[i];
    return HP_OK;
}

// One set on one device, on the calling thread. The thread keeps its block-set object from call to call (host vectors,
// device buffers, the graph-WFA session and its helper thread): a service thread or a caller's worker solves set after set.
static int solve_blocks_on_device(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out, int device_id) {
    static thread_local std::unique_ptr<hp_blockset> tl_bs;
    if (!tl_bs) tl_bs.reset(new hp_blockset());
    hp_blockset* bs = tl_bs.get();
    int rc = blockset_init(bs, n_blocks, in, p, device_id);
    if (rc == HP_OK) rc = hp_blockset_solve(bs, out, nullptr);
    bs->in = nullptr;   // (nothing of the caller's is kept)
    return rc;
}

// ---- the node's GPUs behind the unchanged per-block entry -----------------------------------------------------------------------
// HiPhase calls solve_block once per phase block from its `--threads` pool (reference src/main.rs:326-462: a bounded queue of
// 40 x threads job slots, T workers, results written in order). With the one-call-site patch of INTEGRATION.md that is T threads
// sitting in hp_solve_blocks(1, ..., device_id = -1) at once; with the asynchronous entry (hp_block_submit / hp_block_wait) it is
// the reference's own 40 x T job slots in flight. Either way a request goes into ONE queue. Every visible device has a FEEDER
// thread and a COMPLETER thread around a six-stage block pipeline (hp_stream.hip, the road bench.py's headline takes): the feeder
// waits for a free slot of its pipeline - requests pile up meanwhile, which is all the batching there is: no window, no timer -
// takes its share of what is queued (everything for its own device + queued / devices of the common queue, a few hundred
// records at least), merges it into one block set and submits it; the completer waits for the sets in order and hands the
// results out. While set k is aligned, set k + 1 is laid out and crossing PCIe and set k - 1 is solved: the per-block call runs
// at the stream's rate, not at the serial stages-in-a-row rate of round 3. A call with many blocks (device_id = -1, several
// devices) is cut into LPT chunks of about total / (8 x devices) records that travel through the same queue (sizes are
// heavy-tailed). No collective, no device-to-device traffic: a block's 2 N result bytes + statistics go back over
// PCIe. The threads are started by the first call that needs them and live as long as the process.
namespace {

struct BlocksReq {
    BlocksReq(size_t n_, const hp_block_input* in_, const hp_block_params& prm_, hp_block_output* out_, int device_) : n(n_), in(in_), prm(prm_), out(out_), device(device_) {}
    BlocksReq(const BlocksReq&) = delete;
    BlocksReq& operator=(const BlocksReq&) = delete;
    size_t n; const hp_block_input* in; hp_block_params prm; hp_block_output* out; int device;   // device: -1 = any
    int rc = HP_OK; std::string err;
    uint64_t records = 0;
    // completion is signalled per request (round 5): with one condition variable for the dispatcher, every finished set woke every
    // waiter - 40 x T tickets + T blocking callers - to re-check its flag under the dispatcher's mutex
    std::mutex m; std::condition_variable cv; bool done = false;
    void finish() { std::lock_guard<std::mutex> lk(m); done = true; cv.notify_all(); }   // (notified under the lock: the waiter may free the request as soon as it holds it)
    void wait_done() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&]() { return done; }); }
};
bool same_params(const hp_block_params& a, const hp_block_params& b) {
    return a.astar.min_queue_size == b.astar.min_queue_size && a.astar.queue_increment == b.astar.queue_increment &&
           a.astar.max_segment_size == b.astar.max_segment_size && a.wfa_prune_distance == b.wfa_prune_distance &&
           a.max_edit_distance == b.max_edit_distance && a.global_failure_ratio == b.global_failure_ratio &&
           a.global_failure_minimum == b.global_failure_minimum && a.min_matched_alleles == b.min_matched_alleles &&
           a.global_realignment == b.global_realignment;
}

// a merged set on its way through a device's pipeline
struct MergedSet {
    std::vector<BlocksReq*> reqs;
    std::vector<hp_block_input> in;
    std::vector<hp_block_output> out;
    hp_block_params prm{};
    uint64_t ticket = 0;
    int submit_rc = HP_OK;
    std::string submit_err;
};

class BlockDispatcher {
public:
    static BlockDispatcher& get() { static auto* d = new BlockDispatcher(); return *d; }   // never destroyed
    int devices() { std::lock_guard<std::mutex> lk(m_); start_locked(); return n_vdev_; }
    int real_devices() { std::lock_guard<std::mutex> lk(m_); start_locked(); return real_dev_; }
    // queues the requests; returns at once (wait() collects them)
    void post(BlocksReq* const* reqs, size_t n) {
        std::unique_lock<std::mutex> lk(m_);
        start_locked();
        for (size_t i = 0; i < n; ++i) {
            BlocksReq* r = reqs[i];
            r->records = 1;
            for (size_t b = 0; b < r->n; ++b) r->records += r->in[b].n_records;
            if (r->device >= 0) dev_q_[(size_t)r->device].push_back(r); else any_q_.push_back(r);
        }
        cv_work_.notify_all();
    }
    void wait(BlocksReq* const* reqs, size_t n) { for (size_t i = 0; i < n; ++i) reqs[i]->wait_done(); }
    std::atomic<int> entering{0};     // callers inside the blocking one-block entry (a lone one runs on its own thread)

private:
    struct Dev {
        int device = 0;
        Pipeline* pipe = nullptr;
        std::mutex m;
        std::condition_variable cv;
        std::deque<std::unique_ptr<MergedSet>> inflight;   // submitted, in ticket order
    };
    void start_locked() {
        if (started_) return;
        started_ = true;
        real_dev_ = std::max(1, hp_device_count());
        // test hook: n queue "devices" on a box with fewer GPUs (virtual device v works on GPU v % real); every real device gets
        // its threads whatever the hook says (a request that names a device must find them)
        const char* wenv = std::getenv("HP_QUEUE_WORKERS");
        n_vdev_ = std::max(real_dev_, wenv ? std::max(1, std::atoi(wenv)) : real_dev_);
        dev_q_.resize((size_t)n_vdev_);
        dev_failed_.assign((size_t)n_vdev_, 0);
        devs_.resize((size_t)n_vdev_);
        for (int v = 0; v < n_vdev_; ++v) {
            devs_[(size_t)v].reset(new Dev());
            devs_[(size_t)v]->device = v % real_dev_;
            std::thread([this, v]() { name_thread("hp-feed"); feed(v); }).detach();
            std::thread([this, v]() { name_thread("hp-done"); complete(v); }).detach();
        }
    }
    static uint64_t max_records() {   // records per merged set (the bench's sets hold ~136 k; a pipeline slot's buffers grow to its largest set)
        static const uint64_t v = [] { const char* e = std::getenv("HP_DISPATCH_MAX_RECORDS"); return e ? (uint64_t)std::max(1ll, std::atoll(e)) : 160000ull; }();
        return v;
    }
    static uint64_t min_records() {   // no crumbs: a set of a dozen blocks costs a device the same stage latencies as one of a few hundred
        static const uint64_t v = [] { const char* e = std::getenv("HP_DISPATCH_MIN_RECORDS"); return e ? (uint64_t)std::max(1ll, std::atoll(e)) : 2048ull; }();
        return v;
    }
    void feed(int v) {
        Dev& D = *devs_[(size_t)v];
        (void)hp_set_device(D.device);
        for (;;) {
            {   // something to do?
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&]() { return !any_q_.empty() || !dev_q_[(size_t)v].empty(); });
            }
            if (!D.pipe) {   // (first use: the pipeline's threads, streams and pools are this device's from here on)
                hp_block_params dflt{};
                int rc = HP_OK;
                static const uint32_t depth = [] { const char* e = std::getenv("HP_DISPATCH_DEPTH"); return e ? (uint32_t)std::max(1, std::min(16, std::atoi(e))) : 5u; }();
                D.pipe = pipeline_create(&dflt, D.device, depth, &rc);
                if (!D.pipe) { fail_queued(v, rc != HP_OK ? rc : HP_ERR_HIP, hp_last_error()); continue; }
                { std::lock_guard<std::mutex> lk(m_); dev_failed_[(size_t)v] = 0; }
            }
            // a free slot first: whatever arrives while the pipeline is full joins this set - the batching of a busy device
            pipeline_wait_free(D.pipe);
            const uint32_t free_slots = std::max(1u, pipeline_free_slots(D.pipe));
            std::unique_ptr<MergedSet> ms(new MergedSet());
            {
                std::unique_lock<std::mutex> lk(m_);
                std::deque<BlocksReq*>& mine = dev_q_[(size_t)v];
                if (mine.empty() && any_q_.empty()) continue;   // (another device's feeder was quicker)
                ms->prm = !mine.empty() ? mine.front()->prm : any_q_.front()->prm;
                uint64_t rec = 0;
                auto take_from = [&](std::deque<BlocksReq*>& q, uint64_t cap_records) {
                    for (auto it = q.begin(); it != q.end() && rec < cap_records;) {
                        if (!same_params((*it)->prm, ms->prm)) { ++it; continue; }   // (its turn comes with the next set)
                        rec += (*it)->records;
                        ms->reqs.push_back(*it);
                        it = q.erase(it);
                    }
                };
                take_from(mine, max_records());
                // the common queue: this device's share of what is there - but not in crumbs (a set of a dozen blocks costs a
                // device the same stages as one of a few hundred)
                uint64_t queued = 0;
                for (BlocksReq* r : any_q_) queued += r->records;
                // ... and of that share, what one of this pipeline's FREE slots should carry: callers that block (T calls in flight,
                // main.rs:385) all sit in the sets that are on their way - one set for all of them is one set's latency per T blocks
                // (116 ms per 64 blocks, measured); spread over the free slots the stages of consecutive sets overlap. Callers that
                // submit and go on (40 x T in flight) fill every slot with a full-sized set either way.
                const uint64_t per_dev = (queued + (uint64_t)n_vdev_ - 1) / (uint64_t)n_vdev_;
                const uint64_t per_slot = (per_dev + (uint64_t)free_slots - 1) / (uint64_t)free_slots;
                // (a deep queue means callers that do not wait: full-sized sets)
                const uint64_t share = queued > 65536 ? max_records() : std::max<uint64_t>(per_slot, std::min<uint64_t>(queued, min_records()));
                if (rec < max_records()) take_from(any_q_, std::min(max_records(), rec + share));
            }
            if (ms->reqs.empty()) continue;
            bool submitted = false;
            try {   // (the feeder must never take the host process down: a failed host allocation is these callers' status)
                for (BlocksReq* r : ms->reqs) { ms->in.insert(ms->in.end(), r->in, r->in + r->n); ms->out.insert(ms->out.end(), r->out, r->out + r->n); }
                ms->submit_rc = pipeline_submit(D.pipe, ms->in.size(), ms->in.data(), &ms->prm, ms->out.data(), &ms->ticket);
                submitted = ms->submit_rc == HP_OK;
                if (ms->submit_rc != HP_OK) ms->submit_err = hp_last_error();
                std::lock_guard<std::mutex> lk(D.m);
                D.inflight.push_back(std::move(ms));
            } catch (const std::exception& e) {
                // (a set that IS in the pipeline - the hand-over to the completer is what failed - keeps reading ms->in and writing ms->out
                // and the callers' buffers: it is waited for before anybody is told anything, and its slot is given back)
                if (ms && submitted) (void)pipeline_wait(D.pipe, ms->ticket, nullptr, nullptr);
                if (ms) for (BlocksReq* r : ms->reqs) { r->rc = HP_ERR_OOM; r->err = std::string("host allocation failed while merging a block set: ") + e.what(); r->finish(); }
                continue;
            }
            D.cv.notify_all();
        }
    }
    void complete(int v) {
        Dev& D = *devs_[(size_t)v];
        (void)hp_set_device(D.device);
        for (;;) {
            std::unique_ptr<MergedSet> ms;
            {
                std::unique_lock<std::mutex> lk(D.m);
                D.cv.wait(lk, [&]() { return !D.inflight.empty(); });
                ms = std::move(D.inflight.front());
                D.inflight.pop_front();
            }
            int rc = ms->submit_rc;
            std::string err = ms->submit_err;
            if (rc == HP_OK) { rc = pipeline_wait(D.pipe, ms->ticket, nullptr, nullptr); if (rc != HP_OK) err = hp_last_error(); }
            if (rc == HP_OK) {
                size_t o = 0;
                for (BlocksReq* r : ms->reqs) { std::copy(ms->out.begin() + (ptrdiff_t)o, ms->out.begin() + (ptrdiff_t)(o + r->n), r->out); o += r->n; r->rc = HP_OK; }
            } else if (ms->reqs.size() == 1) { ms->reqs[0]->rc = rc; ms->reqs[0]->err = err; }
            else {
                // a merged set that fails is re-run request by request (on this thread, stages in a row) so that every caller gets
                // the status of its own blocks
                for (BlocksReq* r : ms->reqs) {
                    try {
                        r->rc = solve_blocks_on_device(r->n, r->in, &r->prm, r->out, D.device);
                        if (r->rc != HP_OK) r->err = hp_last_error();
                    } catch (const std::exception& e) { r->rc = HP_ERR_OOM; r->err = std::string("host allocation failed: ") + e.what(); }
                }
            }
            for (BlocksReq* r : ms->reqs) r->finish();
        }
    }
    // device v has no pipeline: its own requests fail; the common queue's fail too once NO device has one (with several devices and
    // every pipeline_create failing, nobody would ever take them and hp_solve_blocks / hp_block_wait would hang)
    void fail_queued(int v, int rc, const char* why) {
        std::vector<BlocksReq*> gone;
        {
            std::lock_guard<std::mutex> lk(m_);
            dev_failed_[(size_t)v] = 1;
            gone.assign(dev_q_[(size_t)v].begin(), dev_q_[(size_t)v].end());
            dev_q_[(size_t)v].clear();
            bool any_usable = false;
            for (int w = 0; w < n_vdev_; ++w) any_usable = any_usable || dev_failed_[(size_t)w] == 0;
            if (!any_usable) { gone.insert(gone.end(), any_q_.begin(), any_q_.end()); any_q_.clear(); }
        }
        for (BlocksReq* r : gone) { r->rc = rc; r->err = why ? why : "no pipeline"; r->finish(); }
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    std::mutex m_;
    std::condition_variable cv_work_;
    std::vector<char> dev_failed_;   // 1: the device's last pipeline_create failed (it is tried again with the next request); devices that have not tried yet count as usable
    std::deque<BlocksReq*> any_q_;
    std::vector<std::deque<BlocksReq*>> dev_q_;
    std::vector<std::unique_ptr<Dev>> devs_;
    bool started_ = false;
    int real_dev_ = 1, n_vdev_ = 1;
};

// a call with many blocks for the node's GPUs: LPT chunks through the dispatcher's queue
int solve_blocks_over_devices(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out) {
    BlockDispatcher& D = BlockDispatcher::get();
    const int ndev = D.devices();
    std::vector<uint32_t> order(n_blocks);
    for (size_t i = 0; i < n_blocks; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return in[a].n_records > in[b].n_records; });
    uint64_t total = 0;
    for (size_t i = 0; i < n_blocks; ++i) total += in[i].n_records + 1;
    const uint64_t target = std::max<uint64_t>(1, total / ((uint64_t)ndev * 8));
    struct Chunk { std::vector<uint32_t> ids; std::vector<hp_block_input> ci; std::vector<hp_block_output> co; std::unique_ptr<BlocksReq> req; };
    std::vector<std::unique_ptr<Chunk>> chunks;
    {
        uint64_t acc = 0;
        for (uint32_t b : order) {
            if (chunks.empty() || (acc >= target && !chunks.back()->ids.empty())) { chunks.emplace_back(new Chunk()); acc = 0; }
            chunks.back()->ids.push_back(b);
            acc += in[b].n_records + 1;
        }
    }
    std::vector<BlocksReq*> reqs;
    for (auto& c : chunks) {
        for (uint32_t b : c->ids) { c->ci.push_back(in[b]); c->co.push_back(out[b]); }
        c->req.reset(new BlocksReq(c->ids.size(), c->ci.data(), *p, c->co.data(), -1));
        reqs.push_back(c->req.get());
    }
    D.post(reqs.data(), reqs.size());
    D.wait(reqs.data(), reqs.size());
    int rc = HP_OK;
    for (auto& c : chunks) {
        if (c->req->rc != HP_OK) { if (rc == HP_OK) { rc = c->req->rc; set_error("%s", c->req->err.c_str()); } continue; }
        for (size_t k = 0; k < c->ids.size(); ++k) out[c->ids[k]] = c->co[k];
    }
    return rc;
}

}  // namespace

extern "C" int hp_solve_blocks(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out, int device_id) {
    if (n_blocks == 0) return HP_OK;
    if (!in || !p || !out) { set_error("null argument"); return HP_ERR_ARG; }
    const int ndev = hp_device_count();
    if (ndev <= 0) { set_error("no HIP device visible; there is no CPU fallback"); return HP_ERR_HIP; }
    if (device_id >= ndev) { set_error("device %d: %d visible", device_id, ndev); return HP_ERR_ARG; }
    BlockDispatcher& D = BlockDispatcher::get();
    if (device_id < 0 && n_blocks >= 2) {   // a whole batch for the node's GPUs
        if (D.devices() == 1) return solve_blocks_on_device(n_blocks, in, p, out, hp_default_device());
        return solve_blocks_over_devices(n_blocks, in, p, out);
    }
    if (!coalescing_enabled()) return solve_blocks_on_device(n_blocks, in, p, out, device_id < 0 ? hp_default_device() : device_id);
    // one block (or a set for a named device) from one of many caller threads: merged with what else is in flight
    if (D.entering.fetch_add(1, std::memory_order_acq_rel) == 0) {
        // nobody else is inside: run on the caller's own thread (its caches are the warm ones for a single-threaded host);
        // whoever arrives meanwhile queues for the pipelines
        const int rc = solve_blocks_on_device(n_blocks, in, p, out, device_id < 0 ? hp_default_device() : device_id);
        D.entering.fetch_sub(1, std::memory_order_acq_rel);
        return rc;
    }
    BlocksReq r(n_blocks, in, *p, out, device_id);
    BlocksReq* rp = &r;
    D.post(&rp, 1);
    D.wait(&rp, 1);
    D.entering.fetch_sub(1, std::memory_order_acq_rel);
    if (r.rc != HP_OK) set_error("%s", r.err.c_str());
    return r.rc;
}

// ---- the asynchronous per-block entry --------------------------------------------------------------------------------------------
// HiPhase keeps `job_slots = 40 x threads` blocks queued but only `threads` calls of solve_block in flight (reference
// src/main.rs:328,344-383). A worker that SUBMITS its block and goes on to load the next one keeps all 40 x threads of them in
// flight on the device side: hp_block_submit queues the block(s) for the pipelines and returns a ticket at once, hp_block_wait
// returns when the results are in `out`. `in`, everything it points at, and `out` must stay valid until the wait returns;
// any thread may wait. The ticket is consumed by the wait.
// Tickets are ids out of a table, never pointers (round 5): a ticket that was never issued, was waited for already or is waited for
// twice at once is an HP_ERR_ARG, as with hp_blockstream_wait - not a use-after-free in the caller's process.
namespace {
struct TicketTable {
    std::mutex m;
    std::unordered_map<uint64_t, std::unique_ptr<BlocksReq>> live;
    uint64_t next = 1;
    static TicketTable& get() { static auto* t = new TicketTable(); return *t; }   // never destroyed
};
}  // namespace

extern "C" int hp_block_submit(size_t n_blocks, const hp_block_input* in, const hp_block_params* p, hp_block_output* out, int device_id, uint64_t* ticket) {
    if (!ticket || !p || (n_blocks && (!in || !out))) { set_error("null argument"); return HP_ERR_ARG; }
    const int ndev = hp_device_count();
    if (ndev <= 0) { set_error("no HIP device visible; there is no CPU fallback"); return HP_ERR_HIP; }
    if (device_id >= ndev) { set_error("device %d: %d visible", device_id, ndev); return HP_ERR_ARG; }
    BlocksReq* rp = nullptr;
    try {
        std::unique_ptr<BlocksReq> r(new BlocksReq(n_blocks, in, *p, out, device_id < 0 ? -1 : device_id));
        rp = r.get();
        if (n_blocks == 0) rp->done = true;
        TicketTable& T = TicketTable::get();
        std::lock_guard<std::mutex> lk(T.m);
        *ticket = T.next++;
        T.live.emplace(*ticket, std::move(r));
    } catch (const std::exception&) { set_error("host allocation failed"); return HP_ERR_OOM; }
    if (n_blocks) BlockDispatcher::get().post(&rp, 1);
    return HP_OK;
}

extern "C" int hp_block_wait(uint64_t ticket) {
    std::unique_ptr<BlocksReq> r;
    {
        TicketTable& T = TicketTable::get();
        std::lock_guard<std::mutex> lk(T.m);
        auto it = T.live.find(ticket);
        if (it == T.live.end()) { set_error("hp_block_wait: ticket %llu was never issued or has been waited for already", (unsigned long long)ticket); return HP_ERR_ARG; }
        r = std::move(it->second);   // (the ticket is consumed here: a second wait on it - even one racing this one - finds nothing)
        T.live.erase(it);
    }
    r->wait_done();
    const int rc = r->rc;
    if (rc != HP_OK) set_error("%s", r->err.c_str());
    return rc;
}
