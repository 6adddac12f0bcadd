// hp_capture.cpp — `.hpbr`: READ-BEARING capture / replay of phase blocks (the `.hpbk` format of hp_api.hip only
// holds the solver's matrix, i.e. what is left AFTER the graph-WFA stage).
//
// BASELINE.json configs[2-4] are real HG002 runs; neither the data nor a Rust toolchain + htslib exist in this image. A HiPhase
// built with the patch of INTEGRATION.md calls hp_hpbr_append where it calls hp_solve_blocks (reference src/phaser.rs:513-543:
// the decoded records, variant calls and reference window of one block are at hand there), optionally with the results its OWN
// solve_block produced; the file then replays here through hp_solve_blocks / hp_blockstream_* / bench.py --replay and every
// block is compared with what the real binary answered.
//
// Layout (little endian; one record per block, records concatenated; every array padded with zeros to 8 bytes):
//   "HPBR0002" ("HPBR0001": the same without the 56 ReadStats words of the expected section; still read)
//   u64[16]: block_index, n_hets, n_homs, n_records, n_qnames, seq_format, ref_lo (chromosome coordinate of the first
//            reference byte stored), ref_len, has_local_hets, has_expected, bytes of the variant section, bytes of the record
//            section, bytes of the expected section, reserved x 3
//   hp_block_params (as the C struct, 88 bytes)
//   reference bytes [ref_len]                       (the hull of the records' windows: nothing else is read)
//   het_types u8[n_hets]
//   variants: hets then homs, each: i64 position, u32 ref_len, u32 flags, u32 allele0_len, u32 allele1_len, allele0, allele1
//   local hets (if has_local_hets), each: i64 position, u32 ref_len, variant_type, prefix_len, postfix_len, allele0_len,
//            allele1_len, flags, 0, allele0, allele1
//   records, each: i64 min_position, i64 max_position, u32 read_len, u32 qname_id, u32 read_offset, u32 has_local,
//            read bytes (seq_format; from the byte that holds base read_offset - read_offset % 2 ... so read_offset is kept mod 2),
//            if has_local: i64 pos, u32 n_cigar, u32 seq_len, u32 seq_format, u32 0, cigar u32[n_cigar], seq bytes, qual bytes
//   expected (if has_expected): status i32 + pad, h1, h2 u8[n_hets], hp_phase_stats u64[7], span_counts u64[n_hets - 1],
//            n_segments, n_solver u32, num_reads, skipped_reads, global_aligned, local_aligned, n_edit_distances u64,
//            (0002:) num_alleles, exact / inexact / failed / allele0 / allele1 _matches u64[11] each,
//            edit_distances, seg_qname, seg_start, seg_end u32[n_segments], seg_solver, seg_haplotag u8[n_segments],
//            seg_first_het u32[n_segments], seg_row_off u64[n_segments + 1], seg_alleles, seg_quals u8[cells]
// Host-only; compiled into libhiphase_gpu.so and the test oracle alike.
#include "../../include/hiphase_gpu.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace {

struct Writer {
    std::vector<uint8_t> buf;
    void raw(const void* p, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(p); buf.insert(buf.end(), b, b + n); }
    void pad() { while (buf.size() % 8) buf.push_back(0); }
    void arr(const void* p, size_t n) { if (n) raw(p, n); pad(); }
    template <class T> void val(T v) { raw(&v, sizeof v); }
};

size_t seq_bytes(uint32_t fmt, uint64_t first_base, uint64_t n_bases) {   // bytes that hold bases [first_base, first_base + n)
    if (fmt == HP_SEQ_BAM4) return n_bases ? (size_t)(((first_base + n_bases - 1) >> 1) - (first_base >> 1) + 1) : 0;
    return (size_t)n_bases;
}

thread_local std::string g_cap_err;

}  // namespace

extern "C" const char* hp_hpbr_last_error(void) { return g_cap_err.c_str(); }

// (host-only like the rest of this file, so that libhiphase_capture.so - `make capture`, g++ alone - carries it: a capture
// machine needs neither ROCm nor a GPU; errors of both writers are read with hp_hpbr_last_error)
// Capture side of the .hpbk format (hiphase_amd/block_io.py): a patched HiPhase calls this at
// reference src/phaser.rs:541-543 with the solver's exact input and, after astar_solver returns, its output.
extern "C" int hp_hpbk_append(const char* path, const hp_block_view* blk, const hp_astar_params* p, const uint8_t* h1, const uint8_t* h2,
                   const hp_phase_stats* stats) {
    if (!path || !blk || !p) { g_cap_err = "null argument"; return HP_ERR_ARG; }
    FILE* f = std::fopen(path, "ab");
    if (!f) { g_cap_err = std::string("cannot open ") + path + " for appending"; return HP_ERR_ARG; }
    const uint64_t N = blk->n_variants, R = blk->n_reads, cells = R ? blk->row_off[R] : 0;
    const bool has_exp = h1 && h2 && stats;
    static const unsigned char zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = true;
    auto put = [&](const void* d, size_t n) { if (n) ok = ok && std::fwrite(d, 1, n, f) == n; const size_t pad = (8 - n % 8) % 8; if (pad) ok = ok && std::fwrite(zeros, 1, pad, f) == pad; };
    const uint64_t hdr[10] = {p->block_index, N, R, cells, p->min_queue_size, p->queue_increment, has_exp ? 1u : 0u, 0, 0, 0};
    ok = ok && std::fwrite("HPBK0001", 1, 8, f) == 8;
    ok = ok && std::fwrite(hdr, 8, 10, f) == 10;
    put(blk->read_start, R * 4); put(blk->read_end, R * 4); put(blk->row_off, (R + 1) * 8);
    put(blk->alleles_2bit, (cells + 3) / 4); put(blk->quals, cells); put(blk->var_flags, N);
    if (has_exp) { put(h1, N); put(h2, N); ok = ok && std::fwrite(stats, 8, 7, f) == 7; }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { g_cap_err = std::string("short write to ") + path; return HP_ERR_ARG; }
    return HP_OK;
}

extern "C" int hp_hpbr_append(const char* path, const hp_block_input* B, const hp_block_params* P, const hp_block_output* E) {
    if (!path || !B || !P) { g_cap_err = "null argument"; return HP_ERR_ARG; }
    if (B->seq_format != HP_SEQ_ASCII && B->seq_format != HP_SEQ_BAM4) { g_cap_err = "unknown seq_format"; return HP_ERR_ARG; }
    // the reference bytes the path can read: the hull of the records' windows (plus nothing - local re-alignment reads the
    // variants' padded alleles, not the reference)
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (uint32_t r = 0; r < B->n_records; ++r) { lo = std::min(lo, B->records[r].min_position); hi = std::max(hi, B->records[r].max_position + 1); }
    if (lo > hi || !B->reference) { lo = hi = (int64_t)B->ref_base; }
    if (lo < (int64_t)B->ref_base) { g_cap_err = "a record starts before the reference buffer"; return HP_ERR_ARG; }
    Writer var, rec, exp;
    auto put_var = [&](const hp_wfa_variant& v) {
        var.val<int64_t>(v.position); var.val<uint32_t>(v.ref_len); var.val<uint32_t>(v.flags);
        const uint32_t l0 = (v.flags & 2u) ? v.allele0_len : 0u;
        var.val<uint32_t>(l0); var.val<uint32_t>(v.allele1_len);
        var.arr(v.allele0, l0); var.arr(v.allele1, v.allele1_len);
    };
    var.arr(B->het_types, B->n_hets);
    for (uint32_t i = 0; i < B->n_hets; ++i) put_var(B->hets[i]);
    for (uint32_t i = 0; i < B->n_homs; ++i) put_var(B->homs[i]);
    if (B->local_hets)
        for (uint32_t i = 0; i < B->n_hets; ++i) {
            const hp_local_variant& v = B->local_hets[i];
            var.val<int64_t>(v.position);
            const uint32_t w[8] = {v.ref_len, v.variant_type, v.prefix_len, v.postfix_len, v.allele0_len, v.allele1_len, v.flags, 0};
            var.raw(w, sizeof w);
            var.arr(v.allele0, v.allele0_len); var.arr(v.allele1, v.allele1_len);
        }
    for (uint32_t r = 0; r < B->n_records; ++r) {
        const hp_block_record& R = B->records[r];
        rec.val<int64_t>(R.min_position); rec.val<int64_t>(R.max_position);
        const uint32_t off = B->seq_format == HP_SEQ_BAM4 ? (R.read_offset & 1u) : 0u;
        rec.val<uint32_t>(R.read_len); rec.val<uint32_t>(R.qname_id); rec.val<uint32_t>(off); rec.val<uint32_t>(R.local ? 1u : 0u);
        const uint8_t* src = R.read_align ? R.read_align + (B->seq_format == HP_SEQ_BAM4 ? (R.read_offset >> 1) : R.read_offset) : nullptr;
        rec.arr(src, seq_bytes(B->seq_format, off, R.read_len));
        if (R.local) {
            const hp_local_read& L = *R.local;
            rec.val<int64_t>(L.pos); rec.val<uint32_t>(L.n_cigar); rec.val<uint32_t>(L.seq_len); rec.val<uint32_t>(L.seq_format); rec.val<uint32_t>(0);
            rec.arr(L.cigar, (size_t)L.n_cigar * 4);
            rec.arr(L.seq, seq_bytes(L.seq_format, 0, L.seq_len));
            rec.arr(L.qual, L.seq_len);
        }
    }
    if (E) {
        const size_t N = B->n_hets, ns = E->n_segments;
        exp.val<int32_t>(E->status); exp.val<uint32_t>(0);
        exp.arr(E->h1, E->h1 ? N : 0); exp.arr(E->h2, E->h2 ? N : 0);
        exp.raw(&E->stats, sizeof E->stats);
        exp.arr(E->span_counts, E->span_counts && N > 1 ? (N - 1) * 8 : 0);
        exp.val<uint32_t>(E->n_segments); exp.val<uint32_t>(E->n_solver);
        exp.val<uint64_t>(E->num_reads); exp.val<uint64_t>(E->skipped_reads); exp.val<uint64_t>(E->global_aligned); exp.val<uint64_t>(E->local_aligned);
        exp.val<uint64_t>(E->n_edit_distances);
        exp.val<uint64_t>(E->num_alleles);
        exp.raw(E->exact_matches, sizeof E->exact_matches); exp.raw(E->inexact_matches, sizeof E->inexact_matches); exp.raw(E->failed_matches, sizeof E->failed_matches);
        exp.raw(E->allele0_matches, sizeof E->allele0_matches); exp.raw(E->allele1_matches, sizeof E->allele1_matches);
        exp.arr(E->edit_distances, E->edit_distances ? (size_t)E->n_edit_distances * 8 : 0);
        if (!E->seg_qname || !E->seg_start || !E->seg_end || !E->seg_solver || !E->seg_haplotag || !E->seg_first_het || !E->seg_row_off || !E->seg_alleles || !E->seg_quals ||
            !E->h1 || !E->h2 || !E->span_counts || !E->edit_distances) { g_cap_err = "expected output with a null array"; return HP_ERR_ARG; }
        exp.arr(E->seg_qname, ns * 4); exp.arr(E->seg_start, ns * 4); exp.arr(E->seg_end, ns * 4); exp.arr(E->seg_solver, ns); exp.arr(E->seg_haplotag, ns);
        exp.arr(E->seg_first_het, ns * 4); exp.arr(E->seg_row_off, (ns + 1) * 8);
        const uint64_t cells = E->seg_row_off[ns];
        exp.arr(E->seg_alleles, (size_t)cells); exp.arr(E->seg_quals, (size_t)cells);
    }
    FILE* f = std::fopen(path, "ab");
    if (!f) { g_cap_err = std::string("cannot open ") + path + " for appending"; return HP_ERR_ARG; }
    const uint64_t hdr[16] = {B->block_index, B->n_hets, B->n_homs, B->n_records, B->n_qnames, B->seq_format, (uint64_t)lo, (uint64_t)(hi - lo),
                              B->local_hets ? 1u : 0u, E ? 1u : 0u, var.buf.size(), rec.buf.size(), exp.buf.size(), 0, 0, 0};
    bool ok = std::fwrite("HPBR0002", 1, 8, f) == 8 && std::fwrite(hdr, 8, 16, f) == 16 && std::fwrite(P, sizeof *P, 1, f) == 1;
    static const uint8_t zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const size_t rl = (size_t)(hi - lo);
    if (rl) ok = ok && std::fwrite(B->reference + ((uint64_t)lo - B->ref_base), 1, rl, f) == rl;
    if (rl % 8) ok = ok && std::fwrite(zeros, 1, 8 - rl % 8, f) == 8 - rl % 8;
    for (const Writer* w : {&var, &rec, &exp}) if (!w->buf.empty()) ok = ok && std::fwrite(w->buf.data(), 1, w->buf.size(), f) == w->buf.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) { g_cap_err = std::string("short write to ") + path; return HP_ERR_ARG; }
    return HP_OK;
}

// ---- reader -------------------------------------------------------------------------------------------------------------------
struct hp_hpbr {
    struct Block {
        std::vector<uint8_t> raw;     // the block's bytes as they are in the file: every pointer below points into it
        std::vector<hp_wfa_variant> hets, homs;
        std::vector<hp_local_variant> local_hets;
        std::vector<hp_block_record> records;
        std::vector<hp_local_read> locals;
        hp_block_params prm{};
    };
    std::vector<std::unique_ptr<Block>> blocks;
    std::vector<hp_block_input> inputs;
    std::vector<hp_block_params> params;
    std::vector<hp_block_output> expected;   // status = INT32_MIN where the capture holds no expected output
    std::vector<uint8_t> has_expected;
};

extern "C" hp_hpbr* hp_hpbr_open(const char* path, int* status) {
    auto fail = [&](int rc, const std::string& msg) -> hp_hpbr* { g_cap_err = msg; if (status) *status = rc; return nullptr; };
    if (!path) return fail(HP_ERR_ARG, "null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(HP_ERR_ARG, std::string("cannot open ") + path);
    auto h = std::unique_ptr<hp_hpbr>(new hp_hpbr());
    for (;;) {
        char magic[8];
        const size_t got = std::fread(magic, 1, 8, f);
        if (got == 0) break;
        uint64_t hdr[16];
        const int version = (got == 8 && !std::memcmp(magic, "HPBR0002", 8)) ? 2 : ((got == 8 && !std::memcmp(magic, "HPBR0001", 8)) ? 1 : 0);
        if (version == 0 || std::fread(hdr, 8, 16, f) != 16) { std::fclose(f); return fail(HP_ERR_ARG, "not an .hpbr capture (bad magic / truncated header)"); }
        auto blk = std::unique_ptr<hp_hpbr::Block>(new hp_hpbr::Block());
        if (std::fread(&blk->prm, sizeof blk->prm, 1, f) != 1) { std::fclose(f); return fail(HP_ERR_ARG, "truncated capture"); }
        const uint64_t n_hets = hdr[1], n_homs = hdr[2], n_rec = hdr[3], ref_len = hdr[7];
        const size_t ref_pad = (size_t)((ref_len + 7) & ~7ull);
        const size_t total = ref_pad + (size_t)hdr[10] + (size_t)hdr[11] + (size_t)hdr[12];
        blk->raw.resize(total + 8);
        if (total && std::fread(blk->raw.data(), 1, total, f) != total) { std::fclose(f); return fail(HP_ERR_ARG, "truncated capture"); }
        const uint8_t* p = blk->raw.data() + ref_pad;
        const uint8_t* end = blk->raw.data() + total;
        bool bad = false;
        auto take = [&](size_t n) -> const uint8_t* { const uint8_t* q = p; const size_t adv = (n + 7) & ~(size_t)7; if ((size_t)(end - p) < adv) { bad = true; return blk->raw.data(); } p += adv; return q; };
        auto rd32 = [&](const uint8_t* q) { uint32_t v; std::memcpy(&v, q, 4); return v; };
        auto rd64 = [&](const uint8_t* q) { uint64_t v; std::memcpy(&v, q, 8); return v; };
        hp_block_input I{};
        I.block_index = hdr[0]; I.n_hets = (uint32_t)n_hets; I.n_homs = (uint32_t)n_homs; I.n_records = (uint32_t)n_rec; I.n_qnames = (uint32_t)hdr[4];
        I.seq_format = (uint32_t)hdr[5]; I.ref_base = hdr[6]; I.reference = blk->raw.data();
        I.het_types = take((size_t)n_hets);
        auto get_var = [&](hp_wfa_variant& v) {
            const uint8_t* q = take(24);
            v = hp_wfa_variant{};
            v.position = (int64_t)rd64(q); v.ref_len = rd32(q + 8); v.flags = rd32(q + 12); v.allele0_len = rd32(q + 16); v.allele1_len = rd32(q + 20);
            v.allele0 = take(v.allele0_len); v.allele1 = take(v.allele1_len);
        };
        blk->hets.resize((size_t)n_hets); blk->homs.resize((size_t)n_homs);
        for (auto& v : blk->hets) { get_var(v); if (bad) break; }
        for (auto& v : blk->homs) { if (bad) break; get_var(v); }
        if (hdr[8] && !bad) {
            blk->local_hets.resize((size_t)n_hets);
            for (auto& v : blk->local_hets) {
                const uint8_t* q = take(40);
                if (bad) break;
                v = hp_local_variant{};
                v.position = (int64_t)rd64(q); v.ref_len = rd32(q + 8); v.variant_type = rd32(q + 12); v.prefix_len = rd32(q + 16); v.postfix_len = rd32(q + 20);
                v.allele0_len = rd32(q + 24); v.allele1_len = rd32(q + 28); v.flags = rd32(q + 32);
                v.allele0 = take(v.allele0_len); v.allele1 = take(v.allele1_len);
            }
        }
        blk->records.resize((size_t)n_rec); blk->locals.resize((size_t)n_rec);
        for (size_t r = 0; r < n_rec && !bad; ++r) {
            const uint8_t* q = take(32);
            if (bad) break;
            hp_block_record& R = blk->records[r];
            R = hp_block_record{};
            R.min_position = (int64_t)rd64(q); R.max_position = (int64_t)rd64(q + 8); R.read_len = rd32(q + 16); R.qname_id = rd32(q + 20); R.read_offset = rd32(q + 24);
            const uint32_t has_local = rd32(q + 28);
            R.read_align = take(seq_bytes(I.seq_format, R.read_offset, R.read_len));
            if (has_local) {
                const uint8_t* l = take(24);
                if (bad) break;
                hp_local_read& L = blk->locals[r];
                L = hp_local_read{};
                L.pos = (int64_t)rd64(l); L.n_cigar = rd32(l + 8); L.seq_len = rd32(l + 12); L.seq_format = rd32(l + 16);
                L.cigar = reinterpret_cast<const uint32_t*>(take((size_t)L.n_cigar * 4));
                L.seq = take(seq_bytes(L.seq_format, 0, L.seq_len));
                L.qual = take(L.seq_len);
                R.local = &L;
            }
        }
        hp_block_output E{};
        E.status = INT32_MIN;
        if (hdr[9] && !bad) {
            const uint8_t* q = take(8);
            int32_t st; std::memcpy(&st, q, 4);
            E.status = st;
            E.h1 = const_cast<uint8_t*>(take((size_t)n_hets)); E.h2 = const_cast<uint8_t*>(take((size_t)n_hets));
            std::memcpy(&E.stats, take(sizeof E.stats), sizeof E.stats);
            E.span_counts = reinterpret_cast<uint64_t*>(const_cast<uint8_t*>(take(n_hets > 1 ? (size_t)(n_hets - 1) * 8 : 0)));
            q = take(8); E.n_segments = rd32(q); E.n_solver = rd32(q + 4);
            q = take(40); E.num_reads = rd64(q); E.skipped_reads = rd64(q + 8); E.global_aligned = rd64(q + 16); E.local_aligned = rd64(q + 24); E.n_edit_distances = rd64(q + 32);
            if (version >= 2) {
                q = take(8 * (1 + 5 * HP_N_VARIANT_TYPES));
                if (!bad) {
                    E.num_alleles = rd64(q);
                    std::memcpy(E.exact_matches, q + 8, sizeof E.exact_matches); std::memcpy(E.inexact_matches, q + 8 + 88, sizeof E.inexact_matches);
                    std::memcpy(E.failed_matches, q + 8 + 176, sizeof E.failed_matches); std::memcpy(E.allele0_matches, q + 8 + 264, sizeof E.allele0_matches);
                    std::memcpy(E.allele1_matches, q + 8 + 352, sizeof E.allele1_matches);
                }
            }
            E.edit_distances = reinterpret_cast<uint64_t*>(const_cast<uint8_t*>(take((size_t)E.n_edit_distances * 8)));
            const size_t ns = E.n_segments;
            E.seg_qname = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(take(ns * 4))); E.seg_start = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(take(ns * 4)));
            E.seg_end = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(take(ns * 4))); E.seg_solver = const_cast<uint8_t*>(take(ns)); E.seg_haplotag = const_cast<uint8_t*>(take(ns));
            E.seg_first_het = reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(take(ns * 4)));
            E.seg_row_off = reinterpret_cast<uint64_t*>(const_cast<uint8_t*>(take((ns + 1) * 8)));
            const uint64_t cells = bad ? 0 : E.seg_row_off[ns];
            E.seg_alleles = const_cast<uint8_t*>(take((size_t)cells)); E.seg_quals = const_cast<uint8_t*>(take((size_t)cells));
            E.seg_cell_cap = cells;
        }
        if (bad) { std::fclose(f); return fail(HP_ERR_ARG, "corrupt capture (a section is shorter than its contents)"); }
        I.hets = blk->hets.data(); I.homs = blk->homs.data(); I.local_hets = blk->local_hets.empty() ? nullptr : blk->local_hets.data(); I.records = blk->records.data();
        h->inputs.push_back(I);
        h->params.push_back(blk->prm);
        h->expected.push_back(E);
        h->has_expected.push_back(hdr[9] ? 1 : 0);
        h->blocks.push_back(std::move(blk));
    }
    std::fclose(f);
    if (status) *status = HP_OK;
    return h.release();
}

extern "C" const hp_block_input* hp_hpbr_inputs(const hp_hpbr* h, size_t* n_blocks) {
    if (!h) return nullptr;
    if (n_blocks) *n_blocks = h->inputs.size();
    return h->inputs.data();
}
extern "C" const hp_block_params* hp_hpbr_params(const hp_hpbr* h) { return h ? h->params.data() : nullptr; }
extern "C" const hp_block_output* hp_hpbr_expected(const hp_hpbr* h) { return h ? h->expected.data() : nullptr; }
extern "C" void hp_hpbr_close(hp_hpbr* h) { delete h; }

// ---- caller-side output buffers for a list of blocks, sized from the inputs ---------------------------------------------------------
struct hp_outputs {
    std::vector<hp_block_output> out;
    struct Store {
        std::vector<uint8_t> h1, h2, seg_solver, seg_haplotag, seg_alleles, seg_quals;
        std::vector<uint64_t> span_counts, seg_row_off, edit_distances;
        std::vector<uint32_t> seg_qname, seg_start, seg_end, seg_first_het;
    };
    std::vector<Store> store;
};

extern "C" hp_outputs* hp_outputs_create(const hp_block_input* in, size_t n) {
    if (!in && n) return nullptr;
    auto o = std::unique_ptr<hp_outputs>(new hp_outputs());
    o->out.resize(n); o->store.resize(n);
    for (size_t b = 0; b < n; ++b) {
        const hp_block_input& I = in[b];
        // cells of the collapsed segments: per read name at most the hull of its records' het ranges (a record's range from its
        // reference span: min / max position, and its CIGAR where it brings one)
        std::vector<uint32_t> qlo(I.n_qnames, UINT32_MAX), qhi(I.n_qnames, 0);
        for (uint32_t r = 0; r < I.n_records; ++r) {
            const hp_block_record& R = I.records[r];
            if (R.qname_id >= I.n_qnames) continue;
            int64_t lo = R.min_position, hi = R.max_position;
            if (R.local) {
                int64_t ref = R.local->pos;
                lo = std::min(lo, ref);
                for (uint32_t k = 0; k < R.local->n_cigar; ++k) { const uint32_t op = R.local->cigar[k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref += R.local->cigar[k] >> 4; }
                hi = std::max(hi, ref);
            }
            // (local re-alignment looks a variant's padding beyond the alignment: one het of slack on either side)
            uint32_t f = (uint32_t)(std::lower_bound(I.hets, I.hets + I.n_hets, lo, [](const hp_wfa_variant& v, int64_t p) { return v.position < p; }) - I.hets);
            uint32_t l = (uint32_t)(std::upper_bound(I.hets, I.hets + I.n_hets, hi, [](int64_t p, const hp_wfa_variant& v) { return p < v.position; }) - I.hets);
            f = f ? f - 1 : 0; l = std::min(I.n_hets, l + 1);
            qlo[R.qname_id] = std::min(qlo[R.qname_id], f); qhi[R.qname_id] = std::max(qhi[R.qname_id], l);
        }
        size_t cap = 16;
        for (uint32_t q = 0; q < I.n_qnames; ++q) if (qhi[q] > qlo[q]) cap += qhi[q] - qlo[q];
        auto& st = o->store[b];
        const size_t N = I.n_hets, q = std::max<uint32_t>(I.n_qnames, 1);
        st.h1.assign(std::max<size_t>(N, 1), 0); st.h2.assign(std::max<size_t>(N, 1), 0); st.span_counts.assign(std::max<size_t>(N, 2) - 1, 0);
        st.seg_qname.assign(q, 0); st.seg_start.assign(q, 0); st.seg_end.assign(q, 0); st.seg_solver.assign(q, 0); st.seg_haplotag.assign(q, 0);
        st.seg_first_het.assign(q, 0); st.seg_row_off.assign(q + 1, 0); st.seg_alleles.assign(cap, 0); st.seg_quals.assign(cap, 0);
        st.edit_distances.assign(std::max<uint32_t>(I.n_records, 1), 0);
        hp_block_output& O = o->out[b];
        O = hp_block_output{};
        O.h1 = st.h1.data(); O.h2 = st.h2.data(); O.span_counts = st.span_counts.data();
        O.seg_qname = st.seg_qname.data(); O.seg_start = st.seg_start.data(); O.seg_end = st.seg_end.data(); O.seg_solver = st.seg_solver.data();
        O.seg_haplotag = st.seg_haplotag.data(); O.seg_first_het = st.seg_first_het.data(); O.seg_row_off = st.seg_row_off.data();
        O.seg_alleles = st.seg_alleles.data(); O.seg_quals = st.seg_quals.data(); O.seg_cell_cap = cap;
        O.edit_distances = st.edit_distances.data();
    }
    return o.release();
}
extern "C" hp_block_output* hp_outputs_array(hp_outputs* o) { return o ? o->out.data() : nullptr; }
// Fills everything a solve may write - every array to its capacity and every scalar result - with the byte `fill`, keeping the pointers
// and the capacities: two output sets poisoned with DIFFERENT bytes can only compare equal in fields that were really written.
extern "C" void hp_outputs_poison(hp_outputs* o, uint8_t fill) {
    if (!o) return;
    for (size_t b = 0; b < o->out.size(); ++b) {
        auto& st = o->store[b];
        auto fillv = [&](auto& v) { if (!v.empty()) std::memset(v.data(), fill, v.size() * sizeof v[0]); };
        fillv(st.h1); fillv(st.h2); fillv(st.span_counts); fillv(st.seg_qname); fillv(st.seg_start); fillv(st.seg_end); fillv(st.seg_solver);
        fillv(st.seg_haplotag); fillv(st.seg_first_het); fillv(st.seg_row_off); fillv(st.seg_alleles); fillv(st.seg_quals); fillv(st.edit_distances);
        hp_block_output& O = o->out[b];
        const hp_block_output keep = O;
        std::memset(&O, fill, sizeof O);
        O.h1 = keep.h1; O.h2 = keep.h2; O.span_counts = keep.span_counts; O.seg_qname = keep.seg_qname; O.seg_start = keep.seg_start; O.seg_end = keep.seg_end;
        O.seg_solver = keep.seg_solver; O.seg_haplotag = keep.seg_haplotag; O.seg_first_het = keep.seg_first_het; O.seg_row_off = keep.seg_row_off;
        O.seg_alleles = keep.seg_alleles; O.seg_quals = keep.seg_quals; O.seg_cell_cap = keep.seg_cell_cap; O.edit_distances = keep.edit_distances;
        O.reserved = 0;
    }
}
extern "C" void hp_outputs_destroy(hp_outputs* o) { delete o; }

// every field hp_solve_blocks fills, block `b` of two output sets over the same inputs: 1 = identical
extern "C" int hp_block_output_equal(const hp_block_input* in, const hp_block_output* a, const hp_block_output* b) {
    if (!in || !a || !b) return 0;
    const size_t N = in->n_hets;
    if (a->status != b->status) return 0;
    if (a->n_segments != b->n_segments || a->n_solver != b->n_solver || a->num_reads != b->num_reads || a->skipped_reads != b->skipped_reads ||
        a->global_aligned != b->global_aligned || a->local_aligned != b->local_aligned || a->n_edit_distances != b->n_edit_distances) return 0;
    if (a->num_alleles != b->num_alleles || std::memcmp(a->exact_matches, b->exact_matches, sizeof a->exact_matches) ||
        std::memcmp(a->inexact_matches, b->inexact_matches, sizeof a->inexact_matches) || std::memcmp(a->failed_matches, b->failed_matches, sizeof a->failed_matches) ||
        std::memcmp(a->allele0_matches, b->allele0_matches, sizeof a->allele0_matches) || std::memcmp(a->allele1_matches, b->allele1_matches, sizeof a->allele1_matches)) return 0;
    if (a->n_edit_distances && std::memcmp(a->edit_distances, b->edit_distances, a->n_edit_distances * 8)) return 0;
    const size_t ns = a->n_segments;
    if (ns && (std::memcmp(a->seg_qname, b->seg_qname, ns * 4) || std::memcmp(a->seg_start, b->seg_start, ns * 4) || std::memcmp(a->seg_end, b->seg_end, ns * 4) ||
               std::memcmp(a->seg_solver, b->seg_solver, ns) || std::memcmp(a->seg_row_off, b->seg_row_off, (ns + 1) * 8))) return 0;
    const uint64_t cells = ns ? a->seg_row_off[ns] : 0;
    if (cells && (std::memcmp(a->seg_alleles, b->seg_alleles, cells) || std::memcmp(a->seg_quals, b->seg_quals, cells))) return 0;
    if (a->status != HP_OK) return 1;   // (an unsupported block carries segments only)
    if (std::memcmp(a->h1, b->h1, N) || std::memcmp(a->h2, b->h2, N) || std::memcmp(&a->stats, &b->stats, sizeof a->stats)) return 0;
    if (N > 1 && std::memcmp(a->span_counts, b->span_counts, (N - 1) * 8)) return 0;
    if (ns && (std::memcmp(a->seg_haplotag, b->seg_haplotag, ns) || std::memcmp(a->seg_first_het, b->seg_first_het, ns * 4))) return 0;
    return 1;
}
