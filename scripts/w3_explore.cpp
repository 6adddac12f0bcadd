// w3_explore.cpp — EXPLORATION TOOL (host only, test infrastructure): runs the CPU model of the third-generation graph-WFA
// formulation (tests/cpp/wfa2_model.cpp) over the reads of a generated bench-like block set, checks it against the oracle and
// prints what a lockstep kernel would pay: rounds, tiles, lanes, inserts by kind. New formulations are tried HERE first.
//   g++ -O2 -std=c++17 -pthread -o build/w3_explore scripts/w3_explore.cpp oracle/hp_oracle_wfa.cpp hiphase_amd/csrc/hp_synth_reads.cpp hiphase_amd/csrc/hp_synth.cpp
//   build/w3_explore [total_hets] [noise] [model: 3|4]
#include "../tests/cpp/wfa2_model.cpp"
#include "../oracle/hp_oracle.h"

#include <cstdio>
#include <cstdlib>

extern uint64_t g_x[64];
extern int g_opt;
extern int g_spec_len;

namespace {
std::vector<uint8_t> decode(const uint8_t* p, uint32_t fmt, uint64_t first, uint64_t n) {
    std::vector<uint8_t> out(n);
    if (fmt == HP_SEQ_BAM4) {
        static const char tab[17] = "=ACMGRSVTWYHKDBN";
        for (uint64_t k = 0; k < n; ++k) { const uint64_t b = first + k; out[k] = (uint8_t)tab[(b & 1u) ? (p[b >> 1] & 15u) : (p[b >> 1] >> 4)]; }
    } else if (n) std::memcpy(out.data(), p + first, n);
    return out;
}
}  // namespace

int w4m_wfa_assign(const hp_wfa_job* job, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* alleles, int* path);

int main(int argc, char** argv) {
    hp_synth_reads_spec s;
    hp_synth_reads_defaults(&s);
    const uint32_t th_ = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
    s.total_hets = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
    if (argc > 2 && atof(argv[2]) < 0) hp_synth_reads_hifi(&s);   // negative noise: the HiFi-shaped model
    else if (argc > 2) s.edit_noise = atof(argv[2]);
    const int which = argc > 3 ? atoi(argv[3]) : 3;
    if (argc > 4) g_opt = atoi(argv[4]);
    if (argc > 5) g_spec_len = atoi(argv[5]);
    s.total_hets = th_; s.max_block_hets = 600;
    int st = 0;
    hp_synth_set* set = hp_synth_reads_create(&s, &st);
    if (!set) { printf("create failed %d\n", st); return 1; }
    size_t nb = 0;
    const hp_block_input* in = hp_synth_reads_inputs(set, &nb);
    uint64_t jobs = 0, mism = 0, big = 0, bases = 0, maxed = 0, sum_score = 0;
    for (size_t b = 0; b < nb; ++b) {
        const hp_block_input* B = &in[b];
        for (uint32_t i = 0; i < B->n_records; ++i) {
            const hp_block_record& rec = B->records[i];
            size_t first_overlap = 0, last_overlap = 0, first_hom = 0, last_hom = 0; bool hf = false, hh = false;
            for (size_t k = 0; k < B->n_hets; ++k) { const int64_t p = B->hets[k].position; if (p >= rec.min_position && p <= rec.max_position) { if (!hf) { first_overlap = k; hf = true; } last_overlap = k + 1; } }
            if (!hf) continue;
            for (size_t k = 0; k < B->n_homs; ++k) { const int64_t p = B->homs[k].position; if (p >= rec.min_position && p <= rec.max_position) { if (!hh) { first_hom = k; hh = true; } last_hom = k + 1; } }
            const std::vector<uint8_t> read = decode(rec.read_align, B->seq_format, rec.read_offset, rec.read_len);
            hp_wfa_job job{};
            job.reference = B->reference; job.ref_base = B->ref_base;
            job.ref_start = (uint64_t)rec.min_position; job.ref_end = (uint64_t)rec.max_position + 1;
            job.hets = B->hets + first_overlap; job.n_hets = (uint32_t)(last_overlap - first_overlap);
            job.homs = last_hom > first_hom ? B->homs + first_hom : nullptr; job.n_homs = last_hom > first_hom ? (uint32_t)(last_hom - first_hom) : 0;
            job.read = read.data(); job.read_len = (uint32_t)read.size();
            hp_wfa_result w0{}, w1{};
            std::vector<uint8_t> a0(job.n_hets + 1, 3), a1(job.n_hets + 1, 3);
            if (hpo_wfa_assign(&job, 500, 500, &w0, a0.data()) < 0) { printf("oracle error\n"); return 1; }
            int path = 0;
            const int rc = which == 4 ? w4m_wfa_assign(&job, 500, 500, &w1, a1.data(), &path) : w3m_wfa_assign(&job, 500, 500, &w1, a1.data(), &path);
            if (rc != 0) { printf("model error %d (block %zu record %u)\n", rc, b, i); ++mism; continue; }
            ++jobs; bases += read.size();
            if (path != 0) { ++big; continue; }
            if (w0.status == HP_WFA_MAX_ED) ++maxed; else sum_score += w0.score;
            if (w0.status != w1.status || w0.score != w1.score || (w0.status == 0 && a0 != a1)) { ++mism; printf("MISMATCH block %zu record %u: oracle %d/%llu model %d/%llu\n", b, i, (int)w0.status, (unsigned long long)w0.score, (int)w1.status, (unsigned long long)w1.score); }
        }
    }
    printf("jobs %llu (%.0f bases mean), %llu left the compact state, %llu at max_ed, mean score %.1f, MISMATCHES %llu\n", (unsigned long long)jobs, (double)bases / (double)jobs,
           (unsigned long long)big, (unsigned long long)maxed, (double)sum_score / (double)(jobs - big - maxed), (unsigned long long)mism);
    const double J = (double)(jobs - big);
    static const char* names[] = {"rounds", "tiles", "targets(lanes act)", "lanes has", "lanes committed", "finished waves", "inserts", "ins append", "ins join", "ins inside rem<=G", "ins inside rem>G",
                                  "tiles with fin", "tiles first-of-round", "targets at round start", "lanes discarded", "third-wave", "fin certain (omax==len)", "fin ext<=2", "fin on len<=2 node", "fin of injected wave", "certain: append", "certain: join", "certain: inside", "beyond first G, not append", "front insert without room", "x25"};
    for (int k = 0; k < 26; ++k) if (g_x[k]) printf("  %-26s %12llu  %.2f per job  %.3f per round  %.3f per tile\n", names[k], (unsigned long long)g_x[k], g_x[k] / J, (double)g_x[k] / (double)g_x[0], (double)g_x[k] / (double)g_x[1]);
    hp_synth_reads_destroy(set);
    return mism ? 1 : 0;
}


// ---- fourth-generation schedule, explored: tiles resolve injections among their own lanes; only a NEW target that lands inside the
// tile's key range cuts the tile; optionally the children of a short new child are placed with it (speculative, one level) ----
int g_spec_len = 0;   // children of a newly created target of at most this length get their own children's targets at once
namespace {
struct Set8 { uint32_t w[8]; };
struct Live4 { uint32_t key, off, kind; Set8 set; };
struct Tgt4 { uint32_t key; int back; std::vector<Set8> src; bool start = false, spec = false; };

int model_wfa4(const Built& b, const uint8_t* ref, const uint8_t* read, uint64_t prune, uint64_t max_ed, int G, uint64_t* score, uint32_t* out_set) {
    const uint32_t nn = b.info.n_nodes, other_len = b.job.read_len, last = nn - 1;
    std::set<std::pair<uint32_t, int32_t>> capped;
    for (int w = 0; w < 8; ++w) out_set[w] = 0;
    uint64_t farthest = 0, min_prog = 0;
    std::vector<Live4> prev, cur;
    auto children = [&](uint32_t n, std::vector<uint32_t>& out) {
        out.clear();
        const W2Node nd = b.nodes[n];
        const uint32_t nch = nd.child & 0xFFFFu;
        uint32_t scan = nd.child >> 16;
        for (uint32_t j = 0; j < nch; ++j) out.push_back(j == 0 ? (nd.c01 & 0xFFFFu) : (j == 1 ? (nd.c01 >> 16) : w2_next_child(b.edges.data(), n, scan)));
    };
    for (uint32_t ed = 0;; ++ed) {
        std::vector<Tgt4> T;
        if (ed == 0) { Tgt4 t; t.key = w3_key(0, 0); t.back = -1; t.start = true; T.push_back(t); }
        else {
            uint32_t prevkey = 0xFFFFFFFFu;
            for (size_t i = 0; i < prev.size(); ++i) {
                const uint32_t key = prev[i].key;
                const bool same = prevkey != 0xFFFFFFFFu && w3_key_node(prevkey) == w3_key_node(key);
                const int32_t gap = same ? w3_key_diag(key) - w3_key_diag(prevkey) : 1 << 30;
                const int cnt = gap == 1 ? 1 : (gap == 2 ? 2 : 3);
                for (int j = 0; j < cnt; ++j) { Tgt4 t; t.key = w3_key(w3_key_node(key), w3_key_diag(key) + 1 - (cnt - 1) + j); t.back = (int)i; T.push_back(t); }
                prevkey = key;
            }
        }
        cur.clear();
        bool final_found = false;
        uint64_t round_far = 0;
        size_t ip = 0;
        g_x[0]++; g_x[13] += T.size();
        bool first = true;
        while (ip < T.size()) {
            size_t wend = std::min(T.size(), ip + (size_t)G);   // the tile: targets [ip, wend)
            g_x[1]++; if (first) g_x[12]++; first = false;
            const size_t tile_lanes = wend - ip;
            g_x[2] += tile_lanes;
            size_t l = ip;
            bool had_fin = false;
            for (; l < wend; ++l) {
                const Tgt4 tgt = T[l];
                const uint32_t n = w3_key_node(tgt.key);
                const int32_t d = w3_key_diag(tgt.key);
                const W2Node nd = b.nodes[n];
                const uint32_t len = nd.len_ref & ~W2_IS_REF;
                const uint8_t* nseq = (nd.len_ref & W2_IS_REF) ? ref + nd.seq_off : b.pool.data() + nd.seq_off;
                int64_t oA = -1, oB = -1, oC = -1;
                const Set8 *qA = nullptr, *qB = nullptr, *qC = nullptr;
                if (tgt.back >= 0)
                    for (size_t k = (size_t)tgt.back; k < (size_t)tgt.back + 3 && k < prev.size(); ++k) {
                        const Live4& e = prev[k];
                        if (w3_key_node(e.key) != n) continue;
                        const int32_t dd = w3_key_diag(e.key);
                        if (dd == d + 1) { if (e.kind & 1u) { oA = (int64_t)e.off + 1; qA = &e.set; } }
                        else if (dd == d) { if (e.kind == W2_KIND_INTERIOR_READ) { oB = (int64_t)e.off + 1; qB = &e.set; } }
                        else if (dd == d - 1) { if (e.kind == W2_KIND_INTERIOR_READ || e.kind == W2_KIND_END_LAST) { oC = (int64_t)e.off; qC = &e.set; } }
                    }
                Set8 qD{};
                bool hinj = tgt.start;
                for (const Set8& sset : tgt.src) { hinj = true; for (int w = 0; w < 8; ++w) qD.w[w] |= sset.w[w]; }
                if (hinj) qD.w[n >> 5] |= 1u << (n & 31u);
                const bool has = oA >= 0 || oB >= 0 || oC >= 0 || hinj;
                if (!has) { g_x[23]++; continue; }
                g_x[3]++;
                const int64_t omax = std::max(std::max(oA, oB), std::max(oC, hinj ? (int64_t)0 : (int64_t)-1));
                auto extend = [&](int64_t o) -> int64_t {
                    int64_t pos = (int64_t)d + o;
                    while (o < (int64_t)len && pos >= 0 && pos < (int64_t)other_len && nseq[o] == read[pos]) { ++o; ++pos; }
                    return o;
                };
                const int64_t E = extend(omax);
                auto ties = [&](int64_t o) -> bool { if (o < 0) return false; if (o == omax) return true; return extend(o) == E; };
                const bool tA = ties(oA), tB = ties(oB), tC = ties(oC), tD = hinj && ties(0);
                const int64_t pos_end = (int64_t)d + E;
                const int64_t cap = std::min<int64_t>((int64_t)len, (int64_t)other_len - (int64_t)d);
                const bool is_capped = capped.count({n, d}) != 0;
                Set8 best{};
                for (int w = 0; w < 8; ++w) best.w[w] = (tA ? qA->w[w] : 0u) | (tB ? qB->w[w] : 0u) | (tC ? qC->w[w] : 0u) | (tD ? qD.w[w] : 0u);
                const bool is_final = n == last && E == (int64_t)len && pos_end == (int64_t)other_len;
                if (is_final) { final_found = true; for (int w = 0; w < 8; ++w) out_set[w] |= best.w[w]; }
                const bool skip = (is_capped && E < cap) || (pos_end < (int64_t)min_prog);
                if (skip) continue;
                if ((uint64_t)pos_end > round_far) round_far = (uint64_t)pos_end;
                if (E == cap && !is_capped) capped.insert({n, d});
                uint32_t kind;
                if (E == (int64_t)len) {
                    if (n == last) { if (pos_end < (int64_t)other_len) kind = W2_KIND_END_LAST; else continue; }
                    else kind = W2_KIND_FINISHED;
                } else kind = (pos_end < (int64_t)other_len) ? W2_KIND_INTERIOR_READ : W2_KIND_INTERIOR;
                if (kind != W2_KIND_FINISHED) { cur.push_back(Live4{tgt.key, (uint32_t)E, kind, best}); continue; }
                g_x[5]++; had_fin = true;
                std::vector<uint32_t> ch, gch;
                children(n, ch);
                const int32_t td = d + (int32_t)len;
                // the children's targets: among the rest of the tile (resolved there), beyond it (joined / inserted), or NEW inside it (cuts the tile)
                auto place = [&](uint32_t key, const Set8* srcset, bool spec) {
                    size_t pos = l + 1;
                    while (pos < T.size() && T[pos].key < key) ++pos;
                    g_x[6]++;
                    if (pos < T.size() && T[pos].key == key) {
                        if (srcset) T[pos].src.push_back(*srcset);
                        if (pos < wend) g_x[8]++; else g_x[9]++;     // joined inside the tile / beyond it
                        return;
                    }
                    Tgt4 t; t.key = key; t.back = -1; t.spec = spec;
                    if (srcset) t.src.push_back(*srcset);
                    T.insert(T.begin() + (long)pos, t);
                    if (pos < wend) { g_x[10]++; g_x[14] += wend - pos; wend = pos; }   // NEW inside the tile: everything from here on waits for the next tile
                    else if (pos + 1 == T.size()) g_x[7]++; else g_x[11]++;   // appended / inserted beyond the tile
                };
                for (uint32_t cid : ch) {
                    const uint32_t key = w3_key(cid, td);
                    const size_t before = T.size();
                    place(key, &best, false);
                    const uint32_t clen = b.nodes[cid].len_ref & ~W2_IS_REF;
                    if (T.size() != before && g_spec_len > 0 && clen <= (uint32_t)g_spec_len && cid != last) {   // a NEW short child: its children's targets come with it
                        children(cid, gch);
                        for (uint32_t gid : gch) { g_x[16]++; place(w3_key(gid, td + (int32_t)clen), nullptr, true); }
                    }
                }
            }
            if (had_fin) g_x[15]++;
            g_x[4] += l - ip;
            ip = wend;
        }
        if (final_found) { *score = ed; return W2_ST_OK; }
        if (round_far > farthest) farthest = round_far;
        if (farthest > prune) min_prog = farthest - prune;
        if ((uint64_t)ed + 1 > max_ed) { *score = max_ed; return W2_ST_MAX_ED; }
        if (cur.empty()) return W2_ST_INTERNAL;
        prev.swap(cur);
    }
}
}  // namespace

int w4m_wfa_assign(const hp_wfa_job* job, uint64_t prune_distance, uint64_t max_ed, hp_wfa_result* out, uint8_t* alleles, int* path) {
    Built b;
    build_from_job(job, b);
    *path = 0;
    if (b.info.status == W2B_NEED_HOST) { *path = 2; return 0; }
    if (b.info.status != W2B_OK) return HP_ERR_INVARIANT;
    if (b.info.n_nodes > 256) { *path = 1; return 0; }
    const uint8_t* ref = job->reference + (job->ref_start - job->ref_base);
    uint64_t score = 0;
    uint32_t set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int st = model_wfa4(b, ref, job->read, prune_distance, max_ed, 8, &score, set);
    if (st != W2_ST_OK && st != W2_ST_MAX_ED) return HP_ERR_INVARIANT;
    out->status = st == W2_ST_OK ? HP_OK : HP_WFA_MAX_ED;
    out->n_nodes = b.info.n_nodes;
    out->score = score;
    w2_map_alleles(b.tags.data(), b.info.n_tags, set, st == W2_ST_OK, alleles, job->n_hets);
    return 0;
}
