// hp_synth.cpp — deterministic synthetic phase-block generator.
//
// Host-only. Produces the read x variant allele matrix of one phase block in the hp_block_view
// CSR layout, the way read_parsing.rs would after ReadSegment::new clipping
// (reference src/data_types/read_segments.rs:40-62), with global-realignment qualities
// (reference src/read_parsing.rs:18-22,815: 2 x {SNV 80, indel 10, SV 20, TR 40}).
//
// Draw order (one splitmix64 draw per decision, strictly in this order):
//   truth[0..N), type[0..N), [ignored[0..N) only if ignored_permille > 0],
//   then per read r: L, start, hap, then per cell left to right: flip draw, ambiguous draw.
#include "../../include/hiphase_gpu.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {
struct SplitMix64 {
    uint64_t x;
    explicit SplitMix64(uint64_t seed) : x(seed) {}
    uint64_t next() {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double u01() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    bool bit() { return u01() < 0.5; }
};

inline uint32_t num_reads(const hp_synth_spec* s) {
    uint64_t nc = (uint64_t)s->n_variants * s->coverage;
    return (uint32_t)((nc + s->span - 1) / s->span);
}
inline uint32_t max_len(const hp_synth_spec* s) {
    uint32_t m = (uint32_t)std::ceil(1.5 * (double)s->span) + 1;
    if (m < 2) m = 2;
    return m < s->n_variants ? m : s->n_variants;
}
}  // namespace

extern "C" uint32_t hp_synth_block_size(const hp_synth_spec* s, uint64_t* n_cells) {
    uint32_t R = num_reads(s);
    if (n_cells) *n_cells = (uint64_t)R * max_len(s);
    return R;
}

extern "C" int hp_synth_block(const hp_synth_spec* s, uint32_t* read_start, uint32_t* read_end,
                              uint64_t* row_off, uint8_t* alleles_2bit, uint8_t* quals,
                              uint8_t* var_flags, uint8_t* truth_out) {
    const uint32_t N = s->n_variants, S = s->span;
    if (N == 0 || S == 0) return HP_ERR_ARG;
    const uint32_t R = num_reads(s);
    const uint32_t ignored_permille = s->reserved;
    SplitMix64 rng(s->seed);

    std::vector<uint8_t> truth(N), qual_of(N), ignored(N, 0);
    for (uint32_t i = 0; i < N; ++i) truth[i] = rng.bit() ? 1 : 0;
    for (uint32_t i = 0; i < N; ++i) {
        double u = rng.u01();
        uint8_t flags = 0, q;
        if (u < 0.85) { q = 160; flags |= HP_VAR_SNV; }   // SNV
        else if (u < 0.97) q = 20;                        // indel
        else if (u < 0.98) q = 40;                        // SV
        else q = 80;                                      // tandem repeat
        qual_of[i] = q;
        var_flags[i] = flags;
    }
    if (ignored_permille > 0) {
        for (uint32_t i = 0; i < N; ++i) {
            if (rng.u01() * 1000.0 < (double)ignored_permille) {
                ignored[i] = 1;
                var_flags[i] |= HP_VAR_IGNORED;
            }
        }
    }
    if (truth_out) std::memcpy(truth_out, truth.data(), N);

    uint64_t total_cells_bound = (uint64_t)R * max_len(s);
    std::memset(alleles_2bit, 0, (size_t)((total_cells_bound + 3) / 4));

    std::vector<uint8_t> row_a, row_q;
    uint64_t off = 0;
    for (uint32_t r = 0; r < R; ++r) {
        double lraw = std::round((double)S * (0.5 + rng.u01()));
        uint32_t L = lraw < 2.0 ? 2u : (uint32_t)lraw;
        if (L > N) L = N;
        uint32_t start = (uint32_t)std::floor(rng.u01() * (double)(N - L + 1));
        if (start > N - L) start = N - L;
        uint8_t hap = rng.bit() ? 1 : 0;
        row_a.assign(L, HP_ALLELE_NOOVERLAP);
        row_q.assign(L, 0);
        for (uint32_t c = 0; c < L; ++c) {
            uint32_t v = start + c;
            bool flip = rng.u01() < s->error_rate;
            bool amb = rng.u01() < s->ambig_rate;
            if (ignored[v]) continue;  // ignored variants are NoOverlap in every row (astar_phaser.rs:435-442)
            if (amb) {
                row_a[c] = HP_ALLELE_AMBIGUOUS;
                row_q[c] = 0;
            } else {
                row_a[c] = (uint8_t)((truth[v] ^ hap) ^ (flip ? 1 : 0));
                row_q[c] = qual_of[v];
            }
        }
        // ReadSegment::new clipping: [first set, last set + 1); no set allele => empty region at len
        uint32_t first = L, last = L;
        for (uint32_t c = 0; c < L; ++c) if (row_a[c] < HP_ALLELE_AMBIGUOUS) { first = c; break; }
        for (uint32_t c = L; c-- > 0;) if (row_a[c] < HP_ALLELE_AMBIGUOUS) { last = c + 1; break; }
        if (first == L) { first = L; last = L; }
        read_start[r] = start + first;
        read_end[r] = start + last;
        row_off[r] = off;
        for (uint32_t c = first; c < last; ++c) {
            uint64_t cell = off + (c - first);
            alleles_2bit[cell >> 2] |= (uint8_t)(row_a[c] << (2 * (cell & 3)));
            quals[cell] = row_q[c];
        }
        off += last - first;
    }
    row_off[R] = off;
    return HP_OK;
}
