// coalesce_test.cpp — the shape of HiPhase's worker pool (reference src/main.rs:385-408: one solve_block job per block on
// `--threads` threads) against the unchanged per-block entry hp_astar_solve: T threads, each solving its own stream of
// small synthetic blocks, once with call coalescing off (every call is its own launch) and once with it on (calls in
// flight together become one resident batch, hiphase_amd/csrc/hp_combine.h). Results must be identical; prints both
// rates. usage: coalesce_test [threads] [blocks per thread] [min speedup]   exit 0 = ok, 1 = mismatch / too slow, 3 = no GPU
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/hiphase_gpu.h"

struct Block {
    hp_synth_spec spec;
    std::vector<uint32_t> rs, re;
    std::vector<uint64_t> ro;
    std::vector<uint8_t> al, ql, fl, h1, h2;
    hp_phase_stats st{};
    hp_block_view view() const {
        hp_block_view v{};
        v.n_variants = spec.n_variants; v.n_reads = (uint32_t)rs.size();
        v.read_start = rs.data(); v.read_end = re.data(); v.row_off = ro.data();
        v.alleles_2bit = al.data(); v.quals = ql.data(); v.var_flags = fl.data();
        return v;
    }
};

static Block make(uint32_t n, uint64_t seed) {
    Block b;
    b.spec = hp_synth_spec{n, 30, 20, 0, 0.01, 0.02, seed};
    uint64_t cells = 0;
    const uint32_t R = hp_synth_block_size(&b.spec, &cells);
    b.rs.resize(R); b.re.resize(R); b.ro.resize(R + 1); b.al.assign(cells / 4 + 2, 0); b.ql.assign(cells + 1, 0); b.fl.resize(n);
    hp_synth_block(&b.spec, b.rs.data(), b.re.data(), b.ro.data(), b.al.data(), b.ql.data(), b.fl.data(), nullptr);
    b.h1.assign(n, 9); b.h2.assign(n, 9);
    return b;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 64, per = argc > 2 ? std::atoi(argv[2]) : 12;
    const double min_speedup = argc > 3 ? std::atof(argv[3]) : 10.0;
    if (hp_device_count() < 1) { std::printf("no GPU: hp_astar_solve has no CPU fallback\n"); return 3; }
    // block sizes like a WGS run's (docs/user_guide.md:257: median 15 hets per block)
    const uint32_t sizes[8] = {15, 9, 40, 22, 120, 12, 60, 15};
    std::vector<std::vector<Block>> blocks(T), ref(T);
    uint64_t hets = 0;
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < per; ++k) { blocks[t].push_back(make(sizes[(t + k) % 8], 1000 + 131 * t + k)); hets += blocks[t].back().spec.n_variants; }
    const hp_astar_params prm{1000, 3, 0, 0};
    auto pass = [&](std::vector<std::vector<Block>>& bl) -> double {
        std::vector<std::thread> th;
        std::vector<int> rcs(T, 0);
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                for (auto& b : bl[t]) {
                    const hp_block_view v = b.view();
                    const int rc = hp_astar_solve(&v, &prm, b.h1.data(), b.h2.data(), &b.st);
                    if (rc != HP_OK) rcs[t] = rc;
                }
            });
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int t = 0; t < T; ++t) if (rcs[t] != 0) { std::printf("thread %d: status %d: %s\n", t, rcs[t], hp_last_error()); std::exit(1); }
        return dt;
    };
    hp_set_coalescing(0);
    ref = blocks;
    pass(ref);                       // warm-up of every thread's first call happens inside: time the second pass
    const double t_alone = pass(ref);
    hp_set_coalescing(1);
    pass(blocks);
    const double t_merged = pass(blocks);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < per; ++k) {
            const Block &a = ref[t][k], &m = blocks[t][k];
            if (a.h1 != m.h1 || a.h2 != m.h2 || std::memcmp(&a.st, &m.st, sizeof a.st) != 0) { std::printf("MISMATCH thread %d block %d\n", t, k); return 1; }
        }
    const double speedup = t_alone / t_merged;
    std::printf("%d threads x %d blocks (%llu hets): one launch per call %.0f hets/s, merged %.0f hets/s, %.1fx, bit-identical\n", T, per,
                (unsigned long long)hets, hets / t_alone, hets / t_merged, speedup);
    return speedup >= min_speedup ? 0 : 1;
}
