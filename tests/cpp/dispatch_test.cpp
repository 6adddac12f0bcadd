// dispatch_test.cpp — an unchanged single-process HiPhase on a multi-GPU node (reference src/main.rs:326-462: one solve_block job
// per phase block on `--threads` worker threads) through the one-call-site patch of INTEGRATION.md: T threads pull blocks from a
// queue and each calls hp_solve_blocks(1, &block, params, &out, device_id = -1). Behind the call the requests that are in flight
// together are merged and spread over the service threads of every visible device (HP_QUEUE_WORKERS = n runs the queue with n
// "devices" on a box with fewer GPUs). Checked against ONE hp_solve_blocks call over all blocks on device 0: every field of
// every block identical. Prints the whole-path rate of the worker pool (upload included) next to the one-call rate.
// usage: dispatch_test [threads = 64] [total hets = 4000]      exit 0 = ok, 1 = mismatch, 3 = no GPU
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/hiphase_gpu.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 64;
    const uint32_t total = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 4000;
    if (hp_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 3; }
    hp_synth_reads_spec spec;
    hp_synth_reads_defaults(&spec);
    spec.total_hets = total; spec.max_block_hets = 300; spec.seed = 4242; spec.noisy_fraction = 0.01;
    int st = 0;
    hp_synth_set* set = hp_synth_reads_create(&spec, &st);
    if (!set) { fprintf(stderr, "generator failed: %d\n", st); return 1; }
    size_t nb = 0;
    const hp_block_input* in = hp_synth_reads_inputs(set, &nb);
    uint64_t info[8];
    hp_synth_reads_info(set, info);
    hp_block_params prm{};
    prm.astar.min_queue_size = 1000; prm.astar.queue_increment = 3;
    prm.wfa_prune_distance = 500; prm.max_edit_distance = 500; prm.global_failure_ratio = 0.5; prm.global_failure_minimum = 50;
    prm.min_matched_alleles = 2; prm.global_realignment = 1;
    hp_outputs* o_ref = hp_outputs_create(in, nb);
    hp_outputs* o_pool = hp_outputs_create(in, nb);
    hp_block_output* ref = hp_outputs_array(o_ref);
    hp_block_output* pool = hp_outputs_array(o_pool);
    // reference: one call, one device (twice: the first one warms the thread's caches up)
    double t_one = 0.0;
    for (int rep = 0; rep < 2; ++rep) {
        const double t0 = now_s();
        if (hp_solve_blocks(nb, in, &prm, ref, 0) != HP_OK) { fprintf(stderr, "one call failed: %s\n", hp_last_error()); return 1; }
        t_one = now_s() - t0;
    }
    // the worker pool (twice: the first round starts the service threads)
    double t_pool = 0.0;
    std::atomic<int> failed{0};
    for (int rep = 0; rep < 2; ++rep) {
        std::atomic<size_t> next{0};
        const double t0 = now_s();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= nb) return;
                    if (hp_solve_blocks(1, &in[b], &prm, &pool[b], -1) != HP_OK) { fprintf(stderr, "block %zu failed: %s\n", b, hp_last_error()); failed.fetch_add(1); return; }
                }
            });
        for (auto& x : th) x.join();
        t_pool = now_s() - t0;
    }
    size_t bad = 0;
    for (size_t b = 0; b < nb; ++b) bad += hp_block_output_equal(&in[b], &ref[b], &pool[b]) ? 0 : 1;
    printf("{\"threads\": %d, \"blocks\": %zu, \"hets\": %llu, \"records\": %llu, \"one_call_s\": %.4f, \"one_call_hets_per_s\": %.0f, "
           "\"pool_s\": %.4f, \"pool_hets_per_s\": %.0f, \"mismatching_blocks\": %zu, \"failed_calls\": %d, \"queue_devices\": \"%s\"}\n",
           T, nb, (unsigned long long)info[1], (unsigned long long)info[2], t_one, (double)info[1] / t_one, t_pool, (double)info[1] / t_pool, bad,
           failed.load(), std::getenv("HP_QUEUE_WORKERS") ? std::getenv("HP_QUEUE_WORKERS") : "all visible");
    hp_outputs_destroy(o_ref); hp_outputs_destroy(o_pool); hp_synth_reads_destroy(set);
    return (bad || failed.load()) ? 1 : 0;
}
