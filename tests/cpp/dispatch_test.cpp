// dispatch_test.cpp — an unchanged single-process HiPhase on a (multi-)GPU node (reference src/main.rs:326-462: one solve_block job
// per phase block on `--threads` worker threads, 40 x threads job slots) through the per-block entries of the library:
//   blocking: T threads pull blocks from a queue and each calls hp_solve_blocks(1, &block, params, &out, -1) - the one-call-site
//             patch of INTEGRATION.md 3: T blocks in flight;
//   async:    T threads pull blocks, hp_block_submit each and keep up to 40 tickets pending per thread before they
//             hp_block_wait the oldest - the ~30-line main.rs patch of INTEGRATION.md 3c: the reference's own 40 x T job slots
//             in flight.
// Behind both the requests are merged into block sets that travel through a five-stage pipeline per visible device
// (HP_QUEUE_WORKERS = n runs the queue with n "devices" on a box with fewer GPUs). Checked against ONE hp_solve_blocks call over
// all blocks on device 0: every field of every block identical, in every pass of both modes. Prints the whole-path rates
// (upload included) next to the one-call rate.
// usage: dispatch_test [threads = 64] [total hets = 4000] [max block hets = 300] [passes = 3]
//        exit 0 = ok, 1 = mismatch / failure, 3 = no GPU
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>
#include <vector>

#include "../../include/hiphase_gpu.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int T = argc > 1 ? std::atoi(argv[1]) : 64;
    const uint32_t total = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 4000;
    const uint32_t max_block = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 300;
    const int passes = argc > 4 ? std::max(1, std::atoi(argv[4])) : 3;
    if (hp_device_count() < 1) { fprintf(stderr, "no GPU\n"); return 3; }
    hp_synth_reads_spec spec;
    hp_synth_reads_defaults(&spec);
    spec.total_hets = total; spec.max_block_hets = max_block; spec.seed = 4242;
    if (max_block <= 300) spec.noisy_fraction = 0.01;
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
    hp_block_output* ref = hp_outputs_array(o_ref);
    // one output set per pass and mode: a pass's results are checked after the clock has stopped
    std::vector<hp_outputs*> o_pool, o_async;
    for (int p = 0; p <= passes; ++p) { o_pool.push_back(hp_outputs_create(in, nb)); o_async.push_back(hp_outputs_create(in, nb)); }
    // reference: one call, one device (twice: the first one warms the thread's caches up)
    double t_one = 0.0;
    for (int rep = 0; rep < 2; ++rep) {
        const double t0 = now_s();
        if (hp_solve_blocks(nb, in, &prm, ref, 0) != HP_OK) { fprintf(stderr, "one call failed: %s\n", hp_last_error()); return 1; }
        t_one = now_s() - t0;
    }
    std::atomic<int> failed{0};
    // ---- blocking: T threads, one block per call. Pass 0 warms up (starts the pipelines), passes 1..P are timed back to back ----
    auto run_blocking = [&](int first_pass, int n_pass) {
        std::atomic<size_t> next{0};
        const size_t todo = nb * (size_t)n_pass;
        std::vector<std::thread> th;
        const double t0 = now_s();
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= todo) return;
                    const size_t b = k % nb;
                    hp_block_output* out = hp_outputs_array(o_pool[(size_t)first_pass + k / nb]);
                    if (hp_solve_blocks(1, &in[b], &prm, &out[b], -1) != HP_OK) { fprintf(stderr, "block %zu failed: %s\n", b, hp_last_error()); failed.fetch_add(1); return; }
                }
            });
        for (auto& x : th) x.join();
        return now_s() - t0;
    };
    run_blocking(0, 1);
    const double t_pool = run_blocking(1, passes);
    // ---- async: T threads, each keeps up to 40 submitted blocks pending (main.rs:328: job_slots = 40 x threads) ----
    auto run_async = [&](int first_pass, int n_pass) {
        std::atomic<size_t> next{0};
        const size_t todo = nb * (size_t)n_pass;
        std::vector<std::thread> th;
        const double t0 = now_s();
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                std::deque<uint64_t> pending;
                auto wait_oldest = [&]() {
                    if (hp_block_wait(pending.front()) != HP_OK) { fprintf(stderr, "a block failed: %s\n", hp_last_error()); failed.fetch_add(1); }
                    pending.pop_front();
                };
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= todo) break;
                    const size_t b = k % nb;
                    hp_block_output* out = hp_outputs_array(o_async[(size_t)first_pass + k / nb]);
                    if (pending.size() >= 40) wait_oldest();
                    uint64_t ticket = 0;
                    if (hp_block_submit(1, &in[b], &prm, &out[b], -1, &ticket) != HP_OK) { fprintf(stderr, "submit of block %zu failed: %s\n", b, hp_last_error()); failed.fetch_add(1); break; }
                    pending.push_back(ticket);
                }
                while (!pending.empty()) wait_oldest();
            });
        for (auto& x : th) x.join();
        return now_s() - t0;
    };
    run_async(0, 1);
    const double t_async = run_async(1, passes);
    size_t bad = 0;
    for (int p = 0; p <= passes; ++p)
        for (size_t b = 0; b < nb; ++b) {
            bad += hp_block_output_equal(&in[b], &ref[b], &hp_outputs_array(o_pool[(size_t)p])[b]) ? 0 : 1;
            bad += hp_block_output_equal(&in[b], &ref[b], &hp_outputs_array(o_async[(size_t)p])[b]) ? 0 : 1;
        }
    const double hets = (double)info[1] * passes;
    printf("{\"threads\": %d, \"blocks\": %zu, \"hets\": %llu, \"records\": %llu, \"largest_block_hets\": %llu, \"passes\": %d, \"one_call_s\": %.4f, \"one_call_hets_per_s\": %.0f, "
           "\"pool_s\": %.4f, \"pool_hets_per_s\": %.0f, \"async_s\": %.4f, \"async_hets_per_s\": %.0f, \"async_in_flight_per_thread\": 40, "
           "\"mismatching_blocks\": %zu, \"failed_calls\": %d, \"queue_devices\": \"%s\"}\n",
           T, nb, (unsigned long long)info[1], (unsigned long long)info[2], (unsigned long long)info[6], passes, t_one, (double)info[1] / t_one,
           t_pool, hets / t_pool, t_async, hets / t_async, bad, failed.load(), std::getenv("HP_QUEUE_WORKERS") ? std::getenv("HP_QUEUE_WORKERS") : "all visible");
    hp_outputs_destroy(o_ref);
    for (auto* o : o_pool) hp_outputs_destroy(o);
    for (auto* o : o_async) hp_outputs_destroy(o);
    hp_synth_reads_destroy(set);
    return (bad || failed.load()) ? 1 : 0;
}
