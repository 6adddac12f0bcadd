// hp_combine.h — call coalescing behind the unchanged per-block entry points.
//
// HiPhase submits one job per phase block to a thread pool (reference src/main.rs:385-408): with `--threads T`, T threads
// sit in hp_astar_solve / hp_wfa_assign_batch / hp_solve_blocks at the same time, each with a block that fills a
// fraction of a percent of the GPU. A Combiner merges the calls that are in flight together into ONE device batch:
// every caller queues its request; the first one to find no leader becomes the leader, waits until every caller
// currently inside the entry point has queued (or a short window has passed), takes the whole queue, runs it as one
// batch, hands the results out and wakes the others. No service thread exists, so nothing has to be shut down at
// thread or process exit; a lone caller never waits (it is all the in-flight callers there are).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace hp {

// hp_api.hip: HP_COALESCE=0 in the environment or hp_set_coalescing(0) turns the merging off (every call runs alone)
bool coalescing_enabled();

template <class Req> class Combiner {
public:
    // run(batch) executes on the leader's thread and fills every request's results + rc
    template <class Run> void submit(Req* r, Run&& run) {
        inflight_.fetch_add(1, std::memory_order_acq_rel);
        std::unique_lock<std::mutex> lk(m_);
        q_.push_back(r);
        cv_.notify_all();   // a leader in its window counts arrivals
        while (!r->done) {
            if (!leader_active_) {
                leader_active_ = true;
                const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us());
                while ((int)q_.size() < inflight_.load(std::memory_order_acquire)) {
                    if (cv_.wait_until(lk, deadline) == std::cv_status::timeout) break;
                }
                std::vector<Req*> batch;
                batch.swap(q_);
                lk.unlock();
                run(batch);
                lk.lock();
                for (Req* x : batch) x->done = true;
                leader_active_ = false;
                cv_.notify_all();
            } else cv_.wait(lk);
        }
        lk.unlock();
        inflight_.fetch_sub(1, std::memory_order_acq_rel);
    }
    static bool enabled() { return coalescing_enabled(); }

private:
    static long window_us() {
        static const long w = [] { const char* e = std::getenv("HP_COALESCE_WINDOW_US"); return e ? std::max(0l, std::atol(e)) : 200l; }();
        return w;
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Req*> q_;
    bool leader_active_ = false;
    std::atomic<int> inflight_{0};
};

}  // namespace hp
