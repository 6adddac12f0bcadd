// hp_combine.h — call coalescing behind the unchanged per-block entry points.
//
// HiPhase submits one job per phase block to a thread pool (reference src/main.rs:385-408): with `--threads T`, T threads
// sit in hp_astar_solve / hp_wfa_assign_batch / hp_solve_blocks at the same time, each with a block that fills a
// fraction of a percent of the GPU. A Combiner merges the calls that are in flight together into ONE device batch:
// every caller queues its request and sleeps; the combiner's service thread waits until every caller currently inside
// the entry point has queued (or a short window has passed), takes the whole queue, runs it as one batch, hands the
// results out and wakes the callers. The service thread is started by the first call and runs every batch, so the
// per-thread device-buffer cache, streams and scratch it uses (hp_common.h) stay warm from one batch to the next - a
// rotating "leader" among the callers would start cold every time. It owns nothing a caller could wait for at thread
// exit; the process simply ends with it asleep (the combiner is never destroyed). A lone caller does not pay the window:
// it is all the in-flight callers there are.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hp {

// hp_api.hip: HP_COALESCE=0 in the environment or hp_set_coalescing(0) turns the merging off (every call runs alone)
bool coalescing_enabled();

template <class Req> class Combiner {
public:
    using Run = void (*)(std::vector<Req*>&);
    explicit Combiner(Run run) : run_(run) {}
    void submit(Req* r) {
        if (inflight_.fetch_add(1, std::memory_order_acq_rel) == 0) {
            // nobody else is inside the entry point: run on the caller's own thread (its caches are the warm ones for a
            // single-threaded host) - whoever arrives meanwhile queues for the service thread and is merged there
            // (a lone runner never queues: the service thread must not wait for it, see serve())
            lone_.fetch_add(1, std::memory_order_acq_rel);
            std::vector<Req*> one(1, r);
            run_guarded(one);
            lone_.fetch_sub(1, std::memory_order_acq_rel);
            inflight_.fetch_sub(1, std::memory_order_acq_rel);
            return;
        }
        {
            std::unique_lock<std::mutex> lk(m_);
            if (!started_) { started_ = true; std::thread([this]() { serve(); }).detach(); }
            q_.push_back(r);
            cv_work_.notify_one();
            cv_done_.wait(lk, [r]() { return r->done; });
        }
        inflight_.fetch_sub(1, std::memory_order_acq_rel);
    }
    static bool enabled() { return coalescing_enabled(); }

private:
    void serve() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [this]() { return !q_.empty(); });
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(window_us());
            while ((int)q_.size() < inflight_.load(std::memory_order_acquire) - lone_.load(std::memory_order_acquire))   // someone is still on the way in
                if (cv_work_.wait_until(lk, deadline) == std::cv_status::timeout) break;
            std::vector<Req*> batch;
            batch.swap(q_);
            lk.unlock();
            run_guarded(batch);
            lk.lock();
            for (Req* x : batch) x->done = true;
            cv_done_.notify_all();
        }
    }
    // a throw out of a batch (std::bad_alloc of a host vector) must not leave its callers asleep for ever
    void run_guarded(std::vector<Req*>& batch) {
        try { run_(batch); }
        catch (...) { for (Req* x : batch) if (x->rc == 0) { x->rc = -2 /* HP_ERR_OOM */; x->err = "host allocation failed in a merged batch"; } }
    }
    static long window_us() {
        static const long w = [] { const char* e = std::getenv("HP_COALESCE_WINDOW_US"); return e ? std::max(0l, std::atol(e)) : 200l; }();
        return w;
    }
    Run run_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<Req*> q_;
    bool started_ = false;
    std::atomic<int> inflight_{0}, lone_{0};
};

}  // namespace hp
