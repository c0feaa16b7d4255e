// See hostpool.h.  A job is a counter of chunks; threads (pool + caller) claim chunks with one atomic add each.
#include "hostpool.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Job {
    const std::function<void(int, int)> *body;
    int n, grain, chunks;
    std::atomic<int> next{0};          // next unclaimed chunk
    std::atomic<int> done{0};          // finished chunks
    std::mutex mu;
    std::condition_variable cv;        // the caller waits here for done == chunks
};

struct Pool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Job>> jobs;   // jobs that may still have unclaimed chunks
    int n_threads = 0;

    static bool run_chunk(Job &j) {
        const int c = j.next.fetch_add(1, std::memory_order_relaxed);
        if (c >= j.chunks) return false;
        const int i0 = c * j.grain, i1 = std::min(j.n, i0 + j.grain);
        (*j.body)(i0, i1);
        if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 == j.chunks) {
            std::lock_guard<std::mutex> lk(j.mu);
            j.cv.notify_all();
        }
        return true;
    }

    void worker() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    while (!jobs.empty() && jobs.front()->next.load(std::memory_order_relaxed) >= jobs.front()->chunks) jobs.pop_front();
                    if (!jobs.empty()) { j = jobs.front(); break; }
                    cv.wait(lk);
                }
            }
            while (run_chunk(*j)) {}
        }
    }

    explicit Pool(int n) : n_threads(n) {
        for (int i = 0; i < n; ++i) std::thread([this] { worker(); }).detach();
    }
};

Pool *pool() {
    // leaked on purpose: detached workers may still be parked on it when the process exits
    static Pool *p = [] {
        int n = -1;
        if (const char *e = getenv("DD_HOST_THREADS")) n = atoi(e);
        if (n < 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            n = (int)std::min(8u, std::max(1u, hw / 2));
        }
        return new Pool(std::min(n, 64));
    }();
    return p;
}

}  // namespace

namespace ddk {

int host_threads() { return pool()->n_threads; }

void parallel_for(int n, int grain, const std::function<void(int, int)> &body) {
    if (n <= 0) return;
    grain = std::max(1, grain);
    Pool *p = pool();
    const int chunks = (n + grain - 1) / grain;
    if (chunks == 1 || p->n_threads == 0) { body(0, n); return; }
    auto j = std::make_shared<Job>();
    j->body = &body; j->n = n; j->grain = grain; j->chunks = chunks;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->jobs.push_back(j);
    }
    p->cv.notify_all();
    while (Pool::run_chunk(*j)) {}
    std::unique_lock<std::mutex> lk(j->mu);
    j->cv.wait(lk, [&] { return j->done.load(std::memory_order_acquire) == j->chunks; });
}

}  // namespace ddk
