// hc_fanout.hpp -- the hand-off behind hc_step_multi / hc_added_mass_mv_multi: one persistent worker thread per shard context
// beyond the first, so that ONE call of the host's time loop rings the doorbells of all G GPUs within one context's cost instead
// of one after the other (5 us per C4/8 context: the 8th GPU used to start 40 us after the first, against a 20 us step).
// Host only, no HIP dependency (unit-tested on CPU, under ThreadSanitizer too: tests/cpp/fanout_test.cpp).
//
// Protocol.  The caller publishes a job (function, argument, count n) and bumps `generation_` (release).  Worker w -- it serves
// item w + 1; item 0 runs on the calling thread -- waits for a generation it has not seen: it spins with `pause` for `spin_us`
// after its last job (a Chrono loop comes back within tens to hundreds of microseconds; a wake-up through the kernel would cost
// more than the step), then sleeps on a condition variable.  It runs its item if it has one and stores the generation into its
// `done` word (release); the caller, after its own item, spins until the `done` word of every worker shows the generation
// (acquire).  Items must not throw.  One caller at a time: a second thread that finds the pool busy runs its items itself, in order.
//
// Which thread runs an item matters to the caller (hc_step.cpp: bind_device caches the current HIP device per WORKER thread only):
// besides item 0, the calling thread also runs items when the pool is busy, when no thread could be created, and the items beyond
// `max_workers`.  on_worker_thread() says which kind of thread the item is on.
//
// fork(): the child has the pool's bookkeeping but none of its threads.  run() compares the process id with the one the workers were
// created under; in a child it forgets the inherited workers (their objects are leaked -- there is nothing to join) and their
// synchronisation objects (a mutex copied while a worker held it would stay locked for ever) and starts afresh.  Several threads of
// the child may enter run() at once: the one that swaps the recorded process id for a marker does the re-initialisation, the others
// run their items themselves meanwhile (the pool is not theirs yet) -- nothing of the pool is touched outside that one thread.
//
// Host cost: a worker that has just served a call spins for `spin_us` before it sleeps, i.e. a host that calls more often than that
// keeps n - 1 cores busy for the whole run (hydrochrono_amd: HC_MULTI_SPIN_US, INTEGRATION.md has the measured trade-off).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define HC_FANOUT_PAUSE() _mm_pause()
#else
#define HC_FANOUT_PAUSE() std::this_thread::yield()
#endif

namespace hc {

class FanOut {
  public:
    using Fn = void (*)(void* arg, int item);

    explicit FanOut(int max_workers = 63, double spin_us = 1000.0)
        : max_workers_(max_workers), spin_us_(spin_us), sync_(std::make_unique<Sync>()), pid_(::getpid()) {}
    FanOut(const FanOut&)            = delete;
    FanOut& operator=(const FanOut&) = delete;
    ~FanOut() {
        if (::getpid() != pid_.load(std::memory_order_acquire)) {  // a forked child: the threads belong to the parent
            forget_inherited_workers();
            return;
        }
        {
            std::lock_guard<std::mutex> lk(sync_->m);
            stop_.store(true, std::memory_order_seq_cst);
            generation_.fetch_add(1, std::memory_order_seq_cst);
        }
        sync_->cv.notify_all();
        for (auto& w : workers_)
            if (w->th.joinable()) w->th.join();
    }

    // true on a thread of a FanOut pool (any pool), false on every other thread -- the calling thread included, whichever items it runs
    static bool on_worker_thread() { return is_worker(); }

    // fn(arg, 0) on this thread, fn(arg, g) for g = 1 .. n - 1 on the workers, side by side; returns when all have returned.
    void run(int n, Fn fn, void* arg) {
        if (n <= 0) return;
        const pid_t me = ::getpid();
        pid_t owner    = pid_.load(std::memory_order_acquire);
        if (owner != me) {
            if (owner != kReinitialising && pid_.compare_exchange_strong(owner, kReinitialising, std::memory_order_acq_rel)) {
                after_fork(me);  // (ends by recording `me`)
            } else {
                for (int g = 0; g < n; ++g) fn(arg, g);  // another thread of this child is re-initialising the pool right now
                return;
            }
        }
        bool expected = false;
        if (n == 1 || max_workers_ <= 0 || !busy_.compare_exchange_strong(expected, true, std::memory_order_acquire)) {
            for (int g = 0; g < n; ++g) fn(arg, g);  // nothing to share out, or another thread is using the pool
            return;
        }
        const int helpers = std::min(n - 1, max_workers_);
        try {
            ensure_workers(helpers);
        } catch (...) {  // no thread to be had: the items run here, the pool stays usable
            while (!workers_.empty() && !workers_.back()->th.joinable()) workers_.pop_back();
            busy_.store(false, std::memory_order_release);
            for (int g = 0; g < n; ++g) fn(arg, g);
            return;
        }
        fn_  = fn;
        arg_ = arg;
        n_   = helpers + 1;
        const uint64_t gen = generation_.fetch_add(1, std::memory_order_seq_cst) + 1;
        if (sleepers_.load(std::memory_order_seq_cst) > 0) {
            { std::lock_guard<std::mutex> lk(sync_->m); }  // a worker between its predicate check and its wait holds the mutex
            sync_->cv.notify_all();
        }
        fn(arg, 0);
        for (int g = helpers + 1; g < n; ++g) fn(arg, g);  // (more items than workers allowed: the rest here)
        // EVERY worker acknowledges every generation, also those without an item: a worker still looking at this job's fields while
        // the next job is being published could otherwise run an item of that one twice
        for (auto& w : workers_)
            while (w->done.load(std::memory_order_acquire) != gen) HC_FANOUT_PAUSE();
        busy_.store(false, std::memory_order_release);
    }
    template <class F>
    void run(int n, F& f) {
        run(n, [](void* a, int g) { (*static_cast<F*>(a))(g); }, &f);
    }

    int workers() const { return static_cast<int>(workers_.size()); }

  private:
    struct alignas(64) Worker {
        std::thread th;
        std::atomic<uint64_t> done{0};
    };
    struct Sync {
        std::mutex m;
        std::condition_variable cv;
    };
    static bool& is_worker() {
        static thread_local bool flag = false;
        return flag;
    }

    // (child of a fork) the inherited Worker objects describe threads this process does not have: their std::thread members must be
    // neither joined nor destroyed as joinable, so the objects are leaked; the same for the mutex / condition variable
    void forget_inherited_workers() {
        for (auto& w : workers_) (void)w.release();
        workers_.clear();
        (void)sync_.release();
    }
    void after_fork(pid_t me) {
        forget_inherited_workers();
        sync_ = std::make_unique<Sync>();
        generation_.store(0, std::memory_order_seq_cst);
        sleepers_.store(0, std::memory_order_seq_cst);
        busy_.store(false, std::memory_order_seq_cst);
        stop_.store(false, std::memory_order_seq_cst);
        pid_.store(me, std::memory_order_release);
    }

    void ensure_workers(int count) {
        while (static_cast<int>(workers_.size()) < count) {
            workers_.push_back(std::make_unique<Worker>());
            Worker* w       = workers_.back().get();
            const int index = static_cast<int>(workers_.size()) - 1;
            // a new worker starts from the generation current NOW: the job that is about to be published is the first it sees
            const uint64_t seen0 = generation_.load(std::memory_order_seq_cst);
            w->done.store(seen0, std::memory_order_relaxed);
            w->th = std::thread([this, w, index, seen0] { loop(w, index, seen0); });
        }
    }

    void loop(Worker* w, int index, uint64_t seen) {
        is_worker() = true;
        Sync* const sy = sync_.get();  // (lives as long as the pool does in this process)
        auto idle_since = std::chrono::steady_clock::now();
        for (;;) {
            // wait for a generation not seen yet: spin first, then sleep
            uint64_t gen = generation_.load(std::memory_order_acquire);
            unsigned spins = 0;
            while (gen == seen) {
                HC_FANOUT_PAUSE();
                gen = generation_.load(std::memory_order_acquire);
                if (gen != seen) break;
                if ((++spins & 0x3FF) == 0 &&
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - idle_since).count() > spin_us_) {
                    std::unique_lock<std::mutex> lk(sy->m);
                    sleepers_.fetch_add(1, std::memory_order_seq_cst);
                    sy->cv.wait(lk, [&] { return generation_.load(std::memory_order_seq_cst) != seen; });
                    sleepers_.fetch_sub(1, std::memory_order_seq_cst);
                    gen = generation_.load(std::memory_order_acquire);
                }
            }
            if (stop_.load(std::memory_order_seq_cst)) return;
            seen = gen;
            // the job fields were written before the generation was bumped (release / acquire through generation_)
            if (index + 1 < n_) fn_(arg_, index + 1);
            w->done.store(gen, std::memory_order_release);
            idle_since = std::chrono::steady_clock::now();
        }
    }

    const int max_workers_;
    const double spin_us_;
    std::vector<std::unique_ptr<Worker>> workers_;
    std::atomic<uint64_t> generation_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> stop_{false}, busy_{false};
    std::unique_ptr<Sync> sync_;
    static constexpr pid_t kReinitialising = -1;  // (no process has this id)
    std::atomic<pid_t> pid_;
    Fn fn_     = nullptr;
    void* arg_ = nullptr;
    int n_     = 0;
};

}  // namespace hc
