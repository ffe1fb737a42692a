// CPU test of the hand-off behind hc_step_multi / hc_added_mass_mv_multi (hydrochrono_amd/csrc/hc_fanout.hpp, host only): every
// item of every call runs exactly once, on the right thread, before the call returns -- through spinning and sleeping workers, calls
// of changing width, a pool that grows, two threads that want the pool at once, and teardown with workers asleep / spinning.
// Built with plain g++ (and with -fsanitize=thread by tests/test_fanout_cpu.py).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include <sys/wait.h>
#include <unistd.h>

#include "../../hydrochrono_amd/csrc/hc_fanout.hpp"

static int failures = 0;
#define CHECK(cond, ...)                          \
    do {                                          \
        if (!(cond)) {                            \
            ++failures;                           \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);             \
            std::printf("\n");                    \
        }                                         \
    } while (0)

struct Job {
    std::vector<long> count;        // plain (non-atomic) words, one per item: a lost hand-off or a double run shows as a wrong count
    std::vector<std::thread::id> who;
    long payload = 0;               // written by the caller before the call, read by every item (publication of the job)
    std::vector<long> seen_payload;
    explicit Job(int n) : count(n, 0), who(n), seen_payload(n, -1) {}
};

static void exercise(hc::FanOut& pool, int rounds, int max_n, int sleep_every, int sleep_us) {
    const auto me = std::this_thread::get_id();
    Job job(max_n);
    std::vector<long> expect(max_n, 0);
    for (int r = 0; r < rounds; ++r) {
        const int n = 1 + (r * 7 + r / 3) % max_n;
        job.payload = r;
        auto item   = [&](int g) {
            job.count[g] += 1;
            job.who[g]          = std::this_thread::get_id();
            job.seen_payload[g] = job.payload;
        };
        pool.run(n, item);
        for (int g = 0; g < n; ++g) {
            expect[g] += 1;
            CHECK(job.count[g] == expect[g], "round %d item %d ran %ld times in all, expected %ld", r, g, job.count[g], expect[g]);
            CHECK(job.seen_payload[g] == r, "round %d item %d saw the payload of round %ld", r, g, job.seen_payload[g]);
        }
        for (int g = n; g < max_n; ++g) CHECK(job.count[g] == expect[g], "round %d: item %d beyond the call's width ran", r, g);
        CHECK(job.who[0] == me, "item 0 must run on the calling thread");
        if (sleep_every > 0 && r % sleep_every == sleep_every - 1) std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
    }
}

int main() {
    {   // spinning workers only (the spin budget is never used up)
        hc::FanOut pool(63, 1e9);
        exercise(pool, 20000, 8, 0, 0);
        CHECK(pool.workers() == 7, "expected 7 workers, have %d", pool.workers());
    }
    {   // workers that fall asleep between calls (50 us of spinning, pauses of 300 us every few calls)
        hc::FanOut pool(63, 50.0);
        exercise(pool, 600, 6, 3, 300);
    }
    {   // fewer workers than items: the surplus runs on the caller
        hc::FanOut pool(2, 200.0);
        exercise(pool, 3000, 7, 0, 0);
        CHECK(pool.workers() <= 2, "the pool grew beyond its limit");
    }
    {   // no workers at all
        hc::FanOut pool(0, 200.0);
        exercise(pool, 100, 5, 0, 0);
        CHECK(pool.workers() == 0, "a pool of zero workers created one");
    }
    {   // two threads want the pool at once: whoever finds it busy runs its items itself
        hc::FanOut pool(63, 200.0);
        std::thread other([&] { exercise(pool, 4000, 5, 0, 0); });
        exercise(pool, 4000, 4, 0, 0);
        other.join();
    }
    {   // teardown right after a call (workers spinning) and after a pause (workers asleep)
        for (int k = 0; k < 20; ++k) {
            hc::FanOut pool(63, 30.0);
            exercise(pool, 5, 4, 0, 0);
            if (k % 2) std::this_thread::sleep_for(std::chrono::microseconds(400));
        }
    }
    {   // which kind of thread an item is on: workers say so, the calling thread never does -- not for item 0, not for the items
        // beyond the pool's limit, not for the items of a call that found the pool busy (hc_step.cpp: bind_device relies on it)
        hc::FanOut pool(2, 200.0);
        const auto me = std::this_thread::get_id();
        std::vector<int> on_worker(7, -1);
        std::vector<std::thread::id> who(7);
        auto item = [&](int g) {
            on_worker[g] = hc::FanOut::on_worker_thread() ? 1 : 0;
            who[g]       = std::this_thread::get_id();
        };
        for (int r = 0; r < 50; ++r) {
            pool.run(7, item);
            for (int g = 0; g < 7; ++g) {
                const bool mine = who[g] == me;
                CHECK(on_worker[g] == (mine ? 0 : 1), "item %d: on_worker_thread() = %d on %s thread", g, on_worker[g], mine ? "the calling" : "a worker");
                CHECK(mine == (g == 0 || g > 2), "item %d ran on the wrong kind of thread", g);
            }
        }
        CHECK(!hc::FanOut::on_worker_thread(), "the calling thread claims to be a worker");
        hc::FanOut none(0, 200.0);
        none.run(3, item);
        for (int g = 0; g < 3; ++g) CHECK(on_worker[g] == 0 && who[g] == me, "a pool without workers ran item %d elsewhere", g);
    }
#if !defined(__SANITIZE_THREAD__)  // (ThreadSanitizer refuses new threads after a multi-threaded fork)
    {   // fork(): the child inherits the pool's bookkeeping but not its threads -- its first call must not wait for workers that do
        // not exist, and its teardown must not join them
        hc::FanOut pool(63, 1e9);
        exercise(pool, 200, 5, 0, 0);
        CHECK(pool.workers() == 4, "expected 4 workers before the fork, have %d", pool.workers());
        std::fflush(stdout);
        const pid_t child = fork();
        if (child == 0) {
            alarm(20);  // a hang ends the child with SIGALRM
            int bad = 0;
            {
                std::vector<long> count(6, 0);
                auto item = [&](int g) { count[g] += 1; };
                for (int r = 0; r < 300; ++r) pool.run(6, item);
                for (int g = 0; g < 6; ++g) bad += count[g] != 300;
                bad += pool.workers() != 5;
            }
            _exit(bad ? 3 : 0);  // (no static destructors in the child; a pool with automatic storage is covered below)
        }
        int status = -1;
        CHECK(child > 0 && waitpid(child, &status, 0) == child, "waitpid failed");
        CHECK(WIFEXITED(status) && WEXITSTATUS(status) == 0, "the forked child did not finish its calls (status 0x%x)", status);
        exercise(pool, 200, 5, 0, 0);  // the parent's pool is untouched
        std::fflush(stdout);
        const pid_t child2 = fork();
        if (child2 == 0) {
            alarm(20);
            pool.~FanOut();  // teardown in a child that never used the pool: nothing to join
            _exit(0);
        }
        CHECK(child2 > 0 && waitpid(child2, &status, 0) == child2, "waitpid failed");
        CHECK(WIFEXITED(status) && WEXITSTATUS(status) == 0, "teardown of an inherited pool in a forked child failed (status 0x%x)", status);
        //  SEVERAL threads of the child enter run() before the pool has been re-initialised: one of them does it, the
        // others run their items themselves meanwhile; every call completes, every item runs exactly once
        std::fflush(stdout);
        const pid_t child3 = fork();
        if (child3 == 0) {
            alarm(20);
            std::atomic<int> go{0}, bad{0};
            auto caller = [&] {
                while (go.load() == 0) {
                }
                std::vector<long> count(6, 0);
                auto item = [&](int g) { count[g] += 1; };
                for (int r = 0; r < 200; ++r) pool.run(6, item);
                for (int g = 0; g < 6; ++g) bad += count[g] != 200;
            };
            std::thread a(caller), b(caller), c2(caller);
            go.store(1);
            a.join();
            b.join();
            c2.join();
            _exit(bad.load() ? 3 : 0);
        }
        CHECK(child3 > 0 && waitpid(child3, &status, 0) == child3, "waitpid failed");
        CHECK(WIFEXITED(status) && WEXITSTATUS(status) == 0, "concurrent first calls in a forked child failed (status 0x%x)", status);
    }
#endif
    std::printf("fanout_test: %d failures\n", failures);
    return failures ? 1 : 0;
}
