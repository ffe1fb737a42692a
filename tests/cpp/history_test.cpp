// CPU unit test of the history bookkeeping (hydrochrono_amd/csrc/hc_history.hpp: history_advance), no GPU and no HIP needed.
//
// A model of the ring (one double per slot: the sample's time) is driven by history_advance through random stepping patterns with
// steps back in time; after every push
//   * the kept samples are where the kernels look for them: sample k (0 = newest) in slot (head - k) mod Hcap, the retired ones
//     behind them;
//   * the kept list equals the reference's rule applied from scratch to the samples that survive (push front + PruneHistory,
//     src/hydro_forces.cpp:327-340,559-574: everything inside the IRF window plus exactly one older sample), as long as a rewind
//     does not reach further back than the retired samples kept addressable (kRewindSlack) -- the ring grows to keep room for them
//     (a ring exactly as large as the kept samples used to lose them: profiles/fuzz_parity.py, seed 1000148).
//   usage: history_test      (exit code 0 = all checks hold)
#include <algorithm>
#include <cstdio>
#include <deque>
#include <random>
#include <vector>

#include "../../hydrochrono_amd/csrc/hc_history.hpp"

namespace {

int run(unsigned seed, double tau_last, int cap0, int* rewinds_out, int* grows_out) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::deque<double> times, retired;
    int head = -1, Hcap = cap0;
    std::vector<double> ring(Hcap, -1e300);
    std::vector<double> alive;  // every sample pushed and not abandoned since, oldest first (the reference list without pruning)
    double t = 0.0, dt = 0.01;
    int rewinds = 0, grows = 0, exhausted_steps = 0;
    double lost_newest = -1e300;  // newest sample ever let go from the retired list (beyond kRewindSlack): what a later step back cannot get back
    bool since_rewind_exhausted = false;
    for (int n = 0; n < 6000; ++n) {
        const double u = U(rng);
        if (u < 0.02) dt = 0.002 + 0.02 * U(rng);                       // a new step size now and then
        double t_next = t + dt * (u < 0.3 ? 0.5 + U(rng) : 1.0);         // some jitter
        if (n > 5 && U(rng) < 0.03) {
            // step back: up to ~20 samples (within the slack), to a time between two samples or exactly onto one
            const int back = 1 + static_cast<int>(U(rng) * 20);
            const int idx  = std::max(0, static_cast<int>(alive.size()) - back);
            t_next         = (U(rng) < 0.4) ? alive[idx] : 0.5 * (alive[idx] + (idx > 0 ? alive[idx - 1] : alive[idx] - dt));
            ++rewinds;
        }
        if (!alive.empty() && t_next == alive.back()) t_next += 1e-9;  // (duplicate times are tested separately)
        t = t_next;
        while (!alive.empty() && alive.back() >= t) alive.pop_back();
        alive.push_back(t);
        const int retired_before = static_cast<int>(retired.size());
        std::vector<double> stored_before(times.begin(), times.end());  // everything still addressable before the call
        stored_before.insert(stored_before.end(), retired.begin(), retired.end());
        const hc::HistoryAdvance r = hc::history_advance(times, retired, head, Hcap, t, tau_last);
        if (r.status != hc::HistoryAdvance::kOk) return 1;
        // samples that are addressable no longer (and were not abandoned by a step back): let go for good by the cap on the retired list
        bool lost_now = false;
        for (double x : stored_before)
            if (x < t && std::find(times.begin(), times.end(), x) == times.end() && std::find(retired.begin(), retired.end(), x) == retired.end()) {
                lost_newest = std::max(lost_newest, x);
                lost_now    = true;
            }
        // ... which is the ONLY reason to let a sample go: never the size of the ring (profiles/fuzz_parity.py, seed 1000148)
        if (lost_now && static_cast<int>(retired.size()) < hc::kRewindSlack) return 8;
        if (r.grow) {
            // what ring_grow does: keep the grow_have newest stored samples, sample k -> slot (have - 1 - k), head = have - 1
            const int cap2 = std::max(2 * Hcap, r.grow_need + 16);
            std::vector<double> nr(cap2, -1e300);
            for (int k = 0; k < r.grow_have; ++k) nr[r.grow_have - 1 - k] = ring[((head - k) % Hcap + Hcap) % Hcap];
            ring.swap(nr);
            Hcap = cap2;
            head = r.grow_have - 1;
            ++grows;
        }
        head       = (head + 1) % Hcap;
        ring[head] = t;
        // 1. slots
        if (r.H != static_cast<int>(times.size()) || r.H + static_cast<int>(retired.size()) > Hcap) return 2;
        for (int k = 0; k < r.H; ++k)
            if (ring[((head - k) % Hcap + Hcap) % Hcap] != times[k]) return 3;
        for (size_t j = 0; j < retired.size(); ++j)
            if (ring[((head - r.H - static_cast<int>(j)) % Hcap + Hcap) % Hcap] != retired[j]) return 4;
        for (size_t k = 1; k < times.size(); ++k)
            if (!(times[k] < times[k - 1])) return 5;
        // 2. the reference's rule from scratch on the surviving samples: newest first, prune while the second-to-last is older than the window
        std::deque<double> ref(alive.rbegin(), alive.rend());
        while (ref.size() > 1 && ref[ref.size() - 2] < t - tau_last) ref.pop_back();
        if (ref.size() != times.size() || !std::equal(ref.begin(), ref.end(), times.begin())) {
            // allowed only when a rewind reached further back than the retired samples still addressable (kRewindSlack, ring room):
            // then every retired sample has been re-admitted and the kept list is the newest part of the reference's
            // (the reference rule would need a sample that the cap on the retired list -- kRewindSlack -- has let go, in this step back or in
            // one before it; a ring without room is no excuse any more: history_advance asks for a larger one instead)
            const bool exhausted = since_rewind_exhausted || (r.rewound && retired.empty() && ref.back() <= lost_newest);
            if (!exhausted || times.size() > ref.size() || !std::equal(times.begin(), times.end(), ref.begin())) {
                std::printf("   step %d t %.6f: kept %zu (oldest %.6f) vs reference rule %zu (oldest %.6f); retired %zu (before the call %d), rewound %d dropped %d, ring %d, newest sample let go %.6f\n", n, t,
                            times.size(), times.back(), ref.size(), ref.back(), retired.size(), retired_before, (int)r.rewound, r.dropped, Hcap, lost_newest);
                return 6;
            }
            since_rewind_exhausted = true;  // stays short until the window has moved past the missing samples
            ++exhausted_steps;
        } else {
            since_rewind_exhausted = false;
        }
    }
    if (exhausted_steps > 6000 / 10) return 7;  // the exact case must be the rule
    *rewinds_out = rewinds;
    *grows_out   = grows;
    return 0;
}

}  // namespace

int main() {
    int failures = 0;
    for (unsigned seed = 1; seed <= 8; ++seed) {
        int rewinds = 0, grows = 0;
        const double tau_last = (seed % 2) ? 1.27 : 0.31;
        const int rc = run(seed, tau_last, seed <= 4 ? 64 + hc::kRewindSlack : 8 + static_cast<int>(seed), &rewinds, &grows);  // (seeds 5-8: a ring that starts far too small)
        std::printf("seed %u tau_last %.2f: %s (%d rewinds, %d ring growths)\n", seed, tau_last, rc == 0 ? "ok" : "FAILED", rewinds, grows);
        if (rc != 0) std::printf("   check %d failed\n", rc);
        failures += rc != 0;
    }
    // the duplicate-time rule and a rewind before everything
    std::deque<double> times, retired;
    int head = -1;
    for (double t : {0.0, 0.01, 0.02}) {
        (void)hc::history_advance(times, retired, head, 128, t, 1.0);
        head = (head + 1) % 128;
    }
    if (hc::history_advance(times, retired, head, 128, 0.02, 1.0).status != hc::HistoryAdvance::kDuplicateTime || times.size() != 3) ++failures;
    const hc::HistoryAdvance r = hc::history_advance(times, retired, head, 128, -5.0, 1.0);
    if (!(r.rewound && r.dropped == 3 && r.H == 1 && head == -1 && times.size() == 1 && times[0] == -5.0)) ++failures;
    std::printf("%d failures\n", failures);
    return failures == 0 ? 0 : 1;
}
