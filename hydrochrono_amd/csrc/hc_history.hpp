// hc_history.hpp -- bookkeeping of the velocity-history ring (host only; no HIP dependency, unit-tested on CPU:
// tests/cpp/history_test.cpp).  The samples themselves live in a ring in HBM; this is the index: which times are kept, newest
// first, exactly as TestHydro keeps them (push front + PruneHistory, src/hydro_forces.cpp:327-340,559-574), which ring slot
// holds the newest one, and what a step BACK in time does to both.
#pragma once
#include <deque>

namespace hc {

// Samples the prune rule has retired stay addressable for a while (their ring slots are simply not overwritten yet): a step back in
// time re-admits them, so that the history after a rewind is what the reference's rule keeps at the earlier time.
constexpr int kRewindSlack = 64;

struct HistoryAdvance {
    enum Status { kOk = 0, kDuplicateTime = 1 };
    Status status = kOk;
    int H         = 0;      // samples kept, the new one included
    int dropped   = 0;      // samples of an abandoned attempt removed by a step back in time
    bool rewound  = false;
    bool grow     = false;  // the ring is too small: re-allocate for `grow_need` samples keeping the `grow_have` newest stored ones
    int grow_need = 0, grow_have = 0;
};

// Registers the sample of time t.  `times` (kept samples, newest first), `retired` (samples the prune rule dropped, newest first,
// still in the ring slots behind the oldest kept one) and `head` (ring slot of the newest STORED sample, -1: none) are updated up
// to the point where the new sample's slot is chosen: the caller re-allocates the ring if asked to (which re-bases head), then
// advances head by one slot for the new sample.
//
// A step BACK in time (t below the newest sample: an integrator that rejected a step and retries from an earlier time -- the YAML
// runner's HHT does): the samples at times >= t belong to the abandoned attempt and are dropped; samples the prune rule had retired
// since then are re-admitted until one older than the window of the new time is there again (the prune rule read backwards), so the
// history is what the reference's rule keeps for a run that arrives at t with the surviving samples.  (The reference itself has no
// such rule: it inserts the earlier time in front of its newest-first list, src/hydro_forces.cpp:559-574, keeps the abandoned
// samples and walks a non-monotone list from then on; that is deliberately not reproduced.)
inline HistoryAdvance history_advance(std::deque<double>& times, std::deque<double>& retired, int& head, int Hcap, double t, double tau_last) {
    HistoryAdvance r;
    if (!times.empty() && t == times.front()) {  // "Tried to compute the radiation damping convolution twice within the same time step!" (:555-557)
        r.status = HistoryAdvance::kDuplicateTime;
        return r;
    }
    const double history_min_time = t - tau_last;
    if (!times.empty() && t < times.front()) {
        while (!times.empty() && times.front() >= t) {
            times.pop_front();
            ++r.dropped;
        }
        while (times.empty() && !retired.empty() && retired.front() >= t) {  // (a rewind past everything the rule had kept)
            retired.pop_front();
            ++r.dropped;
        }
        // the ring keeps the dropped samples' slots for the samples to come
        head = (times.empty() && retired.empty()) ? -1 : ((head - r.dropped) % Hcap + Hcap) % Hcap;
        while (!retired.empty() && (times.empty() || times.back() >= history_min_time)) {
            times.push_back(retired.front());
            retired.pop_front();
        }
        r.rewound = true;
    }
    times.push_front(t);
    while (times.size() > 1 && times[times.size() - 2] < history_min_time) {  // PruneHistory (:327-340)
        retired.push_front(times.back());  // retired, still in its ring slot
        times.pop_back();
    }
    r.H = static_cast<int>(times.size());
    // Retired samples live in the slots behind the oldest kept one: the newest kRewindSlack of them, and the ring makes room for them --
    // a caller whose step is well below the IRF spacing keeps more samples than the IRF has (H ~ S * dt_rirf / dt), and a ring that was
    // just large enough for those had no slot left for a retired one: a step back in time then found nothing to re-admit and the
    // oldest bracket of the window was missing (profiles/fuzz_parity.py, seed 1000148: 0.5 % of the radiation force).
    while (static_cast<int>(retired.size()) > kRewindSlack) retired.pop_back();
    if (r.H + static_cast<int>(retired.size()) > Hcap) {
        r.grow      = true;
        r.grow_need = r.H + kRewindSlack;
        r.grow_have = r.H - 1 + static_cast<int>(retired.size());
    }
    return r;
}

}  // namespace hc
