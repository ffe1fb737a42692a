// hc_plan.hpp -- look-ahead plan of the scatter-form evaluation (host only; no HIP dependency, so that the planner is
// unit-tested on CPU: tests/cpp/plan_test.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <deque>
#include <limits>
#include <vector>

#include "hc_limits.hpp"

namespace hc {

// Look-ahead plan (scatter form, see hc_kernels.hpp).  Made right after a step at time tgrid[0] has been enqueued: the
// block covers the next 16 or 32 predicted steps tgrid[j] = tgrid[0] + j*dt.  The pass launched with the plan has computed, for
// each block step, what the samples known at planning time contribute; the tables below say what each block step adds.
struct Plan {
    bool valid = false;
    double dt = 0.0;
    int j_next = 1;                       // next block step, 1..L
    double tgrid[kLookahead + 1] = {0};
    int s_cut[kLookahead]    = {0};       // pass: block step j+1 takes IRF samples s >= s_cut[j]
    int s_defer[kLookahead]  = {0};       // IRF sample whose "is there an older history sample" test is too close to call ahead of time (-1: none)
    // block step m = 1..L (index m): IRF samples involving the step's own sample (weight x width) ...
    int n_own[kLookahead + 1] = {0};
    int own_s[kLookahead + 1][kNearMax];
    double own_a[kLookahead + 1][kNearMax];
    // ... and the number of scatter results of earlier block steps it adds (term slots k = 0 .. n_terms - 1)
    int n_terms[kLookahead + 1] = {0};
    // scatter of block step i, IRF sample s: the (step, term slot, weight) targets of its result
    int n_tgt[kLookahead + 1][kScatterSamples];
    int tgt_step[kLookahead + 1][kScatterSamples][kTargets];
    int tgt_k[kLookahead + 1][kScatterSamples][kTargets];
    double tgt_coef[kLookahead + 1][kScatterSamples][kTargets];
    // scatter launched after block step i covers IRF samples [scat_lo[i], scat_hi[i]] (hi < lo: nothing)
    int scat_lo[kLookahead + 1] = {0}, scat_hi[kLookahead + 1] = {0};
    int misses = 0, cooldown = 0;
    bool has_exc = false;  // the pass also left the excitation force of the predicted times (E rows)
};

// Plans the block that follows the step just pushed (times[0], newest first).  On the predicted time grid it classifies, for
// every block step m and IRF sample s, the interpolation bracket of the query time tgrid[m] - tau_s by who owns its two
// samples: samples known now (the pass), earlier block steps (scatter targets), step m itself (own entries).  The
// comparisons and the weight arithmetic are those of find_bracket / InterpolateVelocity6D (src/hydro_forces.cpp:343-381).
// Returns false (plan invalid) when a block cannot be planned; pl.cooldown then says for how many steps not to retry.
inline bool build_plan(Plan& pl, int lookahead, const std::deque<double>& times, const std::vector<double>& tau,
                       const std::vector<double>& width) {
    const int keep_misses = pl.misses, keep_cool = pl.cooldown;
    pl             = Plan{};
    pl.misses      = keep_misses;
    pl.cooldown    = keep_cool;
    const int H    = static_cast<int>(times.size());
    const int S    = static_cast<int>(tau.size());
    if (lookahead <= 0 || H < 2 || pl.cooldown > 0 || S < 2 || tau.front() < 0.0) return false;
    const double t0 = times[0], dt = times[0] - times[1];
    if (!(dt > 0.0)) return false;
    const int L = lookahead;  // steps per block: 16 or 32
    if (dt * L > 0.5 * (tau.back() - tau.front())) {
        pl.cooldown = 64;  // a block would span most of the IRF window: the scatter launches would re-read most of K every step
        return false;
    }
    pl.dt = dt;
    for (int j = 0; j <= L; ++j) pl.tgrid[j] = (j == 0) ? t0 : t0 + j * dt;
    auto G = [&](int idx) { return idx >= 1 ? pl.tgrid[idx] : times[static_cast<size_t>(-idx)]; };  // idx > -H
    for (int i = 0; i <= L; ++i) {
        pl.scat_lo[i] = S;
        pl.scat_hi[i] = -1;
        for (int s = 0; s < kScatterSamples; ++s) pl.n_tgt[i][s] = 0;
    }
    const double oldest = times.back();
    for (int m = 1; m <= L; ++m) {
        const int j = m - 1;
        // pass: samples s >= s_cut[j] of block step m need only history known now and the (zero) not-yet-known sample at
        // tgrid[1]: tgrid[m] - tau_s <= tgrid[1] (same expression as the kernel)
        int sc = 0;
        while (sc < S && !(pl.tgrid[m] - tau[sc] <= pl.tgrid[1])) ++sc;
        pl.s_cut[j] = sc;
        // While the history is shorter than the IRF window the reference's "no older sample -> the IRF step contributes
        // nothing" rule (src/hydro_forces.cpp:604-606) makes the sum discontinuous in t where t - tau_s crosses the oldest
        // sample time; a predicted time that differs from the caller's by an ulp could flip that decision.  The one sample
        // per step for which the test is too close to call is left out of the pass and evaluated by the step itself.
        pl.s_defer[j]       = -1;
        const double margin = 8.0 * std::max(1e-9 * dt, 64.0 * std::numeric_limits<double>::epsilon() * std::fabs(pl.tgrid[m]));
        const double target = pl.tgrid[m] - oldest;  // tau of the sample that lands on the oldest history time
        if (target <= tau.back() + margin) {
            const auto it = std::lower_bound(tau.begin(), tau.end(), target - margin);
            if (it != tau.end() && std::fabs(*it - target) <= margin) pl.s_defer[j] = static_cast<int>(it - tau.begin());
        }
        // brackets that touch a block sample (grid index >= 1)
        for (int s = 0; s < S; ++s) {
            const double q = pl.tgrid[m] - tau[s];
            int lo = 0;  // smallest lo with G(m - lo - 1) <= q
            while (m - lo - 1 > -H && G(m - lo - 1) > q) ++lo;
            const int nm = m - lo, om = nm - 1;
            if (nm <= 0) break;  // both samples known now: the pass has it, and so it has every later s
            if (om <= -H) break; // (cannot happen for nm >= 1)
            if (s >= kScatterSamples || s == pl.s_defer[j]) return false;
            const double newer = G(nm), older = G(om);
            double wo = 0.0, wn = 0.0;
            if (q == older) { wo = 1.0; wn = 0.0; }
            else if (q == newer) { wo = 0.0; wn = 1.0; }
            else if (q > older && q < newer) {
                const double td = newer - older;
                wo = (td != 0.0) ? ((newer - q) / td) : 0.0;
                wn = 1.0 - wo;
            } else {
                return false;
            }
            const int idx[2]     = {nm, om};
            const double wgt[2]  = {wn, wo};
            for (int e = 0; e < 2; ++e) {
                if (wgt[e] == 0.0 || idx[e] < 1) continue;
                if (idx[e] == m) {
                    if (pl.n_own[m] >= kNearMax - 1) return false;  // one entry stays free for the deferred sample
                    pl.own_s[m][pl.n_own[m]] = s;
                    pl.own_a[m][pl.n_own[m]] = wgt[e] * width[s];
                    pl.n_own[m]++;
                } else {
                    const int i = idx[e];
                    if (pl.n_terms[m] >= kTermMax || pl.n_tgt[i][s] >= kTargets) return false;
                    const int k = pl.n_terms[m]++;
                    const int t = pl.n_tgt[i][s]++;
                    pl.tgt_step[i][s][t] = m;
                    pl.tgt_k[i][s][t]    = k;
                    pl.tgt_coef[i][s][t] = wgt[e];
                    pl.scat_lo[idx[e]] = std::min(pl.scat_lo[idx[e]], s);
                    pl.scat_hi[idx[e]] = std::max(pl.scat_hi[idx[e]], s);
                }
            }
        }
    }
    pl.j_next = 1;
    pl.valid  = true;
    return true;
}


}  // namespace hc
